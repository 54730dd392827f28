"""ctypes binding of libgpqhe_hip.so (the C ABI in include/gpqhe_hip.h).

There is no fallback: if the HIP library is missing or fails to load, importing
anything that needs it raises.  Nothing here touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpqhe_hip.so")


def use_variant(path):
    """Load another build of the same library (`make -C gpqhe_amd/csrc variant NAME=..`) instead of the product: for the A/B timing
    tools and `pytest --variant PATH` only, and only before the first load.  No environment variable can swap the library (round 3's
    GPQHE_HIP_LIB is gone: the tests must run the product unless they are told otherwise on their command line)."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_variant(%r): the library is already loaded" % path)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    LIB_PATH = os.path.abspath(path)


u64 = C.c_uint64
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/gpqhe_hip.h one to one
SIGNATURES = {
    "gpq_ctx_create": (C.c_int, [C.POINTER(vp), C.c_uint, C.c_uint, C.c_int]),
    "gpq_ctx_create_from_tables": (C.c_int, [C.POINTER(vp), C.c_uint, C.c_uint, C.POINTER(u64),
                                             C.POINTER(C.POINTER(u64)), C.POINTER(C.POINTER(u64)), C.c_int]),
    "gpq_ctx_destroy": (None, [vp]),
    "gpq_ctx_logn": (C.c_uint, [vp]),
    "gpq_ctx_nprimes": (C.c_uint, [vp]),
    "gpq_ctx_device": (C.c_int, [vp]),
    "gpq_ctx_const": (u64, [vp, C.c_uint, C.c_int]),
    "gpq_ctx_zetas": (C.POINTER(u64), [vp, C.c_uint, C.c_int]),
    "gpq_dimub": (C.c_uint, [C.c_uint, C.c_uint]),
    "gpq_last_error": (C.c_char_p, []),
    "gpq_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "gpq_free": (C.c_int, [vp]),
    "gpq_upload": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "gpq_download": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "gpq_copy": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "gpq_stream_sync": (C.c_int, [vp]),
    "gpq_stream_wait": (C.c_int, [vp, vp]),
    "gpq_device_count": (C.c_int, []),
    "gpq_bind_thread_to_device": (C.c_int, [C.c_int]),
    "gpq_device_local_cpus": (C.c_int, [C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "gpq_probe_stream": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_set_device": (C.c_int, [C.c_int]),
    "gpq_stream_create": (C.c_int, [C.POINTER(vp)]),
    "gpq_stream_destroy": (C.c_int, [vp]),
    "gpq_malloc_host": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "gpq_free_host": (C.c_int, [vp]),
    "gpq_ntt": (C.c_int, [vp, vp, C.c_uint, C.c_uint, vp]),
    "gpq_invntt": (C.c_int, [vp, vp, C.c_uint, C.c_uint, vp]),
    "gpq_ntt_reference": (C.c_int, [vp, vp, C.c_uint, C.c_uint, C.c_int, vp]),
    "gpq_rns_mul": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, vp]),
    "gpq_rns_add": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, vp]),
    "gpq_poly_mul_rns": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, vp]),
    "gpq_mulpt_rns": (C.c_int, [vp] * 6 + [C.c_uint, C.c_uint, vp]),
    "gpq_set_chunk": (C.c_int, [vp, C.c_uint]),
    "gpq_set_limb_block": (C.c_int, [vp, C.c_uint]),
    "gpq_set_limb_classes": (C.c_int, [vp, C.c_uint, C.c_uint]),
    "gpq_tensor_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint]),
    "gpq_keyswitch_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint]),
    "gpq_he_mul_tensor": (C.c_int, [vp] * 8 + [C.c_uint, C.c_uint, vp, vp]),
    "gpq_keyswitch": (C.c_int, [vp] * 6 + [C.c_uint, C.c_uint, vp, vp]),
    "gpq_big_words": (C.c_uint, [C.c_uint]),
    "gpq_ctx_phat_invmp": (u64, [vp, C.c_uint, C.c_uint]),
    "gpq_ctx_pbits": (C.c_uint, [vp, C.c_uint]),
    "gpq_rns_decompose": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_rns_decompose_limbs": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_rns_reconstruct": (C.c_int, [vp, vp, C.c_uint, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_rns_reconstruct_one": (C.c_int, [vp, C.POINTER(u64), C.c_uint, C.POINTER(u64), C.c_uint]),
    "gpq_set_exact_crt": (C.c_int, [vp, C.c_int]),
    "gpq_set_stream_bridge": (C.c_int, [vp, C.c_int]),
    "gpq_debug_force_redo": (C.c_int, [vp, C.c_uint]),
    "gpq_debug_zero_watch": (C.c_int, [vp, C.c_int]),
    "gpq_debug_zero_flags": (C.c_long, [vp, vp, vp, C.c_size_t]),
    "gpq_set_lazy_decompose": (C.c_int, [vp, C.c_int]),
    "gpq_set_nt_policy": (C.c_int, [vp, C.c_int]),
    "gpq_set_overlap": (C.c_int, [vp, C.c_int]),
    "gpq_last_lanes": (C.c_uint, [vp]),
    "gpq_debug_fail_peer": (C.c_int, [vp, C.c_int]),
    "gpq_debug_table_bytes": (C.c_size_t, [vp, C.c_int]),
    "gpq_set_fused_tail": (C.c_int, [vp, C.c_int]),
    "gpq_big_transpose": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_int, vp]),
    "gpq_big_addsub": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.c_int, vp]),
    "gpq_set_prescale": (C.c_int, [vp, C.c_int]),
    "gpq_set_bridge_mfma": (C.c_int, [vp, C.c_int]),
    "gpq_poly_mul_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint]),
    "gpq_poly_mul": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_poly_mul_general_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint]),
    "gpq_rns_reconstruct_general": (C.c_int, [vp, vp, C.c_uint, vp, C.c_uint, C.c_uint, C.POINTER(u64), C.c_uint, vp, vp]),
    "gpq_poly_mul_general": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.POINTER(u64), C.c_uint, C.c_uint, vp, vp]),
    "gpq_he_rs": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_he_rescale": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_he_dims": (C.c_int, [vp, C.c_uint, C.c_uint] + [C.POINTER(C.c_uint)] * 4),
    "gpq_he_mul_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint]),
    "gpq_he_mul": (C.c_int, [vp] * 9 + [C.c_uint] * 6 + [vp, vp]),
    "gpq_he_mul_rs": (C.c_int, [vp] * 9 + [C.c_uint] * 7 + [vp, vp]),
    "gpq_relin_tail_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint]),
    "gpq_relin_tail": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_relin_tail_overwriting": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_he_mulpt_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint]),
    "gpq_he_mulpt": (C.c_int, [vp] * 6 + [C.c_uint] * 4 + [vp, vp]),
    "gpq_poly_rot": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_poly_conj": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, vp]),
    "gpq_he_general_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint]),
    "gpq_he_rs_general": (C.c_int, [vp, vp, vp, C.c_uint, C.c_ulonglong, C.POINTER(u64), C.c_uint, C.c_uint, vp, vp]),
    "gpq_relin_tail_general": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.POINTER(u64), C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_he_mul_general": (C.c_int, [vp] * 9 + [C.c_uint, C.POINTER(u64), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_he_mulpt_general": (C.c_int, [vp] * 6 + [C.c_uint, C.POINTER(u64), C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_he_swk_general": (C.c_int, [vp] * 7 + [C.c_uint, C.POINTER(u64), C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "gpq_big_add": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_big_sub": (C.c_int, [vp, vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_big_neg": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_evk_pack": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, vp]),
    "gpq_he_genswk_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint, C.c_uint]),
    "gpq_he_genswk": (C.c_int, [vp] * 7 + [C.c_uint] * 4 + [vp, vp]),
    "gpq_he_swk_workspace_bytes": (C.c_size_t, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint]),
    "gpq_he_swk": (C.c_int, [vp] * 7 + [C.c_uint] * 5 + [vp, vp]),
    "gpq_profile_enable": (C.c_int, [vp, C.c_int]),
    "gpq_profile_kernels": (C.c_int, []),
    "gpq_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "gpq_profile_collect": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
    "gpq_timer_create": (C.c_int, [C.POINTER(vp)]),
    "gpq_timer_start": (C.c_int, [vp, vp]),
    "gpq_timer_stop": (C.c_int, [vp, vp]),
    "gpq_timer_elapsed_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "gpq_timer_destroy": (None, [vp]),
    # reference-named drop-in symbols (include/gpqhe_hip_compat.h)
    "ntt": (None, [vp, vp]),
    "invntt": (None, [vp, vp]),
    "poly_ntt": (None, [vp, vp]),
    "poly_invntt": (None, [vp, vp]),
    "poly_rns_add": (None, [vp, vp, vp, vp]),
    "poly_rns_mul": (None, [vp, vp, vp, vp]),
    "montgomery_inv": (u64, [u64]),
    "barrett_inv": (u64, [u64]),
    "gpq_dropin_set_logn": (None, [C.c_uint]),
    "gpq_dropin_reset": (None, []),
    "gpq_mpi_shim_release": (None, []),
    "gpq_mpi_shim_forget_keys": (None, []),
    "gpq_mpi_shim_set_key_slots": (None, [C.c_uint]),
    "gpq_mpi_shim_set_key_check": (None, [C.c_int]),
    "gpq_mpi_shim_engine": (C.c_void_p, []),
    "gpq_compat_view": (C.c_void_p, [C.c_char_p]),
    "gpq_mpi_shim_resident_keys": (C.c_uint, []),
    "gpq_mpi_shim_set_poly_slots": (None, [C.c_uint]),
    "gpq_mpi_shim_resident_polys": (C.c_uint, []),
    "gpq_mpi_shim_poly_stats": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gpq_mpi_shim_forget_polys": (None, []),
    "gpq_mpi_shim_set_conversion_threads": (C.c_uint, [C.c_uint]),
    "gpq_mpi_shim_poly_bypass": (None, [C.c_int]),
    "gpq_mpi_shim_set_direct_mpi": (C.c_int, [C.c_int]),
    "gpq_mpi_shim_last_timing": (None, [C.POINTER(C.c_double)]),
    "gpq_fill_rns_chain": (C.c_int, [vp, C.c_uint, vp, C.c_int]),
    "gpq_release_rns_chain": (None, [vp]),
}
# by-value unsigned __int128 arguments cannot be expressed in ctypes; these two are
# exercised from C (tests/c/dropin_host.c)
EXPORTED_ONLY = ["montgomery_reduce", "barrett_reduce",
                 # MPI-typed surface: driven from C with real libgcrypt MPIs (tests/c/mpi_host.c)
                 "rns_decompose", "rns_reconstruct", "poly_rns2mpi", "poly_mul", "he_mul", "he_rs", "he_rescale", "he_moddown", "he_mulpt", "he_add", "he_sub", "he_addpt", "he_subpt", "he_neg", "he_copy_ct", "he_dec", "he_conj", "he_rot", "he_genrlk", "he_genck", "he_genrk",
                 ]
# libgpqhe_hip_ctx.so (ctx_compat.hip): context construction / storage names for hosts that are not GPQHE; driven from C
CTX_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgpqhe_hip_ctx.so")
CTX_EXPORTS = ["polyctx_init", "polyctx_exit", "hectx_init", "hectx_exit", "poly_mpi_alloc", "poly_mpi_free", "poly_rns_alloc", "poly_rns_free",
               "polyctx", "hectx", "GPQHE_TWO"]

_lib = None


def load():
    """Load the HIP library (torch first, when present, so that both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gpqhe_amd: %s is missing -- build it with `make -C gpqhe_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  (binds libamdhip64 once for the whole process)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name in EXPORTED_ONLY:
        getattr(lib, name)
    _lib = lib
    return lib


class GpqError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise GpqError("%s failed (%d): %s" % (what, rc, load().gpq_last_error().decode()))
