"""Host-side mirror of the reference's RNS interface on top of the C ABI.

Names follow the reference: `PolyContext` is the RNS part of `polyctx`
(src/poly.h:49-65), `poly_ntt` / `poly_invntt` / `poly_rns_mul` / `poly_rns_add`
are src/ntt.c:37,54 and src/poly.c:71-82 applied to whole limb-major slabs
`uint64[batch][dim][n]` in HBM, `he_mul_tensor` / `he_keyswitch` are the limb
loops of src/he-mult.c:116-138 and :58-66.  Slabs are torch int64 CUDA tensors
carrying the uint64 bit patterns (torch is the allocator/stream provider; the
arithmetic happens in libgpqhe_hip.so).
"""
import ctypes as C

import numpy as np

from . import _native


def _torch():
    import torch
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device(a, device=None):
    """numpy uint64 array -> int64 CUDA tensor with the same bits."""
    torch = _torch()
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device or "cuda")


def to_host(t):
    """int64 CUDA tensor -> numpy uint64 array."""
    return t.detach().cpu().numpy().view(np.uint64)


class PolyContext:
    """`polyctx_init(logn, q)` restricted to what the hot path reads: ring
    degree, the prime chain and the per-prime NTT tables (src/precomp.c:244-264,
    :354-380).  `nprimes` is `polyctx.dimub`; pass `logq` instead to get the
    reference's own bound (src/precomp.c:357)."""

    def __init__(self, logn, nprimes=None, logq=None, device=None):
        self.lib = _native.load()
        if nprimes is None:
            if logq is None:
                raise ValueError("PolyContext needs nprimes or logq")
            nprimes = self.lib.gpq_dimub(logn, logq)
        torch = _torch()
        if not torch.cuda.is_available():
            raise _native.GpqError("gpqhe_amd needs a HIP device; there is no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._dev = torch.device("cuda", self.device)     # workspaces and streams belong to the context's device, not the current one
        h = C.c_void_p()
        _native.check(self.lib.gpq_ctx_create(C.byref(h), logn, nprimes, self.device), "gpq_ctx_create")
        self.h = h
        self.logn, self.n, self.nprimes = logn, 1 << logn, nprimes
        self.p = [self.lib.gpq_ctx_const(h, d, 0) for d in range(nprimes)]

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, t):
        """device pointer of a slab, which must live on this context's device"""
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError("slab on %s, context on cuda:%d" % (t.device, self.device))
        return C.c_void_p(t.data_ptr())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpq_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def const(self, name, d):
        which = {"p": 0, "pinv_mont": 1, "pinv_barr": 2, "ninv": 3, "psi": 4}[name]
        return self.lib.gpq_ctx_const(self.h, d, which)

    def zetas(self, d, inverse=False):
        ptr = self.lib.gpq_ctx_zetas(self.h, d, 1 if inverse else 0)
        return np.ctypeslib.as_array(ptr, shape=(self.n,)).copy()

    def set_chunk(self, chunk):
        _native.check(self.lib.gpq_set_chunk(self.h, chunk), "gpq_set_chunk")

    def set_limb_classes(self, wide, split):
        """first `wide` limbs wide-split butterflies, up to `split` split-twiddle ones, the rest 7-mad (clamped; bit-identical)"""
        _native.check(self.lib.gpq_set_limb_classes(self.h, wide, split), "gpq_set_limb_classes")

    def set_limb_block(self, limbs):
        _native.check(self.lib.gpq_set_limb_block(self.h, limbs), "gpq_set_limb_block")

    # --- MPI <-> RNS bridge on big slabs uint64[batch][W][n] ---
    def rns_decompose(self, slab, big, W, dim):
        """src/rns.c:37-48 for all limbs."""
        batch = big.numel() // (W * self.n)
        _native.check(self.lib.gpq_rns_decompose(self.h, self._ptr(slab), self._ptr(big), W, dim, batch, self._stream()), "gpq_rns_decompose")
        return slab

    def rns_reconstruct(self, big, Wout, slab, dim, logq):
        """src/poly.c:109-120 with q = 2^logq (0: centre mod P only)."""
        batch = self._shape(slab, dim)
        _native.check(self.lib.gpq_rns_reconstruct(self.h, self._ptr(big), Wout, self._ptr(slab), dim, batch, logq, self._stream()), "gpq_rns_reconstruct")
        return big

    def set_bridge_mfma(self, on):
        """matrix-core (default) or integer-VALU rns_decompose; both are exact, the tests compare them"""
        _native.check(self.lib.gpq_set_bridge_mfma(self.h, 1 if on else 0), "gpq_set_bridge_mfma")

    def set_fused_tail(self, on):
        """with set_prescale(2): relinearisation tail in one pass per coefficient (1) or as front + CRT kernels (0, the default: the
        one-pass form measured 2 % slower); both exact, the tests compare them"""
        _native.check(self.lib.gpq_set_fused_tail(self.h, 1 if on else 0), "gpq_set_fused_tail")

    PRESCALE_DEFAULT = 3

    def set_prescale(self, mode):
        """he_mul / he_swk: what the inverse transforms pre-multiply for the kernels behind them -- 3 (default, also True): the weights of the
        key switch's whole basis, the relinearisation tail is one product; 2: per-basis CRT weights + w_j for the relinearisation front;
        1: per-basis CRT weights; 0 / False: nothing.  All exact; the tests compare them."""
        mode = self.PRESCALE_DEFAULT if mode is True else int(mode)
        _native.check(self.lib.gpq_set_prescale(self.h, mode), "gpq_set_prescale")

    def set_stream_bridge(self, on):
        """he_mul / he_swk: the fused streaming bridge kernels (default) or round 3's separate kernels; same words"""
        _native.check(self.lib.gpq_set_stream_bridge(self.h, 1 if on else 0), "gpq_set_stream_bridge")

    def set_lazy_decompose(self, on):
        """he_mul: its internal rns_decompose output may stay in (0, 3p) for the forward transforms (default) or be canonical; same results"""
        _native.check(self.lib.gpq_set_lazy_decompose(self.h, 1 if on else 0), "gpq_set_lazy_decompose")

    def set_overlap(self, on):
        """he_mul / he_swk / tensor / key switch over several launch groups: -1 = two lanes when affordable (default), 0 / False = one, 1 / True = two; same words"""
        v = -1 if (on is not True and on is not False and int(on) < 0) else (1 if on else 0)
        _native.check(self.lib.gpq_set_overlap(self.h, v), "gpq_set_overlap")

    def debug_fail_peer(self, on=True):
        """tests: the next creation of the peer lane (True / 1) or the next allocation of its workspace (2) fails like an allocation would"""
        _native.check(self.lib.gpq_debug_fail_peer(self.h, int(on)), "gpq_debug_fail_peer")

    def debug_table_bytes(self, which=0):
        """read-only device bytes owned by the context (0) / by its peer lane (1: zero, the peer borrows); 2: a peer exists; 3: the peer's table pointers are the context's"""
        return int(self.lib.gpq_debug_table_bytes(self.h, which))

    def last_lanes(self):
        """lanes (1 or 2) the last multi-group call on this context ran on"""
        return int(self.lib.gpq_last_lanes(self.h))

    def set_nt_policy(self, mode):
        """slab traffic of the transform kernels non-temporal: -1 by working set (default), 0 never, 1 always; never changes a word"""
        _native.check(self.lib.gpq_set_nt_policy(self.h, int(mode)), "gpq_set_nt_policy")

    def debug_force_redo(self, every):
        """tests: the streaming bridge kernels also flag every coefficient whose index is a multiple of `every` (0: off)"""
        _native.check(self.lib.gpq_debug_force_redo(self.h, int(every)), "gpq_debug_force_redo")

    def debug_zero_watch(self, on=True):
        """tests: keep copies of gpq_ntt's zero-flag words before and after the redo kernel (src/ntt.c:45-48)"""
        _native.check(self.lib.gpq_debug_zero_watch(self.h, 1 if on else 0), "gpq_debug_zero_watch")

    def debug_zero_flags(self, count):
        """(before, after) flag words of the last gpq_ntt launch group, word [polynomial * dim + limb]"""
        before, after = np.zeros(count, dtype=np.uint32), np.zeros(count, dtype=np.uint32)
        got = self.lib.gpq_debug_zero_flags(self.h, before.ctypes.data_as(C.c_void_p), after.ctypes.data_as(C.c_void_p), count)
        if got != count:
            raise RuntimeError("gpq_debug_zero_flags: %d words watched, %d asked for" % (got, count))
        return before, after

    def set_exact_crt(self, on):
        _native.check(self.lib.gpq_set_exact_crt(self.h, 1 if on else 0), "gpq_set_exact_crt")

    def poly_mul(self, r, a, b, W, dim, logq):
        """src/poly.c:84-107 on big slabs, q = 2^logq."""
        torch = _torch()
        batch = a.numel() // (W * self.n)
        ws = torch.empty(self.lib.gpq_poly_mul_workspace_bytes(self.h, dim, batch) // 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_poly_mul(self.h, self._ptr(r), self._ptr(a), self._ptr(b), W, dim, logq, batch, self._ptr(ws), self._stream()), "gpq_poly_mul")
        return r

    def poly_mul_general(self, r, a, b, W, dim, q):
        """src/poly.c:84-107 on big slabs for an arbitrary modulus q (Python int)."""
        torch = _torch()
        batch = a.numel() // (W * self.n)
        Lq = (q.bit_length() + 63) // 64
        qw = (_native.u64 * Lq)(*[(q >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(Lq)])
        ws = torch.empty(self.lib.gpq_poly_mul_general_workspace_bytes(self.h, dim, batch) // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_poly_mul_general(self.h, self._ptr(r), self._ptr(a), self._ptr(b), W, dim, qw, Lq, batch, self._ptr(ws), self._stream()),
                      "gpq_poly_mul_general")
        return r

    def big_addsub(self, out, a, b, W, mode):
        """out = a + b (mode 0), a - b (1), -a (2) on big slabs of W words, wrapping: src/he-add.c:32-140 before its mpi_smod."""
        polys = a.numel() // (W * self.n)
        _native.check(self.lib.gpq_big_addsub(self.h, self._ptr(out), self._ptr(a), self._ptr(b) if b is not None else None, W, polys, mode, self._stream()),
                      "gpq_big_addsub")
        return out

    def he_rs(self, c0, c1, W, logDelta, logql):
        """src/he-rescale.c:33-54 with Delta = 2^logDelta, q_l = 2^logql, in place."""
        batch = c0.numel() // (W * self.n)
        _native.check(self.lib.gpq_he_rs(self.h, self._ptr(c0), self._ptr(c1), W, logDelta, logql, batch, self._stream()), "gpq_he_rs")

    def he_dims(self, logqL, logql):
        """(dimP, dimA, dimB, dimevk) as src/precomp.c:401,407 and src/he-mult.c:99,51 compute them."""
        v = [C.c_uint() for _ in range(4)]
        _native.check(self.lib.gpq_he_dims(self.h, logqL, logql, *[C.byref(x) for x in v]), "gpq_he_dims")
        return tuple(x.value for x in v)

    def he_mul(self, out_c0, out_c1, ct1c0, ct1c1, ct2c0, ct2c1, rlk0, rlk1, W, logql, dimA, dimB, dimP):
        """src/he-mult.c:88-156 on big slabs, q_l = 2^logql."""
        torch = _torch()
        batch = ct1c0.numel() // (W * self.n)
        nbytes = self.lib.gpq_he_mul_workspace_bytes(self.h, W, dimA, dimB, dimP, batch)
        ws = torch.empty(nbytes // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_mul(self.h, self._ptr(out_c0), self._ptr(out_c1), self._ptr(ct1c0), self._ptr(ct1c1), self._ptr(ct2c0), self._ptr(ct2c1),
                                          self._ptr(rlk0), self._ptr(rlk1), W, logql, dimA, dimB, dimP, batch, self._ptr(ws), self._stream()), "gpq_he_mul")
        return ws

    def he_mul_rs(self, out_c0, out_c1, ct1c0, ct1c1, ct2c0, ct2c1, rlk0, rlk1, W, logql, dimA, dimB, dimP, logDelta):
        """he_mul then he_rs (src/he-mult.c:88-156, src/he-rescale.c:33-54) in one call: the rescale rides in the relinearisation tail."""
        torch = _torch()
        batch = ct1c0.numel() // (W * self.n)
        nbytes = self.lib.gpq_he_mul_workspace_bytes(self.h, W, dimA, dimB, dimP, batch)
        ws = torch.empty(nbytes // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_mul_rs(self.h, self._ptr(out_c0), self._ptr(out_c1), self._ptr(ct1c0), self._ptr(ct1c1), self._ptr(ct2c0), self._ptr(ct2c1),
                                             self._ptr(rlk0), self._ptr(rlk1), W, logql, dimA, dimB, dimP, logDelta, batch, self._ptr(ws), self._stream()), "gpq_he_mul_rs")
        return ws

    def relin_tail(self, out, chat, d, W, logql, dimB, dimP):
        """src/he-mult.c:67-77 alone (d = None: nothing added)."""
        torch = _torch()
        batch = self._shape(chat, dimB)
        ws = torch.empty(self.lib.gpq_relin_tail_workspace_bytes(self.h, W, dimB, dimP, batch) // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_relin_tail(self.h, self._ptr(out), self._ptr(chat), self._ptr(d) if d is not None else None, W, logql, dimB, dimP,
                                              batch, self._ptr(ws), self._stream()), "gpq_relin_tail")
        return out

    def relin_tail_overwriting(self, out, chat, d, W, logql, dimB, dimP):
        """src/he-mult.c:67-77 with `chat` given up as scratch: the tail as one product over all dimB limbs."""
        torch = _torch()
        batch = self._shape(chat, dimB)
        ws = torch.empty(self.lib.gpq_relin_tail_workspace_bytes(self.h, W, dimB, dimP, batch) // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_relin_tail_overwriting(self.h, self._ptr(out), self._ptr(chat), self._ptr(d) if d is not None else None, W, logql, dimB, dimP,
                                                          batch, self._ptr(ws), self._stream()), "gpq_relin_tail_overwriting")
        return out

    def he_swk(self, out_c0, out_c1, d0, d1, swk0, swk1, W, logql, dimB, dimP):
        """src/he-automorphism.c:40-85 on big slabs, q_l = 2^logql."""
        torch = _torch()
        batch = d0.numel() // (W * self.n)
        nbytes = self.lib.gpq_he_swk_workspace_bytes(self.h, W, dimB, dimP, batch)
        ws = torch.empty(nbytes // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_swk(self.h, self._ptr(out_c0), self._ptr(out_c1), self._ptr(d0), self._ptr(d1), self._ptr(swk0), self._ptr(swk1),
                                          W, logql, dimB, dimP, batch, self._ptr(ws), self._stream()), "gpq_he_swk")
        return ws

    @staticmethod
    def _words(q):
        L = (q.bit_length() + 63) // 64
        return (_native.u64 * L)(*[(q >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(L)]), L

    def he_mul_general(self, out_c0, out_c1, ct1c0, ct1c1, ct2c0, ct2c1, rlk0, rlk1, W, ql, dimA, dimB, dimP):
        """src/he-mult.c:88-156 for an arbitrary q_l (Python int)."""
        torch = _torch()
        batch = ct1c0.numel() // (W * self.n)
        qw, L = self._words(ql)
        ws = torch.empty(self.lib.gpq_he_general_workspace_bytes(self.h, W, dimA, dimB, dimP, batch) // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_mul_general(self.h, self._ptr(out_c0), self._ptr(out_c1), self._ptr(ct1c0), self._ptr(ct1c1), self._ptr(ct2c0), self._ptr(ct2c1),
                                                  self._ptr(rlk0), self._ptr(rlk1), W, qw, L, dimA, dimB, dimP, batch, self._ptr(ws), self._stream()),
                      "gpq_he_mul_general")

    def he_swk_general(self, out_c0, out_c1, d0, d1, swk0, swk1, W, ql, dimB, dimP):
        torch = _torch()
        batch = d0.numel() // (W * self.n)
        qw, L = self._words(ql)
        ws = torch.empty(self.lib.gpq_he_general_workspace_bytes(self.h, W, 0, dimB, dimP, batch) // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_swk_general(self.h, self._ptr(out_c0), self._ptr(out_c1), self._ptr(d0), self._ptr(d1), self._ptr(swk0), self._ptr(swk1),
                                                  W, qw, L, dimB, dimP, batch, self._ptr(ws), self._stream()), "gpq_he_swk_general")

    def he_rs_general(self, c0, c1, W, delta, ql):
        """src/he-rescale.c:33-54 for any Delta (< 2^64) and any q_l."""
        torch = _torch()
        batch = c0.numel() // (W * self.n)
        qw, L = self._words(ql)
        scratch = torch.empty(192, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_rs_general(self.h, self._ptr(c0), self._ptr(c1), W, delta, qw, L, batch, self._ptr(scratch), self._stream()), "gpq_he_rs_general")

    def he_mulpt(self, out_c0, out_c1, c0, c1, m, W, logql, dim):
        """src/he-mult.c:159-196 on big slabs."""
        torch = _torch()
        batch = c0.numel() // (W * self.n)
        ws = torch.empty(self.lib.gpq_he_mulpt_workspace_bytes(self.h, dim, batch) // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_mulpt(self.h, self._ptr(out_c0), self._ptr(out_c1), self._ptr(c0), self._ptr(c1), self._ptr(m), W, logql, dim, batch,
                                            self._ptr(ws), self._stream()), "gpq_he_mulpt")

    def he_genswk(self, evk0, evk1, p1, sk, e, sp, W, dimP, logqL, dimevk):
        """src/he-kem.c:74-118 from host-sampled p1 / e and the hidden polynomial sp; q_L = 2^logqL."""
        torch = _torch()
        nbytes = self.lib.gpq_he_genswk_workspace_bytes(self.h, W, dimP, logqL)
        if not nbytes:
            raise _native.GpqError("gpq_he_genswk_workspace_bytes: " + self.lib.gpq_last_error().decode())
        ws = torch.empty(nbytes // 8 + 8, dtype=torch.int64, device=self._dev)
        _native.check(self.lib.gpq_he_genswk(self.h, self._ptr(evk0), self._ptr(evk1), self._ptr(p1), self._ptr(sk), self._ptr(e), self._ptr(sp), W, dimP, logqL, dimevk,
                                             self._ptr(ws), self._stream()), "gpq_he_genswk")

    def poly_rot(self, r, a, W, rot):
        _native.check(self.lib.gpq_poly_rot(self.h, self._ptr(r), self._ptr(a), W, rot, a.numel() // (W * self.n), self._stream()), "gpq_poly_rot")
        return r

    def poly_conj(self, r, a, W):
        _native.check(self.lib.gpq_poly_conj(self.h, self._ptr(r), self._ptr(a), W, a.numel() // (W * self.n), self._stream()), "gpq_poly_conj")
        return r

    def phat_invmp(self, dim):
        return [self.lib.gpq_ctx_phat_invmp(self.h, dim, d) for d in range(dim)]

    def profile(self, on):
        _native.check(self.lib.gpq_profile_enable(self.h, 1 if on else 0), "gpq_profile_enable")

    def profile_collect(self):
        """{kernel name: (total ms, launches)} of the launches recorded since the last call."""
        k = self.lib.gpq_profile_kernels()
        ms = (C.c_double * k)()
        cnt = (C.c_ulonglong * k)()
        _native.check(self.lib.gpq_profile_collect(self.h, ms, cnt), "gpq_profile_collect")
        return {self.lib.gpq_profile_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(k) if cnt[i]}

    def _shape(self, slab, dim):
        per = dim * self.n
        if slab.numel() % per:
            raise ValueError("slab of %d words is not a multiple of dim*n = %d" % (slab.numel(), per))
        return slab.numel() // per

    # --- src/ntt.c:37,54 over slabs, in place ---
    def poly_ntt(self, slab, dim):
        _native.check(self.lib.gpq_ntt(self.h, self._ptr(slab), dim, self._shape(slab, dim), self._stream()), "gpq_ntt")
        return slab

    def poly_invntt(self, slab, dim):
        _native.check(self.lib.gpq_invntt(self.h, self._ptr(slab), dim, self._shape(slab, dim), self._stream()), "gpq_invntt")
        return slab

    def poly_ntt_reference(self, slab, dim, inverse=False):
        """src/ntt.c executed as written on the device (any input words): the kernel gpq_ntt redoes flagged limbs with."""
        _native.check(self.lib.gpq_ntt_reference(self.h, self._ptr(slab), dim, self._shape(slab, dim), 1 if inverse else 0, self._stream()), "gpq_ntt_reference")
        return slab

    # --- src/poly.c:71-82 over slabs ---
    def poly_rns_mul(self, r, a, b, dim):
        _native.check(self.lib.gpq_rns_mul(self.h, self._ptr(r), self._ptr(a), self._ptr(b), dim, self._shape(a, dim), self._stream()), "gpq_rns_mul")
        return r

    def poly_rns_add(self, r, a, b, dim):
        _native.check(self.lib.gpq_rns_add(self.h, self._ptr(r), self._ptr(a), self._ptr(b), dim, self._shape(a, dim), self._stream()), "gpq_rns_add")
        return r

    # --- limb loop of poly_mul, src/poly.c:96-103 ---
    def poly_mul_rns(self, r, a, b, dim):
        _native.check(self.lib.gpq_poly_mul_rns(self.h, self._ptr(r), self._ptr(a), self._ptr(b), dim, self._shape(a, dim), self._stream()),
                      "gpq_poly_mul_rns")
        return r

    def mulpt_rns(self, r0, r1, m, x0, x1, dim):
        _native.check(self.lib.gpq_mulpt_rns(self.h, self._ptr(r0), self._ptr(r1), self._ptr(m), self._ptr(x0), self._ptr(x1), dim, self._shape(m, dim), self._stream()),
                      "gpq_mulpt_rns")
        return r0, r1

    # --- he_mul RNS core ---
    def tensor_workspace(self, dim, batch):
        torch = _torch()
        nbytes = self.lib.gpq_tensor_workspace_bytes(self.h, dim, batch)
        return torch.empty(nbytes // 8, dtype=torch.int64, device=self._dev)

    def keyswitch_workspace(self, dim, batch):
        torch = _torch()
        nbytes = self.lib.gpq_keyswitch_workspace_bytes(self.h, dim, batch)
        return torch.empty(nbytes // 8, dtype=torch.int64, device=self._dev)

    def he_mul_tensor(self, d0, d1, d2, a0, a1, b0, b1, dim, workspace=None):
        """src/he-mult.c:116-138 on decomposed inputs: d0=a0*b0, d1=a0*b1+a1*b0, d2=a1*b1."""
        batch = self._shape(a0, dim)
        ws = workspace if workspace is not None else self.tensor_workspace(dim, batch)
        _native.check(self.lib.gpq_he_mul_tensor(self.h, self._ptr(d0), self._ptr(d1), self._ptr(d2), self._ptr(a0), self._ptr(a1), self._ptr(b0), self._ptr(b1),
                                                 dim, batch, self._ptr(ws), self._stream()), "gpq_he_mul_tensor")
        return d0, d1, d2

    def he_keyswitch(self, c0, c1, x, evk0, evk1, dim, workspace=None):
        """src/he-mult.c:58-66 / src/he-automorphism.c:59-67 on a decomposed input."""
        batch = self._shape(x, dim)
        ws = workspace if workspace is not None else self.keyswitch_workspace(dim, batch)
        _native.check(self.lib.gpq_keyswitch(self.h, self._ptr(c0), self._ptr(c1), self._ptr(x), self._ptr(evk0), self._ptr(evk1),
                                             dim, batch, self._ptr(ws), self._stream()), "gpq_keyswitch")
        return c0, c1


def ints_to_big(values, W):
    """Python ints (signed) -> numpy uint64 big slab [W][n], two's complement."""
    n = len(values)
    out = np.empty((W, n), dtype=np.uint64)
    mod = 1 << (64 * W)
    for i, v in enumerate(values):
        v = int(v) % mod
        for j in range(W):
            out[j, i] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out.reshape(-1)


def big_to_ints(big, W, n):
    """numpy uint64 big slab(s) [..][W][n] -> list of signed Python ints per polynomial."""
    arr = np.asarray(big, dtype=np.uint64).reshape(-1, W, n)
    res = []
    for poly in arr:
        vals = []
        for i in range(n):
            v = 0
            for j in range(W):
                v |= int(poly[j, i]) << (64 * j)
            if v >> (64 * W - 1):
                v -= 1 << (64 * W)
            vals.append(v)
        res.append(vals)
    return res


class StreamTimer:
    """HIP-event timer on the stream the kernels are launched on."""

    def __init__(self):
        self.lib = _native.load()
        self.h = C.c_void_p()
        _native.check(self.lib.gpq_timer_create(C.byref(self.h)), "gpq_timer_create")

    def start(self):
        _native.check(self.lib.gpq_timer_start(self.h, _stream()), "gpq_timer_start")

    def stop(self):
        _native.check(self.lib.gpq_timer_stop(self.h, _stream()), "gpq_timer_stop")

    def elapsed_ms(self):
        ms = C.c_float()
        _native.check(self.lib.gpq_timer_elapsed_ms(self.h, C.byref(ms)), "gpq_timer_elapsed_ms")
        return ms.value

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.gpq_timer_destroy(self.h)
            self.h = None
