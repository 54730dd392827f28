"""Misuse of the slab API: every entry point answers with a negative code and a message (include/gpqhe_hip.h,
"Errors"), never a fault or an abort, and the context keeps working afterwards.  The reference's own convention
(errno + abort, src/reduce.c:95-100) is kept by the drop-in symbols only and is exercised in tests/test_dropin_c_gpu.py."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

GPQ_ERR_INVALID, GPQ_ERR_HIP, GPQ_ERR_UNSUPPORTED = -1, -2, -3


@pytest.fixture(scope="module")
def env():
    import torch
    import gpqhe_amd
    g = gpqhe_amd.PolyContext(13, 12)
    lib = g.lib
    n, dim, W = g.n, 12, 6
    buf = lambda words: torch.zeros(words, dtype=torch.int64, device="cuda")
    yield {"g": g, "lib": lib, "n": n, "dim": dim, "W": W, "buf": buf, "torch": torch}
    g.close()


def err(lib):
    return lib.gpq_last_error().decode()


def test_slab_entry_points_reject_bad_shapes_and_null_pointers(env):
    lib, g, n, dim = env["lib"], env["g"], env["n"], env["dim"]
    slab = env["buf"](2 * dim * n)
    p = slab.data_ptr()
    cases = [
        (lib.gpq_ntt(g.h, p, 0, 2, None), "dim=0"),
        (lib.gpq_ntt(g.h, p, dim + 1, 2, None), "dim=%d" % (dim + 1)),
        (lib.gpq_ntt(g.h, p, dim, 0, None), "empty batch"),
        (lib.gpq_ntt(g.h, None, dim, 2, None), "null slab"),
        (lib.gpq_ntt(None, p, dim, 2, None), "null context"),
        (lib.gpq_invntt(g.h, None, dim, 2, None), "null slab"),
        (lib.gpq_rns_mul(g.h, p, None, p, dim, 2, None), "null slab"),
        (lib.gpq_rns_add(g.h, None, p, p, dim, 2, None), "null slab"),
        (lib.gpq_poly_mul_rns(g.h, p, p, None, dim, 2, None), "null"),
        (lib.gpq_he_mul_tensor(g.h, p, p, p, p, p, p, p, dim, 2, None, None), "null pointer"),
        (lib.gpq_keyswitch(g.h, p, p, p, p, None, dim, 2, p, None), "null pointer"),
        (lib.gpq_set_chunk(g.h, 0), "gpq_set_chunk"),
        (lib.gpq_set_chunk(None, 4), "gpq_set_chunk"),
    ]
    for i, (rc, msg) in enumerate(cases):
        assert rc == GPQ_ERR_INVALID, (i, rc)
    # the message belongs to the last failing call
    assert "gpq_set_chunk" in err(lib)
    assert lib.gpq_ntt(g.h, p, dim + 1, 2, None) == GPQ_ERR_INVALID and "dim=%d outside 1..%d" % (dim + 1, dim) in err(lib)


def test_context_creation_rejects_out_of_range_parameters(env):
    lib = env["lib"]
    h = C.c_void_p()
    for logn, nprimes in ((0, 4), (18, 4), (13, 0), (13, 5000)):
        assert lib.gpq_ctx_create(C.byref(h), logn, nprimes, 0) == GPQ_ERR_INVALID
        assert "out of range" in err(lib)
    assert lib.gpq_ctx_create(None, 13, 4, 0) == GPQ_ERR_INVALID


def test_bridge_entry_points_reject_bad_arguments(env):
    lib, g, n, dim, W = env["lib"], env["g"], env["n"], env["dim"], env["W"]
    slab, big = env["buf"](dim * n), env["buf"](W * n)
    ps, pb = slab.data_ptr(), big.data_ptr()
    assert lib.gpq_rns_decompose(g.h, None, pb, W, dim, 1, None) == GPQ_ERR_INVALID
    assert lib.gpq_rns_decompose(g.h, ps, pb, 0, dim, 1, None) in (GPQ_ERR_INVALID, GPQ_ERR_UNSUPPORTED)
    assert lib.gpq_rns_decompose(g.h, ps, pb, 33, dim, 1, None) in (GPQ_ERR_INVALID, GPQ_ERR_UNSUPPORTED)
    assert lib.gpq_rns_decompose(g.h, ps, pb, W, dim + 1, 1, None) == GPQ_ERR_INVALID
    # output too short for q = 2^logq, and for P when logq = 0
    assert lib.gpq_rns_reconstruct(g.h, pb, 1, ps, dim, 1, 100, None) == GPQ_ERR_INVALID and "cannot hold" in err(lib)
    assert lib.gpq_rns_reconstruct(g.h, pb, 2, ps, dim, 1, 0, None) == GPQ_ERR_INVALID and "mod P" in err(lib)
    # poly_mul: q must be a power of two with logq > 0 here
    ws = env["buf"](lib.gpq_poly_mul_workspace_bytes(g.h, dim, 1) // 8 + 8)
    assert lib.gpq_poly_mul(g.h, pb, pb, pb, W, dim, 0, 1, ws.data_ptr(), None) == GPQ_ERR_INVALID
    assert lib.gpq_poly_mul(g.h, pb, pb, pb, W, dim, 100, 1, None, None) == GPQ_ERR_INVALID
    # relinearisation wants dimB > dimP
    out = env["buf"](W * n)
    ws2 = env["buf"](1 << 20)
    assert lib.gpq_relin_tail(g.h, out.data_ptr(), ps, None, W, 100, 4, 4, 1, ws2.data_ptr(), None) == GPQ_ERR_INVALID
    assert "dimB" in err(lib)
    # rotations are not in place
    assert lib.gpq_poly_rot(g.h, pb, pb, W, 1, 1, None) == GPQ_ERR_INVALID and "not in place" in err(lib)
    assert lib.gpq_poly_conj(g.h, pb, pb, W, 1, None) == GPQ_ERR_INVALID
    # general moduli: zero modulus, bad word count
    # (checked before anything is launched: the scratch is sized as documented all the same)
    zero = (C.c_uint64 * 2)(0, 0)
    wide = (C.c_uint64 * 8)(*([2 ** 64 - 1] * 8))
    scratch = env["buf"](lib.gpq_poly_mul_general_workspace_bytes(g.h, dim, 1) // 8 + 64)
    ps_ = scratch.data_ptr()
    assert lib.gpq_rns_reconstruct_general(g.h, pb, W, ps, dim, 1, zero, 2, ps_, None) == GPQ_ERR_INVALID
    assert "zero modulus" in err(lib)
    assert lib.gpq_rns_reconstruct_general(g.h, pb, W, ps, dim, 1, zero, 0, ps_, None) == GPQ_ERR_INVALID
    assert lib.gpq_rns_reconstruct_general(g.h, pb, W, ps, dim, 1, wide, 8, ps_, None) == GPQ_ERR_INVALID and "cannot hold" in err(lib)
    assert lib.gpq_poly_mul_general(g.h, pb, pb, pb, W, dim, zero, 2, 1, ps_, None) == GPQ_ERR_INVALID
    assert lib.gpq_he_rs_general(g.h, pb, pb, W, 1 << 20, zero, 1, 1, ps_, None) == GPQ_ERR_INVALID
    assert lib.gpq_he_rs_general(g.h, pb, pb, W, 0, wide, 2, 1, ps_, None) == GPQ_ERR_INVALID
    assert lib.gpq_he_mulpt_general(g.h, pb, pb, pb, pb, pb, W, zero, 2, dim, 1, ps_, None) == GPQ_ERR_INVALID
    # host-side single-coefficient reconstruction checks its residues
    res = (C.c_uint64 * dim)(*([2 ** 63] * dim))
    words = (C.c_uint64 * 16)()
    assert lib.gpq_rns_reconstruct_one(g.h, words, 16, res, dim) == GPQ_ERR_INVALID and "not reduced" in err(lib)
    assert lib.gpq_rns_reconstruct_one(g.h, words, 1, res, dim) == GPQ_ERR_INVALID


def test_context_still_exact_after_the_failures(env, oracle_ctx):
    """Nothing above may have left the context (or the device) in a bad state."""
    import numpy as np
    from gpqhe_amd import to_device, to_host
    g, dim = env["g"], 3
    o = oracle_ctx(13, dim)
    a = o.gen(5, dim)
    dev = to_device(a.copy())
    g.poly_ntt(dev, dim)
    assert np.array_equal(to_host(dev), o.ntt_slab(a, dim))
    assert env["lib"].gpq_stream_sync(None) == 0
