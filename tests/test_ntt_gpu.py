"""Parity of the HIP NTT / INTT / pointwise kernels (through the C ABI) with the
CPU oracle and with the reference's golden digests.  Bit-exact: integer work."""
import numpy as np
import pytest

import gpqhe_amd
from gpqhe_amd import to_device, to_host
from oracle.oracle import fnv

pytestmark = pytest.mark.gpu

# (logn, dim, batch): small rings (LDS kernel), every two-pass geometry, ragged batch
SHAPES = [(1, 2, 1), (2, 1, 3), (7, 5, 2), (10, 3, 1), (12, 3, 2), (13, 2, 3), (14, 2, 1), (15, 10, 1), (16, 3, 2), (17, 2, 1)]


def _rand_slab(o, seed, dim, batch):
    return np.concatenate([o.gen(seed + 17 * k, dim) for k in range(batch)])


@pytest.mark.parametrize("logn,dim,batch", SHAPES)
def test_ntt_invntt_match_oracle(engine_ctx, oracle_ctx, logn, dim, batch):
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    assert g.p == o.p
    a = _rand_slab(o, 11, dim, batch)
    dev = to_device(a)
    g.poly_ntt(dev, dim)
    fwd = to_host(dev)
    assert np.array_equal(fwd, o.ntt_slab(a, dim)), "forward NTT differs from src/ntt.c:37-52"
    g.poly_invntt(dev, dim)
    assert np.array_equal(to_host(dev), a), "invntt(ntt(a)) != a"
    # inverse alone on arbitrary canonical input
    dev2 = to_device(a)
    g.poly_invntt(dev2, dim)
    assert np.array_equal(to_host(dev2), o.ntt_slab(a, dim, inverse=True)), "inverse NTT differs from src/ntt.c:54-73"


@pytest.mark.parametrize("logn", ["7", "12", "15", "16", "17"])
def test_golden_single_limb_digest(golden, engine_ctx, oracle_ctx, logn):
    """digest of ntt(gen(seed=1), limb 0) produced by the compiled reference."""
    kat = golden["ntt_kat_seed1_limb0"][logn]
    n = int(logn)
    o, g = oracle_ctx(n, 1), engine_ctx(n, 1)
    a = o.gen(1, 1)
    assert fnv(a) == kat["input"]
    dev = to_device(a)
    g.poly_ntt(dev, 1)
    out = to_host(dev)
    assert fnv(out) == kat["ntt"]
    assert [str(v) for v in out[:3]] == kat["out012"]


@pytest.mark.parametrize("logn", [7, 12, 16])
def test_context_tables_match_reference_constants(golden, engine_ctx, oracle_ctx, logn):
    rec = golden["prime_chain"][str(logn)]
    g = engine_ctx(logn, rec["count"] if logn < 16 else 8)
    assert [str(p) for p in g.p[: len(rec["first"])]] == rec["first"]
    k = golden["p0_constants"][str(logn)]
    assert str(g.const("pinv_mont", 0)) == k["pinv_mont"]
    assert str(g.const("pinv_barr", 0)) == k["pinv_barr"]
    assert str(g.const("ninv", 0)) == k["ninv"]
    z, zi = g.zetas(0), g.zetas(0, inverse=True)
    assert str(z[g.n // 2]) == k["zetas_n_2"] and str(z[1]) == k["zetas_1"] and str(zi[1]) == k["zetas_inv_1"]
    o = oracle_ctx(logn, g.nprimes)
    for d in (0, g.nprimes - 1):
        assert np.array_equal(g.zetas(d), o.zetas(d)) and np.array_equal(g.zetas(d, True), o.zetas(d, True))


@pytest.mark.parametrize("logn,dim,batch", [(7, 5, 2), (12, 2, 1), (16, 2, 2)])
def test_pointwise_match_oracle_and_alias(engine_ctx, oracle_ctx, logn, dim, batch):
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    n = o.n
    a, b = _rand_slab(o, 3, dim, batch), _rand_slab(o, 5, dim, batch)
    # edge values: 0, 1, p-1 in the first coefficients of every limb
    for k in range(batch):
        for d in range(dim):
            base = (k * dim + d) * n
            a[base:base + 4] = [0, 1, o.p[d] - 1, o.p[d] - 1]
            b[base:base + 4] = [o.p[d] - 1, o.p[d] - 1, o.p[d] - 1, 1]
    exp_mul = np.concatenate([o.rns_mul(a[i * n:(i + 1) * n], b[i * n:(i + 1) * n], i % dim) for i in range(dim * batch)])
    exp_add = np.concatenate([o.rns_add(a[i * n:(i + 1) * n], b[i * n:(i + 1) * n], i % dim) for i in range(dim * batch)])
    da, db = to_device(a), to_device(b)
    r = to_device(np.zeros_like(a))
    g.poly_rns_mul(r, da, db, dim)
    assert np.array_equal(to_host(r), exp_mul)
    g.poly_rns_add(r, da, db, dim)
    assert np.array_equal(to_host(r), exp_add)
    g.poly_rns_mul(da, da, db, dim)  # r aliases a, as src/he-mult.c:130 does
    assert np.array_equal(to_host(da), exp_mul)


@pytest.mark.parametrize("logn", [13, 16])
def test_lazy_arithmetic_extremes(engine_ctx, oracle_ctx, logn):
    """All-(p-1) and all-zero limbs drive the lazy ranges of modarith.hpp to their bounds."""
    dim = 2
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    for fill in ("max", "zero", "alt"):
        a = np.empty(dim * o.n, dtype=np.uint64)
        for d in range(dim):
            v = a[d * o.n:(d + 1) * o.n]
            if fill == "max":
                v[:] = o.p[d] - 1
            elif fill == "zero":
                v[:] = 0
            else:
                v[0::2] = o.p[d] - 1
                v[1::2] = 0
        dev = to_device(a)
        g.poly_ntt(dev, dim)
        assert np.array_equal(to_host(dev), o.ntt_slab(a, dim))
        dev = to_device(a)
        g.poly_invntt(dev, dim)
        assert np.array_equal(to_host(dev), o.ntt_slab(a, dim, inverse=True))


def test_bad_arguments_are_reported_not_fatal(engine_ctx):
    g = engine_ctx(7, 5)
    dev = to_device(np.zeros(6 * 128, dtype=np.uint64))
    with pytest.raises(gpqhe_amd.GpqError):
        g.poly_ntt(dev, 6)  # dim beyond the context's prime chain
    with pytest.raises(ValueError):
        g.poly_ntt(dev[:100], 5)  # ragged slab


def test_unsupported_prime_is_an_error_code_not_a_crash():
    """Primes outside the 2^59 + c family (c < 3.19e8) have no kernel: GPQ_ERR_UNSUPPORTED, message, no abort."""
    import ctypes as C
    lib = gpqhe_amd.load()
    n = 128
    p = (1 << 60) - 93                       # a 60-bit prime that is not 2^59 + small
    z = (C.c_uint64 * n)(*([1] * n))
    primes = (C.c_uint64 * 1)(p)
    zz = (C.POINTER(C.c_uint64) * 1)(C.cast(z, C.POINTER(C.c_uint64)))
    h = C.c_void_p()
    rc = lib.gpq_ctx_create_from_tables(C.byref(h), 7, 1, primes, zz, zz, 0)
    assert rc == -3 and b"2^59" in lib.gpq_last_error()
    rc = lib.gpq_ctx_create(C.byref(h), 18, 4, 0)          # ring degree beyond the supported range
    assert rc == -1


def test_largest_c_of_the_supported_chains(engine_ctx, oracle_ctx):
    """The lazy bounds of modarith.hpp are tightest for the largest c = p - 2^59: the 57-prime chain of
    logn = 17 reaches c = 2^28.13.  Every limb, random and all-(p-1) inputs, forward,
    inverse and the fused tensor stage against the oracle."""
    import torch
    logn, dim = 17, 57
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    assert max(o.p) - (1 << 59) > 1 << 28
    n = o.n
    a = o.gen(99, dim)
    a[(dim - 1) * n:] = o.p[dim - 1] - 1
    a[(dim - 2) * n:(dim - 1) * n:2] = o.p[dim - 2] - 1
    dev = to_device(a)
    g.poly_ntt(dev, dim)
    fwd = to_host(dev)
    assert np.array_equal(fwd, o.ntt_slab(a, dim))
    g.poly_invntt(dev, dim)
    assert np.array_equal(to_host(dev), a)
    b = o.gen(100, dim)
    outs = [torch.empty_like(dev) for _ in range(3)]
    g.he_mul_tensor(outs[0], outs[1], outs[2], dev, to_device(b), to_device(b), dev, dim)
    e0, e1, e2 = o.he_mul_tensor(a, b, b, a, dim)
    assert np.array_equal(to_host(outs[0]), e0) and np.array_equal(to_host(outs[1]), e1) and np.array_equal(to_host(outs[2]), e2)


@pytest.mark.parametrize("classes", [(0, 0), (0, 1 << 20), (3, 7)])
@pytest.mark.parametrize("logn,dim", [(16, 58), (17, 57), (13, 20)])
def test_kernel_families_agree_bit_for_bit(logn, dim, classes):
    """The default contexts pick, per limb, the cheapest butterflies its c allows -- wide-split (c < 2^27: a conditional
    subtraction every other forward stage), split (5-mad multiply), plain (7-mad) -- so one transform at n = 2^17 runs all
    three.  gpq_set_limb_classes moves limbs to the more general classes: (0, 0) = 7-mad butterflies on every limb, (0, all) =
    no wide-split ones (a subtraction in every stage), (3, 7) = all three classes inside one transform at every ring size.  Same
    slabs through both settings, whole he_mul core and the poly_mul limb loop included: bit-identical."""
    import os
    import torch
    import gpqhe_amd
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4242)
    outs = []
    for setting in (None, classes):
        g = gpqhe_amd.PolyContext(logn, dim)
        if setting is not None:
            g.set_limb_classes(*setting)
        if not outs:
            batch = 3
            slabs = []
            for _ in range(5):
                s = torch.empty((batch, dim, g.n), dtype=torch.int64, device="cuda")
                for d in range(dim):
                    s[:, d, :] = torch.randint(0, g.p[d], (batch, g.n), dtype=torch.int64, device="cuda", generator=gen)
                s[0, :, :64] = 0
                for d in range(dim):
                    s[0, d, 64:128] = g.p[d] - 1
                slabs.append(s.reshape(-1).contiguous())
            e0, e1 = slabs[3][: dim * g.n].clone(), slabs[4][: dim * g.n].clone()
        f = slabs[0].clone()
        g.poly_ntt(f, dim)
        i = slabs[1].clone()
        g.poly_invntt(i, dim)
        d0, d1, d2 = (torch.empty_like(f) for _ in range(3))
        g.he_mul_tensor(d0, d1, d2, slabs[0], slabs[1], slabs[2], slabs[3], dim, g.tensor_workspace(dim, batch))
        c0, c1 = torch.empty_like(f), torch.empty_like(f)
        g.he_keyswitch(c0, c1, slabs[4], e0, e1, dim, g.keyswitch_workspace(dim, batch))
        pa, pb, pr = slabs[0].clone(), slabs[2].clone(), torch.empty_like(f)
        g.poly_mul_rns(pr, pa, pb, dim)
        outs.append([f, i, d0, d1, d2, c0, c1, pr])
        g.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
