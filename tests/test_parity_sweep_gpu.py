"""Parity sweep of the MPI-level chain at n = 64 (the smallest ring the matrix-core bridge takes) over modulus sizes up to
the headline's: every limb count, word count and k-step count of the bridge kernels appears at least once, and the expected
values come from the Python-integer restatement of src/he-mult.c / src/he-automorphism.c (oracle/bigint_ref.py)."""
import random

import numpy as np
import pytest

from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host
from oracle import bigint_ref as ref

pytestmark = pytest.mark.gpu
LOGN = 6


def _torch():
    import torch
    return torch


def _centred(rng, logq, n):
    half = 1 << (logq - 1)
    edge = [half - 1, -half, 0, 1, -1]
    return [edge[i] if i < len(edge) else rng.randrange(-half, half) for i in range(n)]


@pytest.mark.parametrize("mfma", [True, False])
@pytest.mark.parametrize("logqL,logql", [(300, 300), (300, 180), (438, 438), (438, 88), (610, 610), (610, 350), (850, 850), (850, 400), (1000, 1000)])
def test_he_mul_and_he_swk_sweep(engine_ctx, oracle_ctx, logqL, logql, mfma):
    torch = _torch()
    probe = engine_ctx(LOGN, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    g, o = engine_ctx(LOGN, dimevk), oracle_ctx(LOGN, dimevk)
    assert ref.he_dims(LOGN, o.p, logqL, logql) == (dimP, dimA, dimB, dimevk)
    g.set_bridge_mfma(mfma)
    try:
        n, W = g.n, (logql + 64) // 64
        rng = random.Random(logqL * 1000 + logql)
        k0, k1 = o.gen(7000 + logqL, dimevk), o.gen(7001 + logqL, dimevk)
        ct = [_centred(rng, logql, n) for _ in range(4)]
        dev = [to_device(ints_to_big(v, W)) for v in ct]
        o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
        g.he_mul(o0, o1, dev[0], dev[1], dev[2], dev[3], to_device(k0), to_device(k1), W, logql, dimA, dimB, dimP)
        e0, e1 = ref.he_mul(o, (ct[0], ct[1]), (ct[2], ct[3]), k0[: dimB * n], k1[: dimB * n], dimP, dimA, dimB, logql)
        assert big_to_ints(to_host(o0), W, n)[0] == e0
        assert big_to_ints(to_host(o1), W, n)[0] == e1
        g.he_swk(o0, o1, dev[0], dev[1], to_device(k0), to_device(k1), W, logql, dimB, dimP)
        s0, s1 = ref.he_swk(o, ct[0], ct[1], k0[: dimB * n], k1[: dimB * n], dimP, dimB, logql)
        assert big_to_ints(to_host(o0), W, n)[0] == s0
        assert big_to_ints(to_host(o1), W, n)[0] == s1
    finally:
        g.set_bridge_mfma(True)


@pytest.mark.parametrize("dim,logq", [(4, 60), (5, 128), (8, 250), (11, 320), (16, 448), (20, 500), (23, 640), (31, 850), (37, 896), (44, 1000), (58, 1020)])
def test_poly_mul_sweep(engine_ctx, oracle_ctx, dim, logq):
    """poly_mul (src/poly.c:84-107) with q = 2^logq on `dim` limbs: decompose, NTT, product, INTT, CRT -- against the negacyclic
    product of the centred inputs (inputs sized so that the product fits the basis, as the reference's callers guarantee)."""
    torch = _torch()
    g = engine_ctx(LOGN, max(dim, 12))
    n = g.n
    P = 1
    for d in range(dim):
        P *= g.p[d]
    bits = min(logq, (P.bit_length() - LOGN - 3) // 2)       # |a*b| summed over n terms stays below P/2
    W = (max(bits, logq) + 64) // 64
    rng = random.Random(dim * 100 + logq)
    a, b = _centred(rng, bits, n), _centred(rng, bits, n)
    r = torch.empty(W * n, dtype=torch.int64, device="cuda")
    g.poly_mul(r, to_device(ints_to_big(a, W)), to_device(ints_to_big(b, W)), W, dim, logq)
    assert big_to_ints(to_host(r), W, n)[0] == [ref.centred_mod(v, 1 << logq) for v in ref.negacyclic_mul(a, b)]


@pytest.mark.parametrize("logql,lognu", [(88, 30), (200, 40), (438, 50), (610, 50), (850, 50), (1000, 60)])
def test_he_mulpt_and_he_rs_sweep(engine_ctx, oracle_ctx, logql, lognu):
    """he_mulpt (src/he-mult.c:159-196; dim from log2(pt->nu) as :169) followed by he_rs (src/he-rescale.c:33-54)"""
    torch = _torch()
    dim = (logql + 1 + lognu + LOGN) // 59 + 1
    g, o = engine_ctx(LOGN, max(dim, 20)), oracle_ctx(LOGN, max(dim, 20))
    n, W = g.n, (logql + 64) // 64
    rng = random.Random(logql + lognu)
    ct = [_centred(rng, logql, n) for _ in range(2)]
    m = [rng.randrange(-(1 << lognu), 1 << lognu) for _ in range(n)]
    d0, d1, dm = (to_device(ints_to_big(v, W)) for v in (ct[0], ct[1], m))
    o0, o1 = torch.empty_like(d0), torch.empty_like(d0)
    g.he_mulpt(o0, o1, d0, d1, dm, W, logql, dim)
    e0, e1 = ref.he_mulpt(o, ct, m, dim, logql)
    assert big_to_ints(to_host(o0), W, n)[0] == e0 and big_to_ints(to_host(o1), W, n)[0] == e1
    s = min(lognu, logql - 2)
    g.he_rs(o0, o1, W, s, logql - s)                       # Delta = 2^s, q_{l-1} = 2^(logql - s)
    r0 = [ref.mpi_smod(ref.mpi_rdiv(v, 1 << s), 1 << (logql - s)) for v in e0]
    r1 = [ref.mpi_smod(ref.mpi_rdiv(v, 1 << s), 1 << (logql - s)) for v in e1]
    assert big_to_ints(to_host(o0), W, n)[0] == r0 and big_to_ints(to_host(o1), W, n)[0] == r1


def test_bridge_random_shapes(engine_ctx):
    """Seeded random (limbs, words, first limb) shapes at n = 64: matrix-core and VALU rns_decompose agree with each other and
    with Python's floor mod; poly_rns2mpi of the result gives the centred value back."""
    torch = _torch()
    g = engine_ctx(LOGN, 58)
    n = g.n
    rng = random.Random(20261004)
    for _ in range(24):
        W = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 13, 14, 16, 17, 24, 31, 32])
        dim = rng.randrange(4, 59)
        bits = rng.randrange(8, 64 * W)
        vals = [rng.randrange(-(1 << (bits - 1)), 1 << (bits - 1)) for _ in range(n)]
        vals[:4] = [0, -1, (1 << (bits - 1)) - 1, -(1 << (bits - 1))]
        big = to_device(ints_to_big(vals, W))
        outs = []
        try:
            for mfma in (True, False):
                g.set_bridge_mfma(mfma)
                slab = torch.empty(dim * n, dtype=torch.int64, device="cuda")
                g.rns_decompose(slab, big, W, dim)
                outs.append(slab)
        finally:
            g.set_bridge_mfma(True)
        assert torch.equal(outs[0], outs[1]), (W, dim, bits)
        host = to_host(outs[0]).reshape(dim, n)
        for d in (0, dim // 2, dim - 1):
            assert [int(x) for x in host[d]] == [v % g.p[d] for v in vals], (W, dim, d)
        P = 1
        for d in range(dim):
            P *= g.p[d]
        logq = bits + 1
        if P.bit_length() >= 160 and 2 * logq < P.bit_length() and logq <= 1024:      # the values are below P/2: CRT returns them
            Wq = (logq + 63) // 64
            back = torch.empty(Wq * n, dtype=torch.int64, device="cuda")
            g.rns_reconstruct(back, Wq, outs[0], dim, logq)
            assert big_to_ints(to_host(back), Wq, n)[0] == vals, (W, dim, bits)
