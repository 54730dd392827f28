"""Dense MPI-level parity at the shapes the benchmark reports.

he_mul (src/he-mult.c:88-156) at n = 2^16, q = 2^850 (15 / 30 / 45 limbs) and he_swk (src/he-automorphism.c:40-85) at
n = 2^17, q = 2^835 (44 limbs: BASELINE configs[4]) on DENSE random centred ciphertexts and DENSE random keys: nothing about
the operands is structured, so the fast CRT path's "ambiguous => redo exactly" branch, the relinearisation front's round-bit
fix-up and every carry chain see unstructured data at full size.  Expectation = the restated reference: C oracle for the
RNS limb loops (oracle/gpqhe_oracle.c), Python integers for everything libgcrypt does (oracle/bigint_ref.py), over all
65,536 / 131,072 coefficients."""
import random

import numpy as np
import pytest

from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host
from oracle import bigint_ref as ref

pytestmark = pytest.mark.gpu


def _centred(rng, n, q):
    return [rng.randrange(-(q >> 1), q >> 1) for _ in range(n)]


@pytest.mark.timeout(1200)
def test_he_mul_dense_at_the_headline_shape(engine_ctx, oracle_ctx):
    import torch
    logn, logq = 16, 850
    n, q = 1 << logn, 1 << logq
    dimP, dimA, dimB, dimevk = engine_ctx(logn, 20).he_dims(logq, logq)
    assert (dimP, dimA, dimB, dimevk) == (15, 30, 45, 45)                      # SURVEY.md 8c context dims
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = random.Random(20261004)
    ct = [_centred(rng, n, q) for _ in range(4)]                              # ct1.c0, ct1.c1, ct2.c0, ct2.c1
    for p in ct:                                                               # the extremes of the centred range ride along
        p[:4] = [-(q >> 1), (q >> 1) - 1, 0, -1]
    rlk0, rlk1 = o.gen(3000, dimevk), o.gen(3001, dimevk)                     # dense NTT-domain key limbs (src/he-kem.c:103-110 layout)
    W = logq // 64 + 1
    dev = [to_device(ints_to_big(v, W)) for v in ct]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(o0, o1, *dev, to_device(rlk0[: dimB * n]), to_device(rlk1[: dimB * n]), W, logq, dimA, dimB, dimP)
    torch.cuda.synchronize()
    got0, got1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    exp0, exp1 = ref.he_mul(o, (ct[0], ct[1]), (ct[2], ct[3]), rlk0[: dimB * n], rlk1[: dimB * n], dimP, dimA, dimB, logq)
    bad0 = [i for i in range(n) if got0[i] != exp0[i]]
    bad1 = [i for i in range(n) if got1[i] != exp1[i]]
    assert not bad0 and not bad1, (len(bad0), len(bad1), bad0[:4], bad1[:4])
    assert all(-(q >> 1) <= v < (q >> 1) for v in got0) and len(set(got0)) > n // 2       # centred mod q_l, and not degenerate


@pytest.mark.timeout(1200)
def test_he_swk_dense_at_configs4_shape(engine_ctx, oracle_ctx):
    import torch
    logn, logq = 17, 835
    n, q = 1 << logn, 1 << logq
    dimP = (logq + 1 + logn) // 59 + 1                                          # hectx.dim, src/precomp.c:401
    P = ref.RnsBasis(engine_ctx(logn, dimP).p[:dimP]).P
    dimB = (logq + 1 + (P * q).bit_length() + logn) // 59 + 1                   # src/he-automorphism.c:52
    assert dimB == 44                                                           # BASELINE configs[4]: n = 2^17, 44 limbs
    g, o = engine_ctx(logn, dimB), oracle_ctx(logn, dimB)
    rng = random.Random(17 * 835)
    d0, d1 = _centred(rng, n, q), _centred(rng, n, q)
    d1[:3] = [-(q >> 1), (q >> 1) - 1, 0]
    swk0, swk1 = o.gen(5002, dimB), o.gen(5003, dimB)
    W = logq // 64 + 1
    a0, a1 = to_device(ints_to_big(d0, W)), to_device(ints_to_big(d1, W))
    o0, o1 = torch.empty_like(a0), torch.empty_like(a0)
    g.he_swk(o0, o1, a0, a1, to_device(swk0), to_device(swk1), W, logq, dimB, dimP)
    torch.cuda.synchronize()
    got0, got1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    exp0, exp1 = ref.he_swk(o, d0, d1, swk0, swk1, dimP, dimB, logq)
    bad0 = [i for i in range(n) if got0[i] != exp0[i]]
    bad1 = [i for i in range(n) if got1[i] != exp1[i]]
    assert not bad0 and not bad1, (len(bad0), len(bad1), bad0[:4], bad1[:4])


@pytest.mark.timeout(600)
def test_he_mul_dense_at_the_reference_default_shape(engine_ctx, oracle_ctx):
    """tests/gpqhe.c:1296-1299 -- logn 14, q = 2^438, Delta = 2^50: the only shape the reference itself ever runs, and bench.py's
    `reference_default` leg (16 / 24 limbs, 7 words: the streaming kernels' small instantiations).  Dense random ciphertexts with the
    extremes of the centred range riding along, every coefficient against the restated reference."""
    import torch
    logn, logq = 14, 438
    n, q = 1 << logn, 1 << logq
    dimP, dimA, dimB, dimevk = engine_ctx(logn, 20).he_dims(logq, logq)
    assert (dimA, dimB, dimevk) == (16, 24, 24)                                # SURVEY.md 8c context dims
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = random.Random(14 * 438)
    ct = [_centred(rng, n, q) for _ in range(4)]
    for p in ct:
        p[:4] = [-(q >> 1), (q >> 1) - 1, 0, -1]
    rlk0, rlk1 = o.gen(3000, dimevk), o.gen(3001, dimevk)
    W = logq // 64 + 1
    dev = [to_device(ints_to_big(v, W)) for v in ct]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(o0, o1, *dev, to_device(rlk0[: dimB * n]), to_device(rlk1[: dimB * n]), W, logq, dimA, dimB, dimP)
    torch.cuda.synchronize()
    got0, got1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    exp0, exp1 = ref.he_mul(o, (ct[0], ct[1]), (ct[2], ct[3]), rlk0[: dimB * n], rlk1[: dimB * n], dimP, dimA, dimB, logq)
    bad0 = [i for i in range(n) if got0[i] != exp0[i]]
    bad1 = [i for i in range(n) if got1[i] != exp1[i]]
    assert not bad0 and not bad1, (len(bad0), len(bad1), bad0[:4], bad1[:4])
