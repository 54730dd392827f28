"""The reference's own known-answer test for this path: tests/polymul.c + tests/polymul.gp
(polyctx_init(7, 2^61): n = 128, 5 limbs, q = 2^61), BASELINE.json configs[0].

The reference prints the product for a human to diff against PARI/GP.  Here the
expectation is the same independent computation (negacyclic product over the integers,
reduced mod 2^61 and centred, tests/polymul.gp:4-10), the MPI<->RNS bridge is the
Python-integer restatement in oracle/bigint_ref.py, and the RNS limb loop runs
(a) in the oracle (CPU tier) and (b) on the GPU through the C ABI (gpu tier)."""
import numpy as np
import pytest

from oracle.bigint_ref import RnsBasis, centred_mod, negacyclic_mul, poly_rns2mpi, rns_decompose

N, DIM, Q = 128, 5, 1 << 61


def _inputs(primes):
    # tests/polymul.c:60-63 and :69-74
    case1 = ([i + 2 for i in range(N)], [i + 3 for i in range(N)])
    case2 = ([primes[0] - i - 1 for i in range(N)], [primes[1] - i - 1 for i in range(N)])
    return [case1, case2]


def _expected(a, b):
    return [centred_mod(v, Q) for v in negacyclic_mul(a, b)]


def _slab(a, primes):
    return np.array([v for p in primes for v in rns_decompose(a, p)], dtype=np.uint64)


def test_bigint_bridge_pinned_to_reference_constants(golden, oracle_ctx):
    """phat_invmp for every prefix of the logn=7 chain, as printed by tests/polymul.c:106-112."""
    o = oracle_ctx(7, DIM)
    for dim, exp in enumerate(golden["phat_invmp_logn7"], start=1):
        assert [str(v) for v in RnsBasis(o.p[:dim]).phat_invmp] == exp


def test_polymul_kat_leading_terms_cpu(oracle_ctx):
    o = oracle_ctx(7, DIM)
    basis = RnsBasis(o.p)
    leading = [[382784, 357372, 332350], [18559595904, 18272672062, 17985699960]]  # SURVEY.md 8c, from the reference run
    for (a, b), lead in zip(_inputs(o.p), leading):
        r = o.poly_mul_rns(_slab(a, o.p), _slab(b, o.p), DIM)  # src/poly.c:96-103
        got = poly_rns2mpi([r[d * N:(d + 1) * N] for d in range(DIM)], basis, Q)
        assert got == _expected(a, b)
        assert [got[127], got[126], got[125]] == lead


@pytest.mark.gpu
def test_polymul_kat_gpu(engine_ctx, oracle_ctx):
    from gpqhe_amd import to_device, to_host
    import torch
    g, o = engine_ctx(7, DIM), oracle_ctx(7, DIM)
    basis = RnsBasis(g.p)
    for a, b in _inputs(g.p):
        da, db = to_device(_slab(a, g.p)), to_device(_slab(b, g.p))
        r = torch.empty_like(da)
        g.poly_mul_rns(r, da, db, DIM)
        rh = to_host(r)
        assert poly_rns2mpi([rh[d * N:(d + 1) * N] for d in range(DIM)], basis, Q) == _expected(a, b)


# ---- BASELINE.json configs[0] as it is worded: "single poly_mul at N = 2^12, 1 RNS prime" ------------------------------------------------
# One 60-bit prime carries the product exactly when every coefficient of a*b mod x^n+1 stays below p/2 in magnitude: signed 20-bit
# operands give |coefficient| <= 4096 * 2^40 = 2^52.  q = 2^55, so the centred reductions of poly_mul (src/poly.c:84-107, 109-120) leave
# the integer product itself, which an independent int64 convolution provides (tests/polymul.gp's formula at this size).
N12, Q12 = 1 << 12, 1 << 55


def _inputs12():
    rng = np.random.default_rng(2012)
    a = rng.integers(-(1 << 20) + 1, 1 << 20, N12, dtype=np.int64)
    b = rng.integers(-(1 << 20) + 1, 1 << 20, N12, dtype=np.int64)
    a[:3], b[:3] = [(1 << 20) - 1, -(1 << 20) + 1, 0], [-(1 << 20) + 1, -(1 << 20) + 1, 1]     # the extremes ride along
    full = np.convolve(a, b)                                          # exact in int64: |sum| <= 2^52
    neg = full[:N12].copy()
    neg[: N12 - 1] -= full[N12:]                                      # x^n = -1
    return [int(v) for v in a], [int(v) for v in b], [int(v) for v in neg]


def test_polymul_n4096_one_prime_cpu(oracle_ctx):
    o = oracle_ctx(12, 1)
    a, b, exp = _inputs12()
    assert max(abs(v) for v in exp) < min(o.p[0], Q12) // 2
    r = o.poly_mul_rns(_slab(a, o.p[:1]), _slab(b, o.p[:1]), 1)       # src/poly.c:96-103 with dim = 1
    assert poly_rns2mpi([r], RnsBasis(o.p[:1]), Q12) == exp


@pytest.mark.gpu
def test_polymul_n4096_one_prime_gpu(engine_ctx):
    """The same through the device's whole poly_mul (decompose, NTT, product, INTT, CRT + both centrings on big slabs)."""
    import torch
    from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host
    g = engine_ctx(12, 1)
    a, b, exp = _inputs12()
    W = 1
    da, db = to_device(ints_to_big(a, W)), to_device(ints_to_big(b, W))
    r = torch.empty_like(da)
    g.poly_mul(r, da, db, W, 1, 55)
    assert big_to_ints(to_host(r), W, N12)[0] == exp
