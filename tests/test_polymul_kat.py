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
