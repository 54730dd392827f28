"""The last fixture the reference's own tests hold for this path, on the CPU tier: the walk of tests/crt.c:34-225.

tests/crt.c is built with -DTEST_CRT, which sets GPQHE_LOGP = 9 (src/params.h:30-33): polyctx_init(4, 2^10) then makes
a chain of six 10-bit primes == 1 (mod 2n = 32) above 2^9 + 1 (src/precomp.c:357-376), fills a[i] = P[2] - i - 1 (P[2] =
the product of the first three primes, :82-85), and for dim = 6, 5, 4, 3, 2 decomposes a over the first dim primes
(rns_decompose, src/rns.c:37-48) and reconstructs it from the node of that dim (rns_reconstruct, src/rns.c:60-75),
printing everything.  The file asserts nothing; what it demonstrates -- and what is checked here against
oracle/bigint_ref.py, the restatement the MPI-level GPU tests use as their expectation -- is
  * decompose -> reconstruct is the identity while a < P (dim >= 3) and a mod P once it is not (dim = 2),
  * reconstructed values lie in [0, P), residues in [0, p_d),
  * the CRT constants rns_init builds (src/precomp.c:266-293): phat[d] = P / p_d, phat[d] * phat_invmp[d] == 1 (mod p_d),
  * `a` is carried from one round of the walk into the next (crt.c reuses a.coeffs), so dim = 2 sees the dim = 3 result.
The 9-bit primes are outside the 2^59 + c family the HIP kernels fold, so this is a CPU-tier test of the oracle only."""
from oracle.bigint_ref import RnsBasis, rns_decompose, rns_reconstruct

LOGP, LOGN, LOGQ = 9, 4, 10          # src/params.h:32, tests/crt.c:36,41


def _is_prime(v):
    return v > 1 and all(v % d for d in range(2, int(v ** 0.5) + 1))


def _test_crt_chain():
    """polyctx_init's prime loop with GPQHE_LOGP = 9: src/precomp.c:357-376 (logn = 4 < 10, so logqub = logq, :338-340)."""
    n = 1 << LOGN
    dimub = (1 + LOGN + 4 * LOGQ) // LOGP + 1
    p, chain = (1 << LOGP) + 1, []
    while len(chain) < dimub:
        p += 2 * n
        if _is_prime(p):
            chain.append(p)
    return chain


def test_test_crt_prime_chain():
    chain = _test_crt_chain()
    assert chain == [577, 641, 673, 769, 929, 1153]                       # "dim_max=6" in tests/crt.c:92
    assert all(p % 32 == 1 and p.bit_length() == 10 + (p >= 1024) for p in chain)


def test_rns_init_constants_for_every_prefix():
    chain = _test_crt_chain()
    for dim in range(1, len(chain) + 1):
        b = RnsBasis(chain[:dim])
        assert b.P_2 == b.P // 2
        for d in range(dim):
            assert b.phat[d] * chain[d] == b.P                             # src/precomp.c:281
            assert 0 < b.phat_invmp[d] < chain[d] and b.phat[d] * b.phat_invmp[d] % chain[d] == 1   # :287-290
    assert RnsBasis(chain[:1]).phat_invmp == [1]                           # as for the 59-bit chain


def test_crt_walk_dims_6_to_2():
    chain = _test_crt_chain()
    n = 1 << LOGN
    P2 = RnsBasis(chain[:3]).P                                              # polyctx.rns->next->next->P
    assert P2 == 577 * 641 * 673
    a = [P2 - i - 1 for i in range(n)]                                      # tests/crt.c:84
    start = list(a)
    for dim in (6, 5, 4, 3, 2):                                             # tests/crt.c:93-216
        basis = RnsBasis(chain[:dim])
        ahat = [rns_decompose(a, chain[d]) for d in range(dim)]             # one limb per prime, limb-major
        for d in range(dim):
            assert all(0 <= v < chain[d] for v in ahat[d])
            assert ahat[d] == [v % chain[d] for v in a]
        back = [rns_reconstruct(ahat, i, basis) for i in range(n)]
        assert all(0 <= v < basis.P for v in back)
        if dim >= 3:
            assert back == start                                            # a < P[2] <= P: identity
        else:
            assert basis.P == 577 * 641 < min(start)
            assert back == [v % basis.P for v in start]                     # wraps: a mod P[1]
        a = back                                                            # crt.c reuses a.coeffs for the next round


def test_negative_coefficients_decompose_by_floor_mod():
    """src/rns.c:37-48 uses mpi_mod (non-negative result): centred negative coefficients keep decompose -> reconstruct ->
    centre an identity, which is what poly_rns2mpi (src/poly.c:109-120) relies on."""
    from oracle.bigint_ref import mpi_smod
    chain = _test_crt_chain()
    basis = RnsBasis(chain[:4])
    vals = [-(basis.P // 2) + 1, -12345, -1, 0, 1, 12345, basis.P // 2 - 1]
    ahat = [rns_decompose(vals, p) for p in chain[:4]]
    assert all(0 <= v < p for limb, p in zip(ahat, chain) for v in limb)
    assert [mpi_smod(rns_reconstruct(ahat, i, basis), basis.P) for i in range(len(vals))] == vals
