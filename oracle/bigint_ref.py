"""Big-integer side of the hot path, restated with Python integers.

TEST INFRASTRUCTURE ONLY (same rule as gpqhe_oracle.c).  The reference does this
part with libgcrypt MPIs; Python ints have the same mathematical semantics
(floor division / non-negative mod for positive moduli), so each function is a
line-by-line restatement of:
  rns_init         src/precomp.c:266-293   P, P/2, phat[d] = P/p_d, phat_invmp[d]
  rns_decompose    src/rns.c:37-48         a[i] mod p   (floor mod: non-negative)
  rns_reconstruct  src/rns.c:60-75         sum ahat[d][i] * ((phat_d*phat_invmp_d) mod P) mod P
  mpi_smod         src/types.c:108-113     r mod q, minus q when r >= floor(q/2)
  mpi_rdiv         src/types.c:115-128     floor(a/m), plus one when the remainder > floor(m/2)
  poly_rns2mpi     src/poly.c:109-120      reconstruct, centre mod P, centre mod q
  poly_mul         src/poly.c:84-107       (the RNS limb loop itself is supplied by the caller)
Pinned by the phat_invmp values SURVEY.md 8c took from tests/polymul.c output.
Pure Python: small cases only.
"""


class RnsBasis:
    """The prefix of `dim` primes of a prime chain: what struct rns_ctx node dim-1 holds."""

    def __init__(self, primes):
        self.p = [int(x) for x in primes]
        self.dim = len(self.p)
        self.P = 1
        for x in self.p:
            self.P *= x
        self.P_2 = self.P // 2
        self.phat = [self.P // x for x in self.p]
        self.phat_invmp = [pow(self.phat[d] % self.p[d], self.p[d] - 2, self.p[d]) for d in range(self.dim)]


def rns_decompose(a, p):
    return [int(v) % p for v in a]


def rns_reconstruct(ahat_limbs, i, basis):
    acc = 0
    for d in range(basis.dim):
        c = (basis.phat[d] * basis.phat_invmp[d]) % basis.P
        acc = (acc + (int(ahat_limbs[d][i]) * c) % basis.P) % basis.P
    return acc


def mpi_smod(r, q):
    r %= q
    return r - q if r >= q // 2 else r


def mpi_rdiv(a, m):
    assert m > 0
    q, r = divmod(a, m)  # floor division, as mpi_fdiv
    return q + 1 if r > m // 2 else q


def poly_rns2mpi(rhat_limbs, basis, q):
    n = len(rhat_limbs[0])
    return [mpi_smod(mpi_smod(rns_reconstruct(rhat_limbs, i, basis), basis.P), q) for i in range(n)]


def negacyclic_mul(a, b):
    """a*b mod x^n+1 over the integers: the independent expectation of tests/polymul.gp."""
    n = len(a)
    r = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                r[k] += ai * bj
            else:
                r[k - n] -= ai * bj
    return r


def centred_mod(v, q):
    """liftall(Mod(v, q)) then minus q where >= q/2, as tests/polymul.gp:7-10."""
    v %= q
    return v - q if v >= q // 2 else v
