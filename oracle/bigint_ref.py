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
Pinned by the phat_invmp values SURVEY.md 8c took from tests/polymul.c output, by the polymul KAT and the tests/crt.c walk, and
(round 6) against libgcrypt ITSELF: tests/test_libgcrypt_pin.py runs mpi_smod, mpi_rdiv, rns_decompose and poly_rns2mpi line by line on
the image's libgcrypt.so.20 -- every sign for mpi_smod, non-negative dividends for mpi_rdiv (libgcrypt 1.9.4 loses the quotient's sign on
negative ones; its remainder and |quotient| still agree with this file), CRT at 30 / 45 limbs.
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
        # (phat_d * phat_invmp_d) mod P: the first mpi_mulm of src/rns.c:60-75 depends on d only; kept per basis so that the
        # full-size expectations (65,536 coefficients x 45 limbs) stay within a test's time
        self.crt_coef = [(self.phat[d] * self.phat_invmp[d]) % self.P for d in range(self.dim)]


def rns_decompose(a, p):
    return [int(v) % p for v in a]


def rns_reconstruct(ahat_limbs, i, basis):
    acc = 0
    for d in range(basis.dim):
        acc = (acc + (int(ahat_limbs[d][i]) * basis.crt_coef[d]) % basis.P) % basis.P
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


# ---------------------------------------------------------------------------
# he_mul / he_relin / he_swk with Python integers around the C oracle's RNS core
# ---------------------------------------------------------------------------
import numpy as _np


def he_dims(logn, primes, logqL, logql):
    """hectx.dim (src/precomp.c:401), he_mul dim (src/he-mult.c:99), he_relin dim (:51), dimevk (src/precomp.c:407)
    for q_L = 2^logqL, q_l = 2^logql; nbits(2^k) = k+1."""
    dimP = (logqL + 1 + logn) // 59 + 1
    P = 1
    for x in primes[:dimP]:
        P *= int(x)
    nb_PqL = (P << logqL).bit_length()
    dimA = (2 * (logql + 1) + logn) // 59 + 1
    dimB = ((logql + 1) + nb_PqL + logn) // 59 + 1
    dimevk = ((logqL + 1) + nb_PqL + logn) // 59 + 1
    return dimP, dimA, dimB, dimevk


def _slab(o, coeffs, dim):
    return _np.array([v for d in range(dim) for v in rns_decompose(coeffs, o.p[d])], dtype=_np.uint64)


def _limbs(slab, dim, n):
    return [slab[d * n:(d + 1) * n] for d in range(dim)]


def he_relin_tail(o, c0hat, c1hat, d0, d1, dimP, dimB, ql):
    """src/he-mult.c:67-77 (d1 may be None: he_swk, src/he-automorphism.c:68-76)."""
    n = o.n
    P = RnsBasis(o.p[:dimP]).P
    basisB = RnsBasis(o.p[:dimB])
    out = []
    for chat, dd in ((c0hat, d0), (c1hat, d1)):
        c = poly_rns2mpi(_limbs(chat, dimB, n), basisB, P * ql)          # :67-68
        res = []
        for i in range(n):
            v = mpi_rdiv(c[i], P)                                        # :71-72
            if dd is not None:
                v = (v + dd[i]) % ql                                     # :73-74  mpi_addm
            res.append(mpi_smod(v, ql))                                  # :75-76
        out.append(res)
    return out


def he_mul(o, ct1, ct2, rlk0, rlk1, dimP, dimA, dimB, logql, ql=None):
    """src/he-mult.c:88-156 for q_l = 2^logql (or any integer ql); ct = (c0, c1) lists of centred ints; o = oracle.OracleCtx."""
    n = o.n
    ql = ql if ql is not None else 1 << logql
    ins = [_slab(o, c, dimA) for c in (ct1[0], ct1[1], ct2[0], ct2[1])]  # :117-120
    d0h, d1h, d2h = o.he_mul_tensor(*ins, dimA)                          # :121-136
    basisA = RnsBasis(o.p[:dimA])
    d0 = poly_rns2mpi(_limbs(d0h, dimA, n), basisA, ql)                  # :139
    d2 = poly_rns2mpi(_limbs(d2h, dimA, n), basisA, ql)
    d1 = poly_rns2mpi(_limbs(d1h, dimA, n), basisA, ql)
    c0h, c1h = o.keyswitch(_slab(o, d2, dimB), rlk0, rlk1, dimB)         # :59-64
    return he_relin_tail(o, c0h, c1h, d0, d1, dimP, dimB, ql)


def he_swk(o, d0, d1, swk0, swk1, dimP, dimB, logql, ql=None):
    """src/he-automorphism.c:40-85 for q_l = 2^logql (or any integer ql)."""
    ql = ql if ql is not None else 1 << logql
    c0h, c1h = o.keyswitch(_slab(o, d1, dimB), swk0, swk1, dimB)
    return he_relin_tail(o, c0h, c1h, d0, None, dimP, dimB, ql)


def poly_rot(a, rot):
    """src/poly.c:263-275"""
    n = len(a)
    power = pow(5, rot, 1 << 64)
    r = [0] * n
    for i in range(n):
        k = (i * power) % (2 * n)
        if k < n:
            r[k] = a[i]
        else:
            r[k - n] = -a[i]
    return r


def poly_conj(a):
    """src/poly.c:277-283"""
    n = len(a)
    return [a[0]] + [-a[n - i] for i in range(1, n)]


def he_mulpt(o, ct, m, dim, logql, ql=None):
    """src/he-mult.c:159-196 for q_l = 2^logql (or any integer ql)."""
    n = o.n
    ql = ql if ql is not None else 1 << logql
    basis = RnsBasis(o.p[:dim])
    mh = o.ntt_slab(_slab(o, m, dim), dim)
    out = []
    for c in ct:
        ch = o.ntt_slab(_slab(o, c, dim), dim)
        prod = _np.concatenate([o.rns_mul(ch[d * n:(d + 1) * n], mh[d * n:(d + 1) * n], d) for d in range(dim)])
        out.append(poly_rns2mpi(_limbs(o.ntt_slab(prod, dim, inverse=True), dim, n), basis, ql))
    return out


# src/he-add.c:32-140 -- (c0, c1) pairs of coefficient lists; every result is mpi_smod'ed by q_l (:43-44, :67-68, :91-94, :114-117, :134-137)
def he_add(ct1, ct2, ql):
    return tuple([mpi_smod((x + y) % ql, ql) for x, y in zip(a, b)] for a, b in zip(ct1, ct2))          # mpi_addm, :41-42


def he_sub(ct1, ct2, ql):
    return tuple([mpi_smod((x - y) % ql, ql) for x, y in zip(a, b)] for a, b in zip(ct1, ct2))          # mpi_subm, :65-66


def he_addpt(ct, m, ql):
    return [mpi_smod((x + y) % ql, ql) for x, y in zip(ct[0], m)], [mpi_smod(x % ql, ql) for x in ct[1]]  # :89-90


def he_subpt(ct, m, ql):
    return [mpi_smod((x - y) % ql, ql) for x, y in zip(ct[0], m)], [mpi_smod(x % ql, ql) for x in ct[1]]  # :112-113


def he_neg(ct, ql):
    return tuple([mpi_smod(-x, ql) for x in c] for c in ct)                                               # :132-135


def he_dec_sparse(ct, sk_terms, ql):
    """src/he-encrypt.c:105-125 for a secret key given by its non-zero terms {index: value}: m = c1 * sk + c0, every coefficient
    mpi_smod'ed by q_l (:117-118).  The product is the negacyclic one of poly_mul (src/poly.c:84-107), written out term by term."""
    c0, c1 = ct
    n = len(c0)
    prod = [0] * n
    for j, v in sk_terms.items():
        for i in range(n):
            k = i + j
            if k < n:
                prod[k] += c1[i] * v
            else:
                prod[k - n] -= c1[i] * v
    return [mpi_smod((mpi_smod(p % ql, ql) + c) % ql, ql) for p, c in zip(prod, c0)]

