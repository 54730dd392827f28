"""Pin of the oracle's big-integer semantics against libgcrypt ITSELF ("do this" item 2).

`oracle/bigint_ref.mpi_smod` / `mpi_rdiv` restate /root/reference/src/types.c:108-128 with Python integers; until now that was "read off the source".
This test executes the reference's two functions line by line on the library the reference links -- `gcry_mpi_mod`, `gcry_mpi_cmp`, `gcry_mpi_sub`,
`gcry_mpi_div(q, r, a, m, -1)` (= the `mpi_fdiv` macro), `gcry_mpi_add_ui` through ctypes on the image's runtime `libgcrypt.so.20` (no header is
needed or written: these are the public entry points, declared by hand like tests/c/mpi_host.c:28-49 does) -- and compares with the restatement:

  * mpi_smod: every sign (gcry_mpi_mod is correct for negative dividends in 1.9.4), moduli up to 2.7 kbit, both 2^k and odd;
  * mpi_rdiv: NON-NEGATIVE dividends (correct in 1.9.4), divisors P = products of up to 45 sixty-bit primes and Delta = 2^s, ties and near-ties;
  * mpi_rdiv on NEGATIVE dividends: libgcrypt 1.9.4's floor division returns |q| there (survey: -1000503 fdiv 1000 -> q = +1001, r = 497).  On such a
    library the case is recorded as what it is -- the defect reproduces, the restatement (mathematical floor, what 1.10 does and README.md:29 asks
    for) differs, and nothing on this image can pin that half of the domain; on a fixed library the same inputs must agree with the restatement.

Nothing here touches the product or the reference's sources; the oracle stays test infrastructure."""
import ctypes as C
import ctypes.util
import random

import pytest

from oracle import bigint_ref

FMT_HEX = 4


def _load():
    for name in ("libgcrypt.so.20", ctypes.util.find_library("gcrypt")):
        if not name:
            continue
        try:
            return C.CDLL(name)
        except OSError:
            continue
    return None


L = _load()
pytestmark = pytest.mark.skipif(L is None, reason="no libgcrypt runtime on this machine")

if L is not None:
    MPI = C.c_void_p
    L.gcry_check_version.restype = C.c_char_p
    L.gcry_check_version.argtypes = [C.c_char_p]
    L.gcry_mpi_new.restype = MPI
    L.gcry_mpi_new.argtypes = [C.c_uint]
    L.gcry_mpi_release.argtypes = [MPI]
    L.gcry_mpi_set_ui.restype = MPI
    L.gcry_mpi_set_ui.argtypes = [MPI, C.c_ulong]
    L.gcry_mpi_scan.restype = C.c_uint
    L.gcry_mpi_scan.argtypes = [C.POINTER(MPI), C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.gcry_mpi_aprint.restype = C.c_uint
    L.gcry_mpi_aprint.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), MPI]
    L.gcry_free.argtypes = [C.c_void_p]
    L.gcry_mpi_div.argtypes = [MPI, MPI, MPI, MPI, C.c_int]
    L.gcry_mpi_mod.argtypes = [MPI, MPI, MPI]
    L.gcry_mpi_cmp.restype = C.c_int
    L.gcry_mpi_cmp.argtypes = [MPI, MPI]
    L.gcry_mpi_sub.argtypes = [MPI, MPI, MPI]
    L.gcry_mpi_add_ui.argtypes = [MPI, MPI, C.c_ulong]
    L.gcry_mpi_is_neg.restype = C.c_int
    L.gcry_mpi_is_neg.argtypes = [MPI]
    VERSION = L.gcry_check_version(None).decode()


def to_mpi(v):
    m = MPI()
    text = ("-" if v < 0 else "") + "0" + format(abs(v), "X")     # a leading 0 keeps the HEX scanner from reading a sign bit
    assert L.gcry_mpi_scan(C.byref(m), FMT_HEX, text.encode(), 0, None) == 0
    return m


def to_int(m):
    buf = C.c_void_p()
    assert L.gcry_mpi_aprint(FMT_HEX, C.byref(buf), None, m) == 0
    s = C.string_at(buf).decode()
    L.gcry_free(buf)
    return int(s, 16)


def gcry_smod(r, q):
    """src/types.c:108-113 as written: mpi_mod(r, r, q); if (mpi_cmp(r, qh) >= 0) mpi_sub(r, r, q);  with qh = floor(q/2) as the callers pass it
    (hectx.qh[l], src/precomp.c:399-401; P_2, src/precomp.c:271)."""
    R, Q, QH = to_mpi(r), to_mpi(q), to_mpi(q // 2)
    L.gcry_mpi_mod(R, R, Q)
    if L.gcry_mpi_cmp(R, QH) >= 0:
        L.gcry_mpi_sub(R, R, Q)
    out = to_int(R)
    for m in (R, Q, QH):
        L.gcry_mpi_release(m)
    return out


def gcry_rdiv(a, m):
    """src/types.c:115-128 as written: mh = fdiv(m, 2); fdiv(q, r, a, m); if (mpi_cmp(r, mh) > 0) q += 1   (mpi_fdiv = gcry_mpi_div(..., -1))."""
    A, M, TWO = to_mpi(a), to_mpi(m), to_mpi(2)
    assert not L.gcry_mpi_is_neg(M)
    MH, R, Q = L.gcry_mpi_new(0), L.gcry_mpi_new(0), L.gcry_mpi_new(0)
    L.gcry_mpi_div(MH, None, M, TWO, -1)
    L.gcry_mpi_div(Q, R, A, M, -1)
    if L.gcry_mpi_cmp(R, MH) > 0:
        L.gcry_mpi_add_ui(Q, Q, 1)
    out = to_int(Q)
    for x in (A, M, TWO, MH, R, Q):
        L.gcry_mpi_release(x)
    return out


def _moduli(golden):
    """the divisors the hot path really uses: P over 15 / 30 / 45 limbs of the n = 2^16 chain (he_relin's mpi_rdiv by P, src/he-mult.c:70-71;
    centring mod P, src/poly.c:116), powers of two (Delta = 2^50 of he_rs, src/he-rescale.c:45-48; q_l = 2^(50 l), centring, src/poly.c:118)"""
    primes = [int(p) for p in golden["prime_chain"]["16"]["first"]]
    from oracle.oracle import OracleCtx
    chain = OracleCtx(16, 45).p
    assert [int(p) for p in chain[:4]] == primes              # the chain the survey captured from the compiled reference
    out = []
    for dim in (1, 2, 15, 30, 45):
        P = 1
        for p in chain[:dim]:
            P *= int(p)
        out.append(P)
    out += [1 << s for s in (1, 17, 50, 63, 64, 100, 850)]
    out += [3, 1000, (3 << 61) + 7]
    return out


def test_roundtrip_of_the_conversion_helpers():
    rng = random.Random(1)
    for bits in (0, 1, 7, 8, 63, 64, 65, 886, 2700):
        for sign in (1, -1):
            v = sign * rng.getrandbits(bits) if bits else 0
            m = to_mpi(v)
            assert to_int(m) == v
            L.gcry_mpi_release(m)


def test_mpi_smod_matches_libgcrypt_for_every_sign(golden):
    rng = random.Random(2)
    checked = 0
    for q in _moduli(golden):
        cases = [0, 1, -1, q // 2 - 1, q // 2, q // 2 + 1, q - 1, q, q + 1, -(q // 2), -(q // 2) - 1, -(q // 2) + 1, -q, 1 - q, -q - 1]
        cases += [rng.randrange(-(q << 70), q << 70) for _ in range(40)]
        cases += [rng.getrandbits(2700), -rng.getrandbits(2700)]
        for r in cases:
            assert gcry_smod(r, q) == bigint_ref.mpi_smod(r, q), (r, q)
            checked += 1
    assert checked > 800


def test_mpi_rdiv_matches_libgcrypt_on_non_negative_dividends(golden):
    """floor + "remainder > floor(m/2)" (ties round DOWN for even m, src/types.c:124), up to 2.7-kbit dividends (x < P q_l of he_relin, src/he-mult.c:70)"""
    rng = random.Random(3)
    checked = 0
    for m in _moduli(golden):
        h = m // 2
        cases = [0, 1, h - 1, h, h + 1, m - 1, m, m + 1, m + h - 1, m + h, m + h + 1, 7 * m + h, 7 * m + h + 1, 7 * m + h - 1]
        cases += [rng.randrange(0, m << 80) for _ in range(40)]
        cases += [rng.getrandbits(2700) for _ in range(3)]
        for a in cases:
            if a < 0:
                continue
            assert gcry_rdiv(a, m) == bigint_ref.mpi_rdiv(a, m), (a, m)
            checked += 1
    assert checked > 800


def test_mpi_rdiv_on_negative_dividends_is_recorded_for_what_this_library_does():
    """libgcrypt 1.9.4 mis-signs floor division of a negative dividend (-1000503 fdiv 1000 -> q = +1001, r = 497), so the reference
    linked against THIS library computes wrong he_relin / he_rs results for negative coefficients (its own tests/gpqhe.c `mul` fails on it) and nothing
    here can stand behind `bigint_ref.mpi_rdiv` for a < 0 -- that half is pinned by restatement of src/types.c:115-128 with the mathematical floor the
    source asks for (and README.md:29's libgcrypt 1.10 delivers), by the polymul KAT and by the CRT walk only.  A library that gets the survey's case
    right must agree with the restatement on the whole negative domain as well."""
    a, m = -1000503, 1000
    want = bigint_ref.mpi_rdiv(a, m)
    assert want == -1001                                           # floor(-1000503 / 1000) = -1001, remainder 497 <= 500: no increment
    A, M, Q, R = to_mpi(a), to_mpi(m), L.gcry_mpi_new(0), L.gcry_mpi_new(0)
    L.gcry_mpi_div(Q, R, A, M, -1)
    q, r = to_int(Q), to_int(R)
    for x in (A, M, Q, R):
        L.gcry_mpi_release(x)
    if (q, r) == (-1001, 497):                                     # a correct floor division: the whole domain can be pinned
        rng = random.Random(4)
        for _ in range(400):
            mm = rng.choice([1000, 1 << 50, (1 << 59) + 7471105, (3 << 61) + 7])
            aa = -rng.randrange(0, mm << 70)
            assert gcry_rdiv(aa, mm) == bigint_ref.mpi_rdiv(aa, mm), (aa, mm)
    else:                                                          # the defect, exactly as the survey observed it
        assert (q, r) == (1001, 497), "libgcrypt %s: an unknown floor-division behaviour: q = %d, r = %d" % (VERSION, q, r)
        assert gcry_rdiv(a, m) == 1001 and gcry_rdiv(a, m) != want
        # What the defective library still pins on the negative half: the floor REMAINDER (the quantity src/types.c:124 compares with floor(m/2))
        # and the MAGNITUDE of the floor quotient are right for every negative dividend -- only the sign of a non-zero quotient with a non-zero
        # remainder is lost (the library's sub_ui on a negative value, behind its truncating division).  So the restatement's floor(a/m) and a mod m
        # agree with libgcrypt's limbs on the whole domain, and the one thing taken from the source text alone is "the quotient keeps its sign".
        rng = random.Random(5)
        lost = 0
        for _ in range(1500):
            mm = rng.choice([1000, 1 << 50, (1 << 59) + 7471105, (3 << 61) + 7, rng.getrandbits(rng.choice([5, 64, 65, 128, 886, 2650])) | 1])
            aa = -rng.randrange(0, mm << rng.choice([1, 10, 70, 200]))
            A, M, Q, R = to_mpi(aa), to_mpi(mm), L.gcry_mpi_new(0), L.gcry_mpi_new(0)
            L.gcry_mpi_div(Q, R, A, M, -1)
            qq, rr = to_int(Q), to_int(R)
            for x in (A, M, Q, R):
                L.gcry_mpi_release(x)
            assert rr == aa % mm and abs(qq) == abs(aa // mm), (aa, mm, qq, rr)
            lost += qq != aa // mm
        assert lost > 1000
        pytest.xfail("libgcrypt %s floor-divides -1000503 by 1000 to q = +1001 (r = 497): the SIGN of mpi_rdiv's quotient on negative dividends cannot "
                     "be pinned to this library (remainder and magnitude are: 1500 cases); bigint_ref keeps the mathematical floor of "
                     "src/types.c:115-128" % VERSION)


def test_rns_decompose_and_poly_rns2mpi_match_libgcrypt(golden):
    """src/rns.c:37-48 (mpi_mod by the prime: non-negative for negative coefficients) and src/rns.c:60-75 + src/poly.c:109-120 (three mulm / addm
    per limb, then the two centrings) executed on libgcrypt for a handful of coefficients at the hot path's real sizes -- 30 and 45 limbs of the
    n = 2^16 chain, q_l = 2^850 / 2^250 -- against oracle/bigint_ref's restatement (which keeps (phat_d * phat_invmp_d) mod P per basis)."""
    L.gcry_mpi_mulm.argtypes = [MPI, MPI, MPI, MPI]
    L.gcry_mpi_addm.argtypes = [MPI, MPI, MPI, MPI]
    from oracle.oracle import OracleCtx
    chain = [int(p) for p in OracleCtx(16, 45).p]
    rng = random.Random(6)
    for dim, logq in ((1, 59), (2, 100), (30, 850), (45, 850), (45, 250)):
        basis = bigint_ref.RnsBasis(chain[:dim])
        q = 1 << logq
        # rns_decompose of centred coefficients, negative ones included
        coeffs = [rng.randrange(-(q // 2), q // 2) for _ in range(6)] + [0, -1, 1, -(q // 2), q // 2 - 1]
        for p in (chain[0], chain[dim - 1]):
            Pm = to_mpi(p)
            for a in coeffs:
                A, B = to_mpi(a), L.gcry_mpi_new(0)
                L.gcry_mpi_mod(B, A, Pm)
                assert to_int(B) == bigint_ref.rns_decompose([a], p)[0]
                L.gcry_mpi_release(A); L.gcry_mpi_release(B)
            L.gcry_mpi_release(Pm)
        # poly_rns2mpi of random residues
        n = 5
        limbs = [[rng.randrange(0, chain[d]) for _ in range(n)] for d in range(dim)]
        want = bigint_ref.poly_rns2mpi(limbs, basis, q)
        Pm, P2, Qm, Qh = to_mpi(basis.P), to_mpi(basis.P_2), to_mpi(q), to_mpi(q // 2)
        phat = [to_mpi(v) for v in basis.phat]
        for i in range(n):
            a, b, c = L.gcry_mpi_new(0), L.gcry_mpi_new(0), L.gcry_mpi_new(0)
            L.gcry_mpi_set_ui(a, 0)
            for d in range(dim):
                L.gcry_mpi_set_ui(b, limbs[d][i])
                L.gcry_mpi_set_ui(c, basis.phat_invmp[d])
                L.gcry_mpi_mulm(c, phat[d], c, Pm)
                L.gcry_mpi_mulm(b, b, c, Pm)
                L.gcry_mpi_addm(a, a, b, Pm)
            for mod, half in ((Pm, P2), (Qm, Qh)):                 # mpi_smod twice, src/poly.c:116-118
                L.gcry_mpi_mod(a, a, mod)
                if L.gcry_mpi_cmp(a, half) >= 0:
                    L.gcry_mpi_sub(a, a, mod)
            assert to_int(a) == want[i], (dim, logq, i)
            for x in (a, b, c):
                L.gcry_mpi_release(x)
        for x in [Pm, P2, Qm, Qh] + phat:
            L.gcry_mpi_release(x)
