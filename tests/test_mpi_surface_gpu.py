"""The MPI-typed reference surface (poly_mul, he_mul, he_rs / he_rescale, he_moddown with the signatures of
src/poly.h:86-87 and src/gpqhe.h:136-137,147) driven from C with real libgcrypt MPIs (tests/c/mpi_host.c),
checked against the Python-integer restatement of the reference.  `mpi_host polymul` is tests/polymul.c."""
import os
import random
import subprocess

import pytest

from oracle import bigint_ref as ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mpi_host(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mpi") / "mpi_host")
    lib_dir = os.path.join(ROOT, "gpqhe_amd")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "mpi_host.c"),
                           "-L", lib_dir, "-lgpqhe_hip", "-lgpqhe_hip_ctx", "-l:libgcrypt.so.20", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def _ints(lines):
    return [int(s, 16) for s in lines]


def test_polymul_c_kat_through_reference_signature(mpi_host, oracle_ctx):
    res = subprocess.run([mpi_host, "polymul"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    vals = _ints(res.stdout.split())
    assert len(vals) == 256
    o = oracle_ctx(7, 5)
    N, Q = 128, 1 << 61
    cases = [([i + 2 for i in range(N)], [i + 3 for i in range(N)]),
             ([o.p[0] - i - 1 for i in range(N)], [o.p[1] - i - 1 for i in range(N)])]
    leading = [[382784, 357372, 332350], [18559595904, 18272672062, 17985699960]]   # the reference's own printout
    for t, ((a, b), lead) in enumerate(zip(cases, leading)):
        got = vals[t * N:(t + 1) * N]
        assert got == [ref.centred_mod(v, Q) for v in ref.negacyclic_mul(a, b)]     # tests/polymul.gp
        assert [got[127], got[126], got[125]] == lead


def test_poly_mul_general_modulus_through_reference_signature(mpi_host, oracle_ctx):
    """poly_mul accepts any modulus (he_genswk passes P*q_L, src/he-kem.c:95): q = 3*2^61 + 7 here."""
    res = subprocess.run([mpi_host, "polymulodd"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    vals = _ints(res.stdout.split())
    o = oracle_ctx(7, 5)
    N, Q = 128, 3 * (1 << 61) + 7
    cases = [([i + 2 for i in range(N)], [i + 3 for i in range(N)]),
             ([o.p[0] - i - 1 for i in range(N)], [o.p[1] - i - 1 for i in range(N)])]
    for t, (a, b) in enumerate(cases):
        assert vals[t * N:(t + 1) * N] == [ref.mpi_smod(v, Q) for v in ref.negacyclic_mul(a, b)]


def test_poly_mul_at_a_size_with_threaded_conversions(mpi_host):
    """n = 2^13: the shim cuts the MPI <-> slab conversions into ranges for several host threads.  Dense signed 96-bit a times
    -3 x^5 (negacyclic): r_i = -3 a_{i-5}, with the sign flipped where the index wraps."""
    res = subprocess.run([mpi_host, "polymulmono", "13"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    vals = _ints(res.stdout.split())
    n, q = 1 << 13, 1 << 109
    a, r = vals[:n], vals[n:]
    assert len(r) == n and len(set(a)) > n - 4           # dense, (almost) all distinct
    exp = [ref.centred_mod(-3 * a[i - 5] if i >= 5 else 3 * a[n + i - 5], q) for i in range(n)]
    assert r == exp


def test_crt_bridge_through_reference_signatures(mpi_host, oracle_ctx):
    """rns_decompose per limb, rns_reconstruct per coefficient and poly_rns2mpi with the signatures of src/rns.c:37,60 and
    src/poly.h:88, driven like tests/crt.c:76-109 but on the production 60-bit chain and with negative coefficients."""
    res = subprocess.run([mpi_host, "crt"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    out = res.stdout.split()
    o = oracle_ctx(7, 5)
    N, dim = 128, 5
    P3 = o.p[0] * o.p[1] * o.p[2]
    P5 = P3 * o.p[3] * o.p[4]
    a = [(P3 - i - 1) * (-1 if i & 1 else 1) for i in range(N)]
    assert [int(v) for v in out[:dim * N]] == [a[i] % o.p[d] for d in range(dim) for i in range(N)]     # floor-mod, src/rns.c:43
    pos = dim * N
    assert _ints(out[pos:pos + N]) == [v % P5 for v in a]                                               # [0, P), src/rns.c:64-72
    pos += N
    for q in (1 << 61, 1000003 ** 3):
        assert _ints(out[pos:pos + N]) == [ref.mpi_smod(ref.mpi_smod(v % P5, P5), q) for v in a]        # src/poly.c:115-117
        pos += N
    assert pos == len(out)


@pytest.mark.parametrize("logn,qL,Delta", [
    (7, 1 << 120, 1 << 30),                       # the reference's test family: powers of two (tests/gpqhe.c:1349-1352)
    (7, 1000003 ** 5 * 1048573, 1000003),         # Delta and every q_l odd: the general-modulus kernels
    (12, 1 << 109, 1 << 30),                      # tests/gpqhe.c's default parameters; n >= 4096: threaded conversions, every call of the chain
                                                  # works on the device copies of what the call before wrote (resident polynomials)
    (12, 1000003 ** 4 * 1048573, 1000003),        # the same with odd moduli (100 bits: the security table allows 109 at n = 2^12, src/precomp.c:53-117)
])
def test_he_chain_with_real_mpis(mpi_host, oracle_ctx, tmp_path, logn, qL, Delta):
    """he_mul -> he_rescale -> he_moddown -> he_mul(&ct,&ct,&ct) -> he_rot -> he_conj -> he_mulpt on real libgcrypt MPIs."""
    n = 1 << logn
    logq, logDelta = qL.bit_length() - 1, Delta.bit_length() - 1
    L = logq // logDelta                                      # src/precomp.c:391
    q = [0] * (L + 1)
    cur = qL
    for l in range(L, -1, -1):
        q[l] = cur
        cur //= Delta                                         # :394-400
    level = L
    rng = random.Random(42)
    h = qL // 2
    polys = [[ref.mpi_smod(rng.randrange(qL), qL) for _ in range(n)] for _ in range(4)]
    for p in polys:
        p[:4] = [0, -1, ref.mpi_smod(h - 1, qL), ref.mpi_smod(h, qL)]
    path = tmp_path / "in.txt"
    with open(path, "w") as f:
        f.write("%d %X %d %d\n" % (logn, qL, Delta, level))
        for p in polys:
            for v in p:
                f.write(("-%X\n" % -v) if v < 0 else ("%X\n" % v))
    res = subprocess.run([mpi_host, "hemul", str(path)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.split("\n")
    dimub = (1 + logn + 4 * logq) // 59 + 1
    o = oracle_ctx(logn, dimub)

    def dims(ql):                                             # src/precomp.c:401,407; src/he-mult.c:99,51
        nbL, nbl = qL.bit_length(), ql.bit_length()
        dimP = (nbL + logn) // 59 + 1
        P = ref.RnsBasis(o.p[:dimP]).P
        nbPqL = (P * qL).bit_length()
        return dimP, (2 * nbl + logn) // 59 + 1, (nbl + nbPqL + logn) // 59 + 1, (nbL + nbPqL + logn) // 59 + 1

    dimP, dimA, dimB, dimevk = dims(q[level])
    assert lines[0].split() == ["dims", str(dimP), str(dimevk), str(L)]
    rlk0, rlk1 = o.gen(3000, dimevk), o.gen(3001, dimevk)

    # he_mul at level L (src/he-mult.c:88-156); nu and B as :93-95
    hdr = lines[1].split()
    assert hdr[0] == "he_mul" and int(hdr[1]) == level
    assert float(hdr[2]) == 3.0 * 7.0 and float(hdr[3]) == 3.0 * 11.0 + 7.0 * 5.0 + 5.0 * 11.0 + (100.0 + level)
    c0, c1 = _ints(lines[2:2 + n]), _ints(lines[2 + n:2 + 2 * n])
    e0, e1 = ref.he_mul(o, (polys[0], polys[1]), (polys[2], polys[3]), rlk0[: dimB * n], rlk1[: dimB * n], dimP, dimA, dimB, 0, ql=q[level])
    assert c0 == e0 and c1 == e1

    # he_rescale = he_rs (src/he-rescale.c:33-54)
    base = 2 + 2 * n
    hdr = lines[base].split()
    assert hdr[0] == "he_rs" and int(hdr[1]) == level - 1
    assert float(hdr[2]) == 21.0 / float(Delta)
    r0 = [ref.mpi_smod(ref.mpi_rdiv(v, Delta), q[level - 1]) for v in e0]
    r1 = [ref.mpi_smod(ref.mpi_rdiv(v, Delta), q[level - 1]) for v in e1]
    assert _ints(lines[base + 1:base + 1 + n]) == r0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == r1

    # he_moddown (src/he-rescale.c:56-70)
    base += 1 + 2 * n
    assert lines[base].split() == ["he_moddown", str(level - 2)]
    ql = q[level - 2]
    m0, m1 = [ref.mpi_smod(v, ql) for v in r0], [ref.mpi_smod(v, ql) for v in r1]
    assert _ints(lines[base + 1:base + 1 + n]) == m0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == m1

    # he_mul(&ct, &ct, &ct, rlk): output aliases both operands (src/he-algo.c:151), level L-2
    base += 1 + 2 * n
    assert lines[base].split() == ["he_sq", str(level - 2)]
    dP2, dA2, dB2, _ = dims(ql)
    s0, s1 = ref.he_mul(o, (m0, m1), (m0, m1), rlk0[: dB2 * n], rlk1[: dB2 * n], dP2, dA2, dB2, 0, ql=ql)
    assert _ints(lines[base + 1:base + 1 + n]) == s0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == s1

    # he_rot(ct, 1, rk): poly_rot both polynomials, he_swk with rk[1] (src/he-automorphism.c:100-115)
    base += 1 + 2 * n
    assert lines[base].split() == ["he_rot", str(level - 2)]
    rk0, rk1 = o.gen(5002, dimevk), o.gen(5003, dimevk)
    t0, t1 = ref.he_swk(o, ref.poly_rot(s0, 1), ref.poly_rot(s1, 1), rk0[: dB2 * n], rk1[: dB2 * n], dP2, dB2, 0, ql=ql)
    assert _ints(lines[base + 1:base + 1 + n]) == t0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == t1

    # he_conj(ct, ck) (src/he-automorphism.c:87-98)
    base += 1 + 2 * n
    assert lines[base].split() == ["he_conj", str(level - 2)]
    ck0, ck1 = o.gen(6000, dimevk), o.gen(6001, dimevk)
    u0, u1 = ref.he_swk(o, ref.poly_conj(t0), ref.poly_conj(t1), ck0[: dB2 * n], ck1[: dB2 * n], dP2, dB2, 0, ql=ql)
    assert _ints(lines[base + 1:base + 1 + n]) == u0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == u1

    # he_mulpt (src/he-mult.c:159-196): dim from log2(pt->nu) as :169; nu, B as :163-164
    base += 1 + 2 * n
    hdr = lines[base].split()
    assert hdr[0] == "he_mulpt" and int(hdr[1]) == level - 2 and float(hdr[2]) == 2.0 * 1024.0 and float(hdr[3]) == 3.0 * 1024.0
    m = [(((i + 1) << 20) + 12345 * i) * (-1 if i & 1 else 1) for i in range(n)]
    dim_pt = int((ql.bit_length() + 10.0 + logn) / 59 + 1)
    v0, v1 = ref.he_mulpt(o, (u0, u1), m, dim_pt, 0, ql=ql)
    assert _ints(lines[base + 1:base + 1 + n]) == v0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == v1
    # src/he-add.c on the same chain: ct still holds he_conj's result (u), prod the he_mulpt result (v)
    base += 1 + 2 * n
    hdr = lines[base].split()
    assert hdr[0] == "he_add" and int(hdr[1]) == level - 2 and float(hdr[2]) == 2048.0 and float(hdr[3]) == 3.0 * 1024.0 + 3.0
    a0, a1 = ref.he_add((v0, v1), (u0, u1), ql)
    assert _ints(lines[base + 1:base + 1 + n]) == a0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == a1
    base += 1 + 2 * n
    assert lines[base].split() == ["he_sub", str(level - 2)]
    b0, b1 = ref.he_sub((a0, a1), (v0, v1), ql)
    assert _ints(lines[base + 1:base + 1 + n]) == b0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == b1
    base += 1 + 2 * n
    hdr = lines[base].split()
    assert hdr[0] == "he_addpt" and int(hdr[1]) == level - 2 and float(hdr[2]) == 2048.0 and float(hdr[3]) == 3.0 * 1024.0
    p0, p1 = ref.he_addpt((v0, v1), m, ql)
    assert _ints(lines[base + 1:base + 1 + n]) == p0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == p1
    base += 1 + 2 * n
    assert lines[base].split() == ["he_subpt", str(level - 2)]
    s0_, s1_ = ref.he_subpt((p0, p1), m, ql)
    assert _ints(lines[base + 1:base + 1 + n]) == s0_ and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == s1_
    base += 1 + 2 * n
    assert lines[base].split() == ["he_neg", str(level - 2)]
    n0, n1 = ref.he_neg((s0_, s1_), ql)
    assert _ints(lines[base + 1:base + 1 + n]) == n0 and _ints(lines[base + 1 + n:base + 1 + 2 * n]) == n1
    base += 1 + 2 * n
    assert lines[base].split() == ["he_copy_ct", "identical"]                                # src/he-mem.c:88-97: l, nu, B and every integer
    # he_dec (src/he-encrypt.c:105-125) of the copy with the sparse key 1 - x^5 + x^(n-1), twice (the second time everything is resident)
    want = ref.he_dec_sparse((n0, n1), {0: 1, 5: -1, n - 1: 1}, ql)
    base += 1
    assert lines[base].split() == ["he_dec", "6.5"]
    assert _ints(lines[base + 1:base + 1 + n]) == want
    base += 1 + n
    assert lines[base].split() == ["he_dec", "again"]
    assert _ints(lines[base + 1:base + 1 + n]) == want
    base -= n                                                                                 # (n integers behind the last header: the stride below adds 1 + 2 n)
    # he_rescale, he_moddown, the squaring (its key is the resident rlk, a prefix of it at this level) and he_mulpt took their ciphertext
    # (2 polynomials each) from the device copies the call before left; he_rot / he_conj meet their keys for the first time and upload
    # everything -- at n >= 4096; smaller rings always convert and upload
    base += 1 + 2 * n
    tag, confirmed, changed = lines[base].split()
    # ... and so did the five additive calls where q_l is a power of two (4 + 4 + 3 + 3 + 2 polynomials; the plaintext is resident since he_mulpt)
    assert tag == "resident" and int(changed) == 0
    # + he_copy_ct's two polynomials (any modulus) + he_dec's three the second time (the first time the key is new and the call measures and uploads)
    assert int(confirmed) == (0 if logn < 12 else 8 + 16 + 2 + 3 if qL & (qL - 1) == 0 else 8 + 2)


def _splitmix(state):
    state[0] = (state[0] + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state[0]
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


@pytest.mark.parametrize("logn,logq", [(7, 120), (8, 200)])
def test_key_generation_through_reference_signatures(mpi_host, oracle_ctx, logn, logq):
    """he_genrlk / he_genck / he_genrk (src/gpqhe.h:131-133, src/he-kem.c:74-170): the library calls the host program's
    sample_error / sample_uniform in the reference's order and does the rest on the device.  The C host's samplers are
    deterministic, so the keys are restated here with Python integers and the oracle's NTT and compared by digest."""
    import numpy as np
    from oracle.oracle import fnv
    res = subprocess.run([mpi_host, "keygen", str(logn), str(logq)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = [ln.split() for ln in res.stdout.splitlines()]
    dims = next(ln for ln in lines if ln and ln[0] == "dims")
    got = {int(ln[1]): (int(ln[2], 16), int(ln[3], 16)) for ln in lines if ln and ln[0] == "key"}
    n, q = 1 << logn, 1 << logq
    dimP = (logq + 1 + logn) // 59 + 1
    o = oracle_ctx(logn, 64 if logn > 7 else 5 + 40)
    P = ref.RnsBasis(o.p[:dimP]).P
    PqL = P * q
    dimevk = (logq + 1 + PqL.bit_length() + logn) // 59 + 1
    assert (int(dims[1]), int(dims[2])) == (dimP, dimevk)
    st = [333]
    s = [{0: 0, 1: 1, 2: -1}[_splitmix(st) % 3] for _ in range(n)]
    es, us = [111], [222]
    nb = (PqL.bit_length() + 7) // 8 + 8

    def sample_error():
        return [_splitmix(es) % 17 - 8 for _ in range(n)]

    def sample_uniform():
        out = []
        for _ in range(n):
            buf = b"".join(_splitmix(us).to_bytes(8, "little") for _ in range((nb + 7) // 8))
            out.append(int.from_bytes(buf[:nb], "big") % PqL)
        return out

    def slab(poly):
        return o.ntt_slab(np.array([v % o.p[d] for d in range(dimevk) for v in poly], dtype=np.uint64), dimevk)

    def genswk(sp):
        e = sample_error()
        p1 = sample_uniform()
        p0 = [ref.mpi_smod(-a + b + P * c, PqL) for a, b, c in zip(ref.negacyclic_mul(p1, s), e, sp)]
        return int(fnv(slab(p0)), 16), int(fnv(slab([ref.mpi_smod(v, PqL) for v in p1])), 16)

    s2 = [ref.centred_mod(v, q) for v in ref.negacyclic_mul(s, s)]
    assert got[0] == genswk(s2)                               # he_genrlk
    assert got[1] == genswk(ref.poly_conj(s))                 # he_genck
    assert got[2] == genswk(ref.poly_rot(s, 0))               # he_genrk, rot = 0, 1
    assert got[3] == genswk(ref.poly_rot(s, 1))


@pytest.mark.parametrize("logn,logq,Delta", [(7, 61, 1 << 30), (14, 438, 1 << 50), (16, 850, 1 << 50)])
def test_context_symbols_of_the_library(mpi_host, oracle_ctx, golden, logn, logq, Delta):
    """polyctx_init / hectx_init / poly_mpi_alloc / poly_rns_alloc and the data symbols polyctx, hectx, GPQHE_TWO as the library
    defines them (weak; src/poly.h:80-83,94-95, src/gpqhe.h:100-101): every field against the restated formulas of
    src/precomp.c:266-293 and :328-450, the per-prime constants against the oracle, and L / dim / dimevk / dimub / nbits(P) /
    nbits(P q_L) against the values SURVEY.md 8c captured from the reference's own hectx_init."""
    import math
    res = subprocess.run([mpi_host, "ctxcheck", str(logn), str(logq), str(Delta)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.split("\n")
    it = iter(lines)
    n = 1 << logn
    logqub = {10: 27, 11: 54, 12: 109, 13: 218, 14: 438, 15: 881}.get(logn, logq)       # src/precomp.c:53-64, :338-340
    dimub = (1 + logn + 4 * logqub) // 59 + 1                                             # :357
    assert next(it).split() == ["poly", str(logn), str(n), str(2 * n), str(logq), str(logqub), str(dimub)]
    assert next(it).split() == ["two", "02"]
    assert int(next(it).split()[1], 16) == 1 << logq
    o = oracle_ctx(logn, dimub)
    for d in range(dimub):
        f = next(it).split()
        z, zi = o.zetas(d), o.zetas(d, inverse=True)
        assert f == ["node", str(d + 1), str(o.p[d]), str(o.const("pinv_mont", d)), str(o.const("pinv_barr", d)), str(o.const("ninv", d)),
                     str(z[1]), str(zi[n // 2])]
        if d + 1 <= 5 or d + 1 == dimub:
            basis = ref.RnsBasis(o.p[: d + 1])
            assert int(next(it).split()[1], 16) == basis.P
            assert int(next(it).split()[1], 16) == basis.P // 2
            for k in range(d + 1):
                g = next(it).split()
                assert g[:3] == ["phat", str(k), str(basis.phat_invmp[k])] and int(g[3], 16) == basis.phat[k]
            if logn == 7:
                assert [str(v) for v in basis.phat_invmp] == golden["phat_invmp_logn7"][d]       # the reference's own printout
    assert next(it).split() == ["count", str(dimub)]
    f = next(it).split()
    assert f[:3] == ["ring", "5", str(pow(5, n // 2 - 1, 2 * n))]                                # src/precomp.c:300-303
    assert float(f[3]) == math.cos(2 * 3.141592653589793238462643383279502884 / (2 * n)) and abs(float(f[4])) == 0.0
    # qtable_init, src/precomp.c:386-409
    logDelta = Delta.bit_length() - 1
    L = logq // logDelta
    qL = 1 << logq
    dimP = (qL.bit_length() + logn) // 59 + 1
    P = ref.RnsBasis(o.p[:dimP]).P
    dimevk = (qL.bit_length() + (P * qL).bit_length() + logn) // 59 + 1
    f = next(it).split()
    assert f[:7] == ["he", str(L), str(dimP), str(dimevk), str(P.bit_length()), str((P * qL).bit_length()), "2"] and float(f[7]) == float(Delta)
    key = "%d_%d_%d" % (logn, logq, logDelta)
    if key in golden["context_dims"]:                                                           # the reference's own numbers
        c = golden["context_dims"][key]
        assert (L, dimP, dimevk, dimub) == (c["L"], c["dim"], c["dimevk"], c["dimub"])
        if "nbits_P" in c:
            assert (P.bit_length(), (P * qL).bit_length()) == (c["nbits_P"], c["nbits_PqL"])
    cur = qL
    qs = {}
    for l in range(L, -1, -1):
        qs[l] = cur
        cur //= Delta
    for l in range(L + 1):
        a, b = next(it).split(), next(it).split()
        assert a[:2] == ["q", str(l)] and int(a[2], 16) == qs[l] and b[:2] == ["qh", str(l)] and int(b[2], 16) == qs[l] // 2
    # bounds_init, src/precomp.c:411-432 (doubles: same formulas, compared to a few ulps)
    sigma, h = 3.1915382432114616, 64
    Bclean = 8 * math.sqrt(2) * sigma * n + 6 * sigma * math.sqrt(n) + 16 * sigma * math.sqrt(h * n)
    Brs, Bks = math.sqrt(n / 3.0) * (3 + 8 * math.sqrt(h)), 8 * sigma * n / math.sqrt(3)
    f = [float(v) for v in next(it).split()[1:]]
    assert f[0] == pytest.approx(Bclean, rel=1e-14) and f[1] == pytest.approx(Brs, rel=1e-14) and f[2] == pytest.approx(Bks, rel=1e-14)
    pinv = 1.0
    for p in o.p:
        pinv /= p
    assert f[3] == pytest.approx(pinv * qs[0] * Bks + Brs, rel=1e-9) and f[4] == pytest.approx(pinv * qs[0] * float(Delta) ** L * Bks + Brs, rel=1e-9)
    assert next(it).split() == ["alloc", "0", str(dimevk * n - 1)]
    assert next(it).split() == ["exit", "1", "1"]
    assert next(it).split() == ["again", str(dimub), str(dimevk)]


@pytest.mark.timeout(600)
def test_reference_signature_he_mul_at_the_headline_shape_keeps_its_key_on_the_device(mpi_host):
    """`mpi_host hemultime 16 850`: he_mul(he_ct_t*, ...) / he_rescale with real MPIs at n = 2^16, q = 2^850 (hectx_init of the library:
    L = 17, dims 15 / 45).  The evaluation key is uploaded once and recognised afterwards; a key rewritten in place is seen (fingerprint)
    and gives the same result as after gpq_mpi_shim_forget_keys(); the per-stage timing of the last call adds up."""
    import re
    res = subprocess.run([mpi_host, "hemultime", "16", "850"], capture_output=True, text=True, timeout=560)
    assert res.returncode == 0, res.stderr
    assert "key cache: rewritten key seen, cached vs fresh upload identical" in res.stdout, res.stdout
    assert "direct mpi access: in use, print/scan path identical" in res.stdout, res.stdout
    # safe by default: ONE word edited in place, at an index a sampled fingerprint never looks at, multiplies as edited
    assert "key cache: one unsampled word edited in place seen, cached vs fresh upload identical" in res.stdout, res.stdout
    assert "key cache: resident 1, after set_key_slots(1) 1" in res.stdout or re.search(r"after set_key_slots\(1\) 1\b", res.stdout), res.stdout
    m = re.search(r"dims 15/45: ([0-9.]+) ms per call; he_rescale ([0-9.]+) ms", res.stdout)
    assert m and 0.3 < float(m.group(1)) < 200 and 0.1 < float(m.group(2)) < 200, res.stdout
    t = re.search(r"convert\+upload ([0-9.]+) ms, kernels ([0-9.]+) ms, download\+convert ([0-9.]+) ms, call ([0-9.]+) ms", res.stdout)
    parts = [float(v) for v in t.groups()]
    assert 0.1 < parts[1] < 5 and parts[0] + parts[2] <= parts[3] * 1.05 and parts[3] < 200
    # resident polynomials (round 3): one coefficient changed by one, a sign flipped, an integer object replaced, a coefficient overwritten,
    # a result touched between two chained calls -- every product equals the one of a library that uploads everything afresh
    assert "resident polynomials: edits and chains identical to fresh uploads" in res.stdout, res.stdout
    r = re.search(r"(\d+) operands confirmed, (\d+) found changed", res.stdout)
    assert r and int(r.group(1)) > 0 and int(r.group(2)) == 6, res.stdout
    ch = re.search(r"chained .*he_mul p50 ([0-9.]+) .*he_mul\(&ct, &ct, &ct\) p50 ([0-9.]+) .*he_rescale of a product p50 ([0-9.]+)", res.stdout)
    assert ch and all(0.05 < float(v) < 200 for v in ch.groups()), res.stdout
    # the square + rescale ladder down to level 0, where q_0 = 2^(850 - 17 * 50) = 1 and the reference's mpi_smod leaves -1 everywhere
    assert len(re.findall(r"ladder of 17 x .* ([0-9.]+) ms", res.stdout)) == 2, res.stdout
    assert res.stdout.count("level 0 has q_0 = 1: every coefficient is -1") == 2, res.stdout
    # the additive calls, he_copy_ct and he_inv's whole call sequence (src/he-algo.c:130-165) on resident ciphertexts
    assert re.search(r"additive calls in a chain .*he_add p50 ([0-9.]+)", res.stdout) and re.search(r"he_copy_ct of a chained ciphertext p50", res.stdout), res.stdout
    assert len(re.findall(r"he_inv's call sequence, 8 iterations \(45 calls, level 17 -> 8\)", res.stdout)) == 2, res.stdout


@pytest.mark.timeout(900)
@pytest.mark.parametrize("logn,logq,logDelta,steps,seed", [(12, 109, 20, 400, 1), (12, 109, 30, 300, 2), (13, 200, 25, 250, 3)])
def test_random_walk_with_resident_polynomials_equals_fresh_uploads(mpi_host, logn, logq, logDelta, steps, seed):
    """`mpi_host residentfuzz`: a random walk over he_mul (every aliasing pattern) / he_rescale / he_moddown / he_rot / he_conj / he_mulpt,
    host-side edits of ciphertexts between calls, copies, evictions and keys rewritten in place.  Every step runs on a ciphertext the
    library may hold device copies of and on a twin that is always converted and uploaded afresh: the integers must never differ."""
    import re
    res = subprocess.run([mpi_host, "residentfuzz", str(logn), str(logq), str(logDelta), str(steps), str(seed)], capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    m = re.search(r"residentfuzz ok: (\d+) steps .* (\d+) operands confirmed, (\d+) found changed", res.stdout)
    assert m and int(m.group(2)) > 20 and int(m.group(3)) > 0, res.stdout[-2000:]

