"""Semantic check of the MPI-level he_mul, independent of the restated he_relin: a miniature CKKS round trip
in the spirit of tests/gpqhe.c:422-537 (encrypt -> he_mul -> [he_rs] -> decrypt, error small against the scale).

Keys follow the reference's construction: secret s with small coefficients; relinearisation key as he_genswk
builds it (src/he-kem.c:74-118): p1 uniform mod P*q_L, p0 = -p1*s + e + P*s^2 mod P*q_L, both stored as NTT-domain
slabs over dimevk limbs (:103-110).  Messages are integer polynomials at scale Delta (the encoder is out of scope).
Everything around the device call is Python integers."""
import random

import numpy as np
import pytest

from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host
from oracle import bigint_ref as ref

pytestmark = pytest.mark.gpu


def _negacyclic_mod(a, b, q):
    return [ref.centred_mod(v, q) for v in ref.negacyclic_mul(a, b)]


def _evk_slab(o, poly, dimevk):
    """rns_decompose + ntt per limb, src/he-kem.c:103-110"""
    slab = np.array([v % o.p[d] for d in range(dimevk) for v in poly], dtype=np.uint64)
    return o.ntt_slab(slab, dimevk)


@pytest.mark.parametrize("logn,logq,logDelta", [(7, 120, 30), (8, 150, 40)])
def test_encrypt_mul_rescale_decrypt(engine_ctx, oracle_ctx, logn, logq, logDelta):
    import torch
    n, q, Delta = 1 << logn, 1 << logq, 1 << logDelta
    probe = engine_ctx(logn, 12)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = random.Random(2024 + logn)
    P = ref.RnsBasis(o.p[:dimP]).P
    PqL = P * q

    small = lambda: [rng.choice((-1, 0, 0, 1)) for _ in range(n)]            # secret / error shape (sampler out of scope)
    err = lambda: [rng.randrange(-8, 9) for _ in range(n)]
    s = small()
    s2 = ref.negacyclic_mul(s, s)
    # relinearisation key, src/he-kem.c:80-101 with sp = s^2
    p1 = [rng.randrange(PqL) for _ in range(n)]
    e = err()
    p0 = [ref.mpi_smod(-a + b + P * c, PqL) for a, b, c in zip(ref.negacyclic_mul(p1, s), e, s2)]
    p1c = [ref.mpi_smod(v, PqL) for v in p1]
    rlk0, rlk1 = _evk_slab(o, p0, dimevk), _evk_slab(o, p1c, dimevk)

    def encrypt(m):                                                          # he_enc_sk shape, src/he-encrypt.c:80-99
        a = [rng.randrange(q) for _ in range(n)]
        c0 = [ref.centred_mod(-x + mm + ee, q) for x, mm, ee in zip(ref.negacyclic_mul(a, s), m, err())]
        return c0, [ref.centred_mod(v, q) for v in a]

    def decrypt(c0, c1, ql):                                                 # he_dec, src/he-encrypt.c:105-123
        return [ref.centred_mod(x + y, ql) for x, y in zip(c0, ref.negacyclic_mul(c1, s))]

    m1 = [rng.randrange(-50, 51) * Delta for _ in range(n)]
    m2 = [rng.randrange(-50, 51) * Delta for _ in range(n)]
    ct1, ct2 = encrypt(m1), encrypt(m2)
    assert max(abs(x - y) for x, y in zip(decrypt(*ct1, q), m1)) < 2**10    # sanity of the miniature scheme itself

    W = logq // 64 + 1
    dev = [to_device(ints_to_big(v, W)) for v in (ct1[0], ct1[1], ct2[0], ct2[1])]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(o0, o1, *dev, to_device(rlk0), to_device(rlk1), W, logq, dimA, dimB, dimP)   # src/he-mult.c:88-156
    c0, c1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    want = _negacyclic_mod(m1, m2, q)                                        # scale Delta^2
    got = decrypt(c0, c1, q)
    noise = max(abs(x - y) for x, y in zip(got, want))
    assert noise < Delta * 2**22, "he_mul does not decrypt to m1*m2: relative error 2^%.1f" % (noise.bit_length() - 2 * logDelta)
    assert noise < (max(abs(v) for v in want) >> 12)                         # < 2^-12 relative, tests/gpqhe.c uses 1e-5 on decoded values

    g.he_rs(o0, o1, W, logDelta, logq - logDelta)                            # src/he-rescale.c:33-54
    r0, r1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    got_rs = decrypt(r0, r1, q >> logDelta)
    want_rs = [ref.centred_mod(ref.mpi_rdiv(v, Delta), q >> logDelta) for v in want]
    assert max(abs(x - y) for x, y in zip(got_rs, want_rs)) < 2**24          # back at scale Delta, small absolute noise


def test_add_sub_neg_and_evk_pack(engine_ctx, oracle_ctx):
    """src/he-add.c semantics (mpi_addm/mpi_subm + mpi_smod) on big slabs, and the key-slab storage of
    src/he-kem.c:103-110 (decompose + ntt over dimevk limbs)."""
    import ctypes as C
    import torch
    logn, logql, W, dimevk = 7, 100, 2, 6
    g, o = engine_ctx(logn, 12), oracle_ctx(logn, 12)
    n, ql = g.n, 1 << logql
    rng = random.Random(4)
    h = ql // 2
    a = [rng.randrange(-h, h) for _ in range(n)]
    b = [rng.randrange(-h, h) for _ in range(n)]
    a[:4], b[:4] = [h - 1, -h, -h, 0], [h - 1, -h, h - 1, 0]
    da, db = to_device(ints_to_big(a, W)), to_device(ints_to_big(b, W))
    r = torch.empty_like(da)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert g.lib.gpq_big_add(g.h, P(r), P(da), P(db), W, logql, 1, st) == 0
    assert big_to_ints(to_host(r), W, n)[0] == [ref.mpi_smod((x + y) % ql, ql) for x, y in zip(a, b)]      # he_add, src/he-add.c:40-45
    assert g.lib.gpq_big_sub(g.h, P(r), P(da), P(db), W, logql, 1, st) == 0
    assert big_to_ints(to_host(r), W, n)[0] == [ref.mpi_smod((x - y) % ql, ql) for x, y in zip(a, b)]      # he_sub
    assert g.lib.gpq_big_neg(g.h, P(r), P(da), W, logql, 1, st) == 0
    assert big_to_ints(to_host(r), W, n)[0] == [ref.mpi_smod(-x, ql) for x in a]                           # he_neg
    evk = torch.empty(dimevk * n, dtype=torch.int64, device="cuda")
    assert g.lib.gpq_evk_pack(g.h, P(evk), P(da), W, dimevk, 1, st) == 0
    assert np.array_equal(to_host(evk), _evk_slab(o, a, dimevk))


def _sparse_negacyclic(dense, terms, n):
    """dense * (sum of c x^k) mod x^n + 1, exactly, in O(len(terms) n)"""
    out = [0] * n
    for k, c in terms:
        for i, v in enumerate(dense):
            j = i + k
            if j < n:
                out[j] += c * v
            else:
                out[j - n] -= c * v
    return out


def _dense_of(terms, n):
    out = [0] * n
    for k, c in terms:
        out[k] += c
    return out


@pytest.mark.parametrize("logn,logq,logDelta", [(13, 438, 40), (14, 438, 50), (16, 850, 50)])
def test_encrypt_mul_rescale_decrypt_at_two_pass_sizes(engine_ctx, oracle_ctx, logn, logq, logDelta):
    """The same round trip at ring sizes that take the two-pass NTT kernels and the matrix-core bridge (the reference's own test
    default is logn = 14, q = 2^438, tests/gpqhe.c:1349-1352).  Every host-side polynomial product has one sparse factor (secret,
    key mask, encryption mask and the second message have a few terms), so the expected values stay exact Python integers."""
    import torch
    n, q, Delta = 1 << logn, 1 << logq, 1 << logDelta
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = random.Random(77 + logn)
    P = ref.RnsBasis(o.p[:dimP]).P
    PqL = P * q
    sparse = lambda cnt, draw: sorted({rng.randrange(n): draw() for _ in range(cnt)}.items())
    err = lambda: [rng.randrange(-8, 9) for _ in range(n)]
    s_terms = sparse(24, lambda: rng.choice((-1, 1)))
    s = _dense_of(s_terms, n)
    s2 = _sparse_negacyclic(s, s_terms, n)
    p1_terms = sparse(6, lambda: rng.randrange(PqL))
    p1 = _dense_of(p1_terms, n)
    p1s = _sparse_negacyclic(s, p1_terms, n)                                 # p1 * s
    e = err()
    p0 = [ref.mpi_smod(-a + b + P * c, PqL) for a, b, c in zip(p1s, e, s2)]
    rlk0, rlk1 = _evk_slab(o, p0, dimevk), _evk_slab(o, [ref.mpi_smod(v, PqL) for v in p1], dimevk)

    def encrypt(m):
        a_terms = sparse(6, lambda: rng.randrange(q))
        a_s = _sparse_negacyclic(s, a_terms, n)
        return [ref.centred_mod(-x + mm + ee, q) for x, mm, ee in zip(a_s, m, err())], [ref.centred_mod(v, q) for v in _dense_of(a_terms, n)]

    def decrypt(c0, c1, ql):
        return [ref.centred_mod(x + y, ql) for x, y in zip(c0, _sparse_negacyclic(c1, s_terms, n))]

    m1 = [rng.randrange(-50, 51) * Delta for _ in range(n)]
    m2_terms = sparse(5, lambda: rng.randrange(-50, 51) * Delta)
    m2 = _dense_of(m2_terms, n)
    ct1, ct2 = encrypt(m1), encrypt(m2)
    W = logq // 64 + 1
    dev = [to_device(ints_to_big(v, W)) for v in (ct1[0], ct1[1], ct2[0], ct2[1])]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(o0, o1, *dev, to_device(rlk0), to_device(rlk1), W, logq, dimA, dimB, dimP)
    c0, c1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    want = [ref.centred_mod(v, q) for v in _sparse_negacyclic(m1, m2_terms, n)]
    got = decrypt(c0, c1, q)
    noise = max(abs(x - y) for x, y in zip(got, want))
    assert noise < (max(abs(v) for v in want) >> 12), "relative error 2^%d" % (noise.bit_length() - max(abs(v) for v in want).bit_length())
    g.he_rs(o0, o1, W, logDelta, logq - logDelta)
    r0, r1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    got_rs = decrypt(r0, r1, q >> logDelta)
    want_rs = [ref.centred_mod(ref.mpi_rdiv(v, Delta), q >> logDelta) for v in want]
    assert max(abs(x - y) for x, y in zip(got_rs, want_rs)) < 2**30


@pytest.mark.parametrize("logn,logq,which", [(13, 438, "rot3"), (16, 850, "rot1"), (16, 850, "conj")])
def test_rotate_and_conjugate_decrypt_to_the_permuted_message(engine_ctx, oracle_ctx, logn, logq, which):
    """he_rot / he_conj (src/he-automorphism.c:87-115): permute both polynomials (src/poly.c:263-283), key-switch from the permuted
    secret back to s with a key built as he_genswk does (src/he-kem.c:74-118).  The result must decrypt under s to the permuted
    message.  Sparse secret and masks keep every host-side product exact and cheap, at the two-pass sizes up to the headline's."""
    import torch
    n, q = 1 << logn, 1 << logq
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = random.Random(99 + logn)
    P = ref.RnsBasis(o.p[:dimP]).P
    PqL = P * q
    perm = (lambda a: ref.poly_conj(a)) if which == "conj" else (lambda a: ref.poly_rot(a, int(which[3:])))
    sparse = lambda cnt, draw: sorted({rng.randrange(n): draw() for _ in range(cnt)}.items())
    terms_of = lambda dense: [(i, v) for i, v in enumerate(dense) if v]
    err = lambda: [rng.randrange(-8, 9) for _ in range(n)]
    s_terms = sparse(24, lambda: rng.choice((-1, 1)))
    s = _dense_of(s_terms, n)
    sp = perm(s)                                                             # the secret the permuted ciphertext is under
    p1_terms = sparse(6, lambda: rng.randrange(PqL))
    p1 = _dense_of(p1_terms, n)
    p0 = [ref.mpi_smod(-a + b + P * c, PqL) for a, b, c in zip(_sparse_negacyclic(s, p1_terms, n), err(), sp)]
    swk0, swk1 = _evk_slab(o, p0, dimevk), _evk_slab(o, [ref.mpi_smod(v, PqL) for v in p1], dimevk)
    a_terms = sparse(6, lambda: rng.randrange(q))
    m = [rng.randrange(-1000, 1001) << 40 for _ in range(n)]
    c0 = [ref.centred_mod(-x + mm + ee, q) for x, mm, ee in zip(_sparse_negacyclic(s, a_terms, n), m, err())]
    c1 = [ref.centred_mod(v, q) for v in _dense_of(a_terms, n)]
    W = logq // 64 + 1
    d0, d1 = to_device(ints_to_big(c0, W)), to_device(ints_to_big(c1, W))
    r0, r1 = torch.empty_like(d0), torch.empty_like(d0)
    if which == "conj":
        g.poly_conj(r0, d0, W); g.poly_conj(r1, d1, W)
    else:
        g.poly_rot(r0, d0, W, int(which[3:])); g.poly_rot(r1, d1, W, int(which[3:]))
    assert big_to_ints(to_host(r1), W, n)[0] == perm(c1)                    # the permutation itself, exactly
    o0, o1 = torch.empty_like(d0), torch.empty_like(d0)
    g.he_swk(o0, o1, r0, r1, to_device(swk0), to_device(swk1), W, logq, dimB, dimP)
    k0, k1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    got = [ref.centred_mod(x + y, q) for x, y in zip(k0, _sparse_negacyclic(k1, s_terms, n))]
    want = perm(m)
    assert max(abs(x - y) for x, y in zip(got, want)) < 1 << 30              # key-switching noise, far below the 2^40 scale
    assert terms_of(sp) != s_terms


@pytest.mark.parametrize("logn,logq", [(13, 438), (16, 850)])
def test_plaintext_multiplication_decrypts_to_the_product(engine_ctx, oracle_ctx, logn, logq):
    """he_mulpt (src/he-mult.c:159-196): both ciphertext polynomials times the plaintext polynomial mod q_l.  The plaintext is
    dense in the device call and has a few terms, so m * pt is exact on the host."""
    import torch
    n, q, lognu = 1 << logn, 1 << logq, 45
    dim = (logq + 1 + lognu + logn) // 59 + 1                                # :169 with nu = 2^lognu
    g = engine_ctx(logn, max(dim, 20))
    rng = random.Random(5 + logn)
    sparse = lambda cnt, draw: sorted({rng.randrange(n): draw() for _ in range(cnt)}.items())
    s_terms = sparse(24, lambda: rng.choice((-1, 1)))
    s = _dense_of(s_terms, n)
    a_terms = sparse(6, lambda: rng.randrange(q))
    m = [rng.randrange(-1000, 1001) << 40 for _ in range(n)]
    c0 = [ref.centred_mod(-x + mm + rng.randrange(-8, 9), q) for x, mm in zip(_sparse_negacyclic(s, a_terms, n), m)]
    c1 = [ref.centred_mod(v, q) for v in _dense_of(a_terms, n)]
    pt_terms = sparse(7, lambda: rng.randrange(-(1 << lognu), 1 << lognu))
    W = logq // 64 + 1
    d0, d1, dp = (to_device(ints_to_big(v, W)) for v in (c0, c1, _dense_of(pt_terms, n)))
    o0, o1 = torch.empty_like(d0), torch.empty_like(d0)
    g.he_mulpt(o0, o1, d0, d1, dp, W, logq, dim)
    k0, k1 = big_to_ints(to_host(o0), W, n)[0], big_to_ints(to_host(o1), W, n)[0]
    assert k0 == [ref.centred_mod(v, q) for v in _sparse_negacyclic(c0, pt_terms, n)]       # exact, polynomial by polynomial
    assert k1 == [ref.centred_mod(v, q) for v in _sparse_negacyclic(c1, pt_terms, n)]
    got = [ref.centred_mod(x + y, q) for x, y in zip(k0, _sparse_negacyclic(k1, s_terms, n))]
    want = [ref.centred_mod(v, q) for v in _sparse_negacyclic(m, pt_terms, n)]
    assert max(abs(x - y) for x, y in zip(got, want)) < 1 << (lognu + 12)                    # noise: 7 terms * |e| * |pt|


@pytest.mark.parametrize("logn,logq,sparse_s", [(7, 120, False), (8, 200, False), (13, 438, True), (16, 850, True)])
def test_he_genswk_on_device_matches_the_reference_construction(engine_ctx, oracle_ctx, logn, logq, sparse_s):
    """gpq_he_genswk (src/he-kem.c:74-118 with the sampling left to the caller): swk.p0 = smod(-p1 sk + e + P sp, P q_L),
    swk.p1 = smod(p1, P q_L), stored as rns_decompose + ntt over dimevk limbs -- against Python integers and the oracle's NTT."""
    import torch
    n, q = 1 << logn, 1 << logq
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = random.Random(31 + logn)
    P = ref.RnsBasis(o.p[:dimP]).P
    PqL = P * q
    if sparse_s:
        s_terms = sorted({rng.randrange(n): rng.choice((-1, 1)) for _ in range(24)}.items())
        s = _dense_of(s_terms, n)
        mul_s = lambda a: _sparse_negacyclic(a, s_terms, n)
    else:
        s = [rng.choice((-1, 0, 1)) for _ in range(n)]
        mul_s = lambda a: ref.negacyclic_mul(a, s)
    sp = mul_s(s)                                                            # he_genrlk: the key hides s^2
    p1 = [rng.randrange(PqL) for _ in range(n)]                             # sample_uniform(&swkp1, PqL): [0, P q_L)
    e = [rng.randrange(-8, 9) for _ in range(n)]
    p0 = [ref.mpi_smod(-a + b + P * c, PqL) for a, b, c in zip(mul_s(p1), e, sp)]
    want0, want1 = _evk_slab(o, p0, dimevk), _evk_slab(o, [ref.mpi_smod(v, PqL) for v in p1], dimevk)
    W = PqL.bit_length() // 64 + 1
    dev = [to_device(ints_to_big(v, W)) for v in (p1, s, e, sp)]
    evk0 = torch.empty(dimevk * n, dtype=torch.int64, device="cuda")
    evk1 = torch.empty_like(evk0)
    g.he_genswk(evk0, evk1, *dev, W, dimP, logq, dimevk)
    assert np.array_equal(to_host(evk0), want0)
    assert np.array_equal(to_host(evk1), want1)
