"""he_mul / he_swk at the MPI level (big slabs) against the restatement of
src/he-mult.c:40-156 / src/he-automorphism.c:40-85 built from the C oracle's RNS core and
Python integers for the libgcrypt parts (oracle/bigint_ref.py)."""
import random

import numpy as np
import pytest

from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host
from oracle import bigint_ref as ref

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def test_he_dims_match_reference_contexts(golden, engine_ctx):
    """hectx_init(14, 2^438, ., 2^50) and hectx_init(16, 2^850, ., 2^50), SURVEY.md 8c."""
    for key, logn, logq in (("14_438_50", 14, 438), ("16_850_50", 16, 850)):
        rec = golden["context_dims"][key]
        g = engine_ctx(logn, rec["dimevk"])
        dimP, dimA, dimB, dimevk = g.he_dims(logq, logq)
        assert (dimP, dimA, dimB, dimevk) == (rec["dim"], rec["dimA"], rec["dimB"], rec["dimevk"])
        assert ref.he_dims(logn, g.p, logq, logq) == (dimP, dimA, dimB, dimevk)
        if "nbits_P" in rec:
            assert g.lib.gpq_ctx_pbits(g.h, dimP) == rec["nbits_P"]


def _centred(rng, logq, n):
    h = 1 << (logq - 1)
    vals = [rng.randrange(-h, h) for _ in range(n)]
    vals[:6] = [0, 1, -1, h - 1, -h, h // 2]
    return vals


@pytest.mark.parametrize("logn,logqL,logql,batch", [(7, 120, 120, 2), (7, 120, 90, 1), (8, 200, 150, 1), (7, 61, 61, 3)])
def test_he_mul_matches_reference_semantics(engine_ctx, oracle_ctx, logn, logqL, logql, batch):
    torch = _torch()
    probe = engine_ctx(logn, 12)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    assert ref.he_dims(logn, o.p, logqL, logql) == (dimP, dimA, dimB, dimevk)
    n, W = g.n, (logql + 63) // 64
    rng = random.Random(logqL * 7 + logql)
    rlk0, rlk1 = o.gen(3000, dimevk), o.gen(3001, dimevk)     # NTT-domain key slabs (synthetic)
    cts = [[_centred(rng, logql, n) for _ in range(4)] for _ in range(batch)]
    dev = [to_device(np.concatenate([ints_to_big(cts[k][j], W) for k in range(batch)])) for j in range(4)]
    out0, out1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(out0, out1, dev[0], dev[1], dev[2], dev[3], to_device(rlk0), to_device(rlk1), W, logql, dimA, dimB, dimP)
    got0, got1 = big_to_ints(to_host(out0), W, n), big_to_ints(to_host(out1), W, n)
    for k in range(batch):
        e0, e1 = ref.he_mul(o, (cts[k][0], cts[k][1]), (cts[k][2], cts[k][3]), rlk0[: dimB * n], rlk1[: dimB * n], dimP, dimA, dimB, logql)
        assert got0[k] == e0, "c0 of ciphertext %d" % k
        assert got1[k] == e1, "c1 of ciphertext %d" % k


@pytest.mark.parametrize("logn,logqL,logql", [(7, 120, 120), (7, 200, 100)])
def test_he_swk_matches_reference_semantics(engine_ctx, oracle_ctx, logn, logqL, logql):
    torch = _torch()
    probe = engine_ctx(logn, 12)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    n, W = g.n, (logql + 63) // 64
    rng = random.Random(5)
    swk0, swk1 = o.gen(4000, dimevk), o.gen(4001, dimevk)
    d0, d1 = _centred(rng, logql, n), _centred(rng, logql, n)
    a0, a1 = to_device(ints_to_big(d0, W)), to_device(ints_to_big(d1, W))
    out0, out1 = torch.empty_like(a0), torch.empty_like(a0)
    g.he_swk(out0, out1, a0, a1, to_device(swk0), to_device(swk1), W, logql, dimB, dimP)
    e0, e1 = ref.he_swk(o, d0, d1, swk0[: dimB * n], swk1[: dimB * n], dimP, dimB, logql)
    assert big_to_ints(to_host(out0), W, n)[0] == e0
    assert big_to_ints(to_host(out1), W, n)[0] == e1


@pytest.mark.parametrize("mfma", [True, False, "fused", "direct"])      # matrix-core front + CRT, integer-VALU kernels, the one-pass tail (gpq_set_fused_tail), the one-product tail
@pytest.mark.parametrize("logqL", [120, 200, 438, 610])
def test_relin_tail_rounding_ties_and_wrap_corner(engine_ctx, oracle_ctx, logqL, mfma):
    """mpi_rdiv rounds up only when the remainder is strictly above floor(P/2) (src/types.c:124): key-switch
    outputs are built so that x mod P is floor(P/2)-1, floor(P/2), floor(P/2)+1, 0, P-1, and so that the quotient
    sits exactly on floor(Pi'/2), where the centring of x -- not of the quotient -- decides the sign."""
    torch = _torch()
    logn = 7
    probe = engine_ctx(logn, 12)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logqL)       # dimP = 3 (VALU tail only), 4, 8, 11 (matrix-core front)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    g.set_bridge_mfma(bool(mfma))
    g.set_fused_tail(mfma == "fused")
    n, W, ql = g.n, (logqL + 64) // 64, 1 << logqL
    P = ref.RnsBasis(o.p[:dimP]).P
    PiB = ref.RnsBasis(o.p[:dimB]).P
    Piq = PiB // P
    half, hq = P // 2, Piq // 2
    rng = random.Random(3)
    xs = []
    for i in range(n):
        r = [half - 1, half, half + 1, 0, P - 1][i % 5] if i < 60 else rng.randrange(P)
        if i < 20:
            q = hq                      # quotient on floor(Pi'/2): x wraps (or not) depending on r
        elif i < 40:
            q = hq - 1
        elif i < 50:
            q = -hq - 1 if r else -hq   # most negative representable x
        else:
            q = rng.randrange(-hq + 2, hq - 2)
        xs.append((q * P + r) % PiB)    # the residues only know x mod Pi_B
    chat = np.array([v % o.p[d] for d in range(dimB) for v in xs], dtype=np.uint64)
    dvals = [rng.randrange(-(ql // 2), ql // 2) for _ in range(n)]
    exp = ref.he_relin_tail(o, chat, chat, dvals, None, dimP, dimB, ql)
    out = torch.empty(W * n, dtype=torch.int64, device="cuda")
    tail = g.relin_tail_overwriting if mfma == "direct" else g.relin_tail   # "direct": chat is scratch, the whole tail is one product (gpq_he_mul's form)
    try:
        tail(out, to_device(chat), to_device(ints_to_big(dvals, W)), W, logqL, dimB, dimP)
        assert big_to_ints(to_host(out), W, n)[0] == exp[0]
        tail(out, to_device(chat), None, W, logqL, dimB, dimP)
        assert big_to_ints(to_host(out), W, n)[0] == exp[1]
        inplace = to_device(ints_to_big(dvals, W))               # c0 += d0 in place, as he_swk may be called
        tail(inplace, to_device(chat), inplace, W, logqL, dimB, dimP)
        assert big_to_ints(to_host(inplace), W, n)[0] == exp[0]
    finally:
        g.set_bridge_mfma(True)
        g.set_fused_tail(False)


def test_he_mul_by_one_is_identity_full_size(engine_ctx):
    """Size-independent property at the headline shape (n = 2^16, q = 2^850, dimA/dimB = 30/45):
    multiplying by the trivial ciphertext (c0 = 1, c1 = 0) gives d0 = ct.c0, d1 = ct.c1, d2 = 0, so the
    relinearisation adds rdiv(0) = 0 and he_mul must return ct itself, whatever the key."""
    torch = _torch()
    logn, logq = 16, 850
    g = engine_ctx(logn, 45)
    dimP, dimA, dimB, dimevk = g.he_dims(logq, logq)
    assert (dimP, dimA, dimB) == (15, 30, 45)
    n, W = g.n, 14
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (W, n), dtype=torch.int64, device="cuda", generator=gen)
        big[W - 1] = torch.randint(-(1 << 16), 1 << 16, (n,), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    c0, c1 = centred(), centred()
    one = torch.zeros(W * n, dtype=torch.int64, device="cuda")
    one[0] = 1
    zero = torch.zeros_like(one)
    rlk = [torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]
    rlk0 = torch.cat(rlk)
    rlk1 = torch.cat(rlk[::-1])
    o0, o1 = torch.empty_like(c0), torch.empty_like(c0)
    g.he_mul(o0, o1, c0, c1, one, zero, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    assert torch.equal(o0, c0) and torch.equal(o1, c1)
    g.he_mul(o0, o1, one, zero, c0, c1, rlk0, rlk1, W, logq, dimA, dimB, dimP)
    assert torch.equal(o0, c0) and torch.equal(o1, c1)


def test_he_swk_with_the_key_P_at_config5_size(engine_ctx):
    """BASELINE configs[4] shape (n = 2^17, 44 limbs: q = 2^835 gives dimP/dimB = 15/44), size-independent property: with
    swk.p0 = the constant polynomial P (its residues in every NTT slot; zero in the limbs of P itself) and swk.p1 = 0 the key
    switch computes d1 * P, the division by P gives d1 back exactly, so he_swk returns (d0 + d1, 0) centred mod q.  Runs the
    split-twiddle and the 7-mad butterflies in one transform (the chain's larger primes) and the matrix-core tail."""
    torch = _torch()
    logn, logq = 17, 835
    g = engine_ctx(logn, 44)
    dimP = (logq + 1 + logn) // 59 + 1
    P = 1
    for d in range(dimP):
        P *= g.p[d]
    dimB = (logq + 1 + P.bit_length() + logq + logn) // 59 + 1
    assert (dimP, dimB) == (15, 44)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(17)

    def centred(bits):
        big = torch.randint(-(1 << 62), 1 << 62, (W, n), dtype=torch.int64, device="cuda", generator=gen)
        top = bits - 64 * (W - 1)
        big[W - 1] = torch.randint(-(1 << (top - 1)), 1 << (top - 1), (n,), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    d0, d1 = centred(logq - 2), centred(logq - 2)            # |d0 + d1| < q/2: no wrap in the expected sum
    swk0 = torch.cat([torch.full((n,), P % g.p[d], dtype=torch.int64, device="cuda") for d in range(dimB)])
    swk1 = torch.zeros_like(swk0)
    o0, o1 = torch.empty_like(d0), torch.empty_like(d0)
    g.he_swk(o0, o1, d0, d1, swk0, swk1, W, logq, dimB, dimP)
    host0, host1 = to_host(d0).reshape(W, n), to_host(d1).reshape(W, n)
    exp = np.zeros((W, n), dtype=np.uint64)
    carry = np.zeros(n, dtype=np.uint64)
    for j in range(W):                                        # multiword d0 + d1 with numpy (two's complement)
        a, b = host0[j].astype(np.uint64), host1[j].astype(np.uint64)
        s1 = a + b
        c1 = (s1 < a).astype(np.uint64)
        s2 = s1 + carry
        c2 = (s2 < s1).astype(np.uint64)
        exp[j] = s2
        carry = c1 + c2
    assert np.array_equal(to_host(o0).reshape(W, n).astype(np.uint64), exp)
    assert not to_host(o1).any()


@pytest.mark.parametrize("logn,W", [(7, 2), (9, 14)])
def test_poly_rot_and_conj_on_big_slabs(engine_ctx, logn, W):
    """src/poly.c:263-283 as signed permutations of big slabs (two polynomials per call)."""
    torch = _torch()
    g = engine_ctx(logn, 5)
    n = g.n
    rng = random.Random(8 + W)
    lim = 1 << (64 * W - 2)
    polys = [[rng.randrange(-lim, lim) for _ in range(n)] for _ in range(2)]
    polys[0][:3] = [0, -1, lim - 1]
    a = to_device(np.concatenate([ints_to_big(p, W) for p in polys]))
    r = torch.empty_like(a)
    for rot in (0, 1, 3, 7):
        g.poly_rot(r, a, W, rot)
        assert big_to_ints(to_host(r), W, n) == [ref.poly_rot(p, rot) for p in polys]
    g.poly_conj(r, a, W)
    assert big_to_ints(to_host(r), W, n) == [ref.poly_conj(p) for p in polys]


def test_he_mulpt_matches_reference_semantics(engine_ctx, oracle_ctx):
    torch = _torch()
    logn, logql, dim = 8, 150, 4
    g, o = engine_ctx(logn, 12), oracle_ctx(logn, 12)
    n, W = g.n, 3
    rng = random.Random(17)
    ct = [_centred(rng, logql, n) for _ in range(2)]
    m = [rng.randrange(-(1 << 40), 1 << 40) for _ in range(n)]
    d0, d1, dm = (to_device(ints_to_big(v, W)) for v in (ct[0], ct[1], m))
    o0, o1 = torch.empty_like(d0), torch.empty_like(d0)
    g.he_mulpt(o0, o1, d0, d1, dm, W, logql, dim)
    e0, e1 = ref.he_mulpt(o, ct, m, dim, logql)
    assert big_to_ints(to_host(o0), W, n)[0] == e0 and big_to_ints(to_host(o1), W, n)[0] == e1


def _general_setup(logn, qL, ql):
    """dims as src/precomp.c:401,407 and src/he-mult.c:99,51 compute them from bit lengths, for arbitrary moduli"""
    nbL, nbl = qL.bit_length(), ql.bit_length()
    dimP = (nbL + logn) // 59 + 1
    return dimP, nbL, nbl


@pytest.mark.parametrize("logn,Delta,L,lvl", [(7, 1000003, 5, 5), (7, (1 << 30) - 35, 4, 3)])
def test_general_moduli_he_mul_swk_rs(engine_ctx, oracle_ctx, logn, Delta, L, lvl):
    """Delta not a power of two: q_L = Delta^L * 2^20-ish odd modulus, q_l = floor(q_{l+1}/Delta) (src/precomp.c:394-400)."""
    torch = _torch()
    qL = Delta ** L * 1048573
    q = [0] * (L + 1)
    cur = qL
    for l in range(L, -1, -1):
        q[l] = cur
        cur //= Delta
    ql = q[lvl]
    probe = engine_ctx(logn, 12)
    dimP, nbL, nbl = _general_setup(logn, qL, ql)
    P = ref.RnsBasis(probe.p[:dimP]).P
    nbPqL = (P * qL).bit_length()
    dimA = (2 * nbl + logn) // 59 + 1
    dimB = (nbl + nbPqL + logn) // 59 + 1
    dimevk = (nbL + nbPqL + logn) // 59 + 1
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    n, W = g.n, ql.bit_length() // 64 + 1
    rng = random.Random(Delta % 9973)
    h = ql // 2
    cts = [[rng.randrange(-h, ql - h) for _ in range(n)] for _ in range(4)]   # centred as mpi_smod leaves them
    cts = [[ref.mpi_smod(v, ql) for v in c] for c in cts]
    rlk0, rlk1 = o.gen(3000, dimevk), o.gen(3001, dimevk)
    dev = [to_device(ints_to_big(c, W)) for c in cts]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul_general(o0, o1, *dev, to_device(rlk0), to_device(rlk1), W, ql, dimA, dimB, dimP)
    e0, e1 = ref.he_mul(o, (cts[0], cts[1]), (cts[2], cts[3]), rlk0[: dimB * n], rlk1[: dimB * n], dimP, dimA, dimB, 0, ql=ql)
    assert big_to_ints(to_host(o0), W, n)[0] == e0 and big_to_ints(to_host(o1), W, n)[0] == e1
    # he_swk on the product
    g.he_swk_general(o0, o1, to_device(ints_to_big(e0, W)), to_device(ints_to_big(e1, W)), to_device(rlk1), to_device(rlk0), W, ql, dimB, dimP)
    s0, s1 = ref.he_swk(o, e0, e1, rlk1[: dimB * n], rlk0[: dimB * n], dimP, dimB, 0, ql=ql)
    assert big_to_ints(to_host(o0), W, n)[0] == s0 and big_to_ints(to_host(o1), W, n)[0] == s1
    # he_rs: mpi_rdiv by Delta (ties: remainder == Delta/2 exactly only when Delta is even) then mpi_smod q_{l-1}
    c0 = to_device(ints_to_big(e0, W))
    c1 = to_device(ints_to_big(e1, W))
    g.he_rs_general(c0, c1, W, Delta, q[lvl - 1])
    assert big_to_ints(to_host(c0), W, n)[0] == [ref.mpi_smod(ref.mpi_rdiv(v, Delta), q[lvl - 1]) for v in e0]
    assert big_to_ints(to_host(c1), W, n)[0] == [ref.mpi_smod(ref.mpi_rdiv(v, Delta), q[lvl - 1]) for v in e1]


def test_rdiv_word_ties(engine_ctx):
    """even Delta: remainder exactly Delta/2 must not round up; negative values floor first (src/types.c:115-128)."""
    g = engine_ctx(7, 5)
    n, W, Delta = g.n, 2, 1000
    ql = 10 ** 30 + 57
    vals = [0, 499, 500, 501, 1500, -1, -499, -500, -501, -1500, -1000, 999999, -999999, 10 ** 25 + 500, -(10 ** 25) - 500]
    vals += [(-1) ** k * (k * 7919 + 250 * k) for k in range(n - len(vals))]
    c0, c1 = to_device(ints_to_big(vals, W)), to_device(ints_to_big([-v for v in vals], W))
    g.he_rs_general(c0, c1, W, Delta, ql)
    assert big_to_ints(to_host(c0), W, n)[0] == [ref.mpi_smod(ref.mpi_rdiv(v, Delta), ql) for v in vals]
    assert big_to_ints(to_host(c1), W, n)[0] == [ref.mpi_smod(ref.mpi_rdiv(-v, Delta), ql) for v in vals]


def test_he_mul_headline_dims_against_restated_reference(engine_ctx, oracle_ctx):
    """The headline limb counts (q = 2^850: dimP/dimA/dimB = 15/30/45, 14-word coefficients) through the two-pass NTT
    kernels and the fast CRT path, at n = 2^13 so that the Python-integer side of the restated reference finishes in
    seconds: every coefficient of he_mul must match src/he-mult.c:88-156, then he_rs src/he-rescale.c:33-54."""
    torch = _torch()
    logn, logq, logDelta = 13, 850, 50
    probe = engine_ctx(logn, 16)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    assert (dimP, dimA, dimB, dimevk) == (15, 30, 45, 45)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    n, W = g.n, 14
    rng = random.Random(850)
    h = 1 << (logq - 1)
    cts = [[rng.randrange(-h, h) for _ in range(n)] for _ in range(4)]
    rlk0, rlk1 = o.gen(3000, dimevk), o.gen(3001, dimevk)
    dev = [to_device(ints_to_big(c, W)) for c in cts]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(o0, o1, *dev, to_device(rlk0), to_device(rlk1), W, logq, dimA, dimB, dimP)
    e0, e1 = ref.he_mul(o, (cts[0], cts[1]), (cts[2], cts[3]), rlk0, rlk1, dimP, dimA, dimB, logq)
    assert big_to_ints(to_host(o0), W, n)[0] == e0
    assert big_to_ints(to_host(o1), W, n)[0] == e1
    g.he_rs(o0, o1, W, logDelta, logq - logDelta)
    ql = 1 << (logq - logDelta)
    assert big_to_ints(to_host(o0), W, n)[0] == [ref.mpi_smod(ref.mpi_rdiv(v, 1 << logDelta), ql) for v in e0]
    assert big_to_ints(to_host(o1), W, n)[0] == [ref.mpi_smod(ref.mpi_rdiv(v, 1 << logDelta), ql) for v in e1]


def test_he_mul_is_graph_capturable_after_the_first_call(engine_ctx):
    """After one warm-up call (tables are built on first use) gpq_he_mul is a pure sequence of kernel launches on the caller's
    stream: it can be captured into a HIP graph and replayed on new inputs."""
    torch = _torch()
    logn, logq = 13, 438
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(31)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (W, n), dtype=torch.int64, device="cuda", generator=gen)
        big[W - 1] = torch.randint(-(1 << 20), 1 << 20, (n,), dtype=torch.int64, device="cuda", generator=gen)   # 438 - 6*64 = 54 bits
        return big.reshape(-1).contiguous()

    ins = [centred() for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    o0, o1 = torch.empty_like(ins[0]), torch.empty_like(ins[0])
    nbytes = g.lib.gpq_he_mul_workspace_bytes(g.h, W, dimA, dimB, dimP, 1)
    ws = torch.empty(nbytes // 8 + 8, dtype=torch.int64, device="cuda")

    def call():
        from gpqhe_amd import _native
        from gpqhe_amd.engine import _ptr, _stream
        _native.check(g.lib.gpq_he_mul(g.h, _ptr(o0), _ptr(o1), *[_ptr(v) for v in ins], _ptr(rlk[0]), _ptr(rlk[1]), W, logq, dimA, dimB, dimP,
                                       1, _ptr(ws), _stream()), "gpq_he_mul")

    call()                                               # warm-up: builds the CRT / matrix tables
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    fresh = [centred() for _ in range(4)]
    for dst, src in zip(ins, fresh):
        dst.copy_(src)
    graph.replay()
    torch.cuda.synchronize()
    r0, r1 = o0.clone(), o1.clone()
    call()
    torch.cuda.synchronize()
    assert torch.equal(r0, o0) and torch.equal(r1, o1)
    assert bool((r0 != 0).any())

    # A later eager call at a LARGER batch grows the context's CRT flag scratch.  The graph captured above still holds the
    # old buffer: it is retired, not freed, so the replay stays valid and exact.
    big_ins = [torch.cat([v, v, v]) for v in ins]
    bo0, bo1 = torch.empty_like(big_ins[0]), torch.empty_like(big_ins[0])
    g.he_mul(bo0, bo1, *big_ins, rlk[0], rlk[1], W, logq, dimA, dimB, dimP)
    torch.cuda.synchronize()
    assert torch.equal(bo0[: o0.numel()], r0) and torch.equal(bo0[2 * o0.numel():], r0) and torch.equal(bo1[o1.numel(): 2 * o1.numel()], r1)
    o0.zero_(); o1.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(r0, o0) and torch.equal(r1, o1)


def _sparse_mul(dense, terms, n):
    out = [0] * n
    for k, c in terms:
        for i, v in enumerate(dense):
            j = i + k
            if j < n:
                out[j] += c * v
            else:
                out[j - n] -= c * v
    return out


@pytest.mark.parametrize("logn,logq", [(13, 438), (16, 850)])
def test_he_mul_bit_exact_at_full_size_through_sparse_operands(engine_ctx, oracle_ctx, logn, logq):
    """Exact parity of the whole he_mul (src/he-mult.c:88-156) at the headline shape: the second ciphertext and the two key
    polynomials have a few (full-size) coefficients each, so every polynomial product of the reference's formula is a dense-by-
    sparse product Python integers do exactly:  d0 = c0 c0', d1 = c0 c1' + c1 c0', d2 = c1 c1' (smod q);
    out_i = smod(d_i + rdiv(smod(d2 * rlk.p_i, P q_L), P), q)."""
    torch = _torch()
    n, q = 1 << logn, 1 << logq
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    P = ref.RnsBasis(o.p[:dimP]).P
    PqL = P * q
    rng = random.Random(4321 + logn)
    sparse = lambda cnt, lim: sorted({rng.randrange(n): rng.randrange(-lim, lim) for _ in range(cnt)}.items())
    dense_of = lambda terms: [dict(terms).get(i, 0) for i in range(n)]
    c0 = [rng.randrange(-(q >> 1), q >> 1) for _ in range(n)]
    c1 = [rng.randrange(-(q >> 1), q >> 1) for _ in range(n)]
    t0, t1 = sparse(5, q >> 1), sparse(5, q >> 1)                            # ct2 = (c0', c1')
    k0, k1 = sparse(6, PqL >> 1), sparse(6, PqL >> 1)                        # rlk.p0, rlk.p1 as polynomials
    smod = ref.mpi_smod
    d0 = [smod(v, q) for v in _sparse_mul(c0, t0, n)]
    d2 = [smod(v, q) for v in _sparse_mul(c1, t1, n)]
    d1 = [smod(x + y, q) for x, y in zip(_sparse_mul(c0, t1, n), _sparse_mul(c1, t0, n))]
    exp0 = [smod(a + ref.mpi_rdiv(smod(b, PqL), P), q) for a, b in zip(d0, _sparse_mul(d2, k0, n))]
    exp1 = [smod(a + ref.mpi_rdiv(smod(b, PqL), P), q) for a, b in zip(d1, _sparse_mul(d2, k1, n))]
    slab = lambda poly: o.ntt_slab(np.array([v % o.p[d] for d in range(dimevk) for v in poly], dtype=np.uint64), dimevk)
    W = logq // 64 + 1
    dev = [to_device(ints_to_big(v, W)) for v in (c0, c1, dense_of(t0), dense_of(t1))]
    o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
    g.he_mul(o0, o1, *dev, to_device(slab(dense_of(k0))), to_device(slab(dense_of(k1))), W, logq, dimA, dimB, dimP)
    assert big_to_ints(to_host(o0), W, n)[0] == exp0
    assert big_to_ints(to_host(o1), W, n)[0] == exp1


@pytest.mark.parametrize("logn,logq", [(14, 438), (17, 835)])
def test_he_swk_bit_exact_at_full_size_through_sparse_keys(engine_ctx, oracle_ctx, logn, logq):
    """Exact parity of he_swk (src/he-automorphism.c:40-85) up to BASELINE configs[4]'s shape (n = 2^17, 44 limbs): the key
    polynomials have a few full-size coefficients, so out_0 = smod(d0 + rdiv(smod(d1 k0, P q_L), P), q) and
    out_1 = smod(rdiv(smod(d1 k1, P q_L), P), q) are exact Python integers."""
    torch = _torch()
    n, q = 1 << logn, 1 << logq
    dimP = (logq + 1 + logn) // 59 + 1
    g0 = engine_ctx(logn, dimP)
    P = 1
    for d in range(dimP):
        P *= g0.p[d]
    PqL = P * q
    dimB = (logq + 1 + PqL.bit_length() + logn) // 59 + 1
    g, o = engine_ctx(logn, dimB), oracle_ctx(logn, dimB)
    rng = random.Random(987 + logn)
    sparse = lambda cnt, lim: sorted({rng.randrange(n): rng.randrange(-lim, lim) for _ in range(cnt)}.items())
    dense_of = lambda terms: [dict(terms).get(i, 0) for i in range(n)]
    d0 = [rng.randrange(-(q >> 1), q >> 1) for _ in range(n)]
    d1 = [rng.randrange(-(q >> 1), q >> 1) for _ in range(n)]
    k0, k1 = sparse(6, PqL >> 1), sparse(6, PqL >> 1)
    smod = ref.mpi_smod
    exp0 = [smod(a + ref.mpi_rdiv(smod(b, PqL), P), q) for a, b in zip(d0, _sparse_mul(d1, k0, n))]
    exp1 = [smod(ref.mpi_rdiv(smod(b, PqL), P), q) for b in _sparse_mul(d1, k1, n)]
    slab = lambda poly: o.ntt_slab(np.array([v % o.p[d] for d in range(dimB) for v in poly], dtype=np.uint64), dimB)
    W = logq // 64 + 1
    a0, a1 = to_device(ints_to_big(d0, W)), to_device(ints_to_big(d1, W))
    o0, o1 = torch.empty_like(a0), torch.empty_like(a0)
    g.he_swk(o0, o1, a0, a1, to_device(slab(dense_of(k0))), to_device(slab(dense_of(k1))), W, logq, dimB, dimP)
    assert big_to_ints(to_host(o0), W, n)[0] == exp0
    assert big_to_ints(to_host(o1), W, n)[0] == exp1


def test_he_mul_squaring_path_equals_the_general_path(engine_ctx):
    """gpq_he_mul with ct2 the same slabs as ct1 (he_mul(&ct, &ct, &ct, rlk), src/he-algo.c:151) decomposes and transforms the ciphertext
    once; the result equals the call with copies of the operands (general path), batch 3, n = 2^13, q = 2^438."""
    torch = _torch()
    logn, logq = 13, 438
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W, batch = g.n, (logq + 64) // 64, 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(151)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
        big[:, W - 1] = torch.randint(-(1 << 20), 1 << 20, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    c0, c1 = centred(), centred()
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    s0, s1 = torch.empty_like(c0), torch.empty_like(c0)
    g.he_mul(s0, s1, c0, c1, c0, c1, rlk[0], rlk[1], W, logq, dimA, dimB, dimP)                       # aliased: squaring path
    t0, t1 = torch.empty_like(c0), torch.empty_like(c0)
    g.he_mul(t0, t1, c0, c1, c0.clone(), c1.clone(), rlk[0], rlk[1], W, logq, dimA, dimB, dimP)       # copies: general path
    torch.cuda.synchronize()
    assert torch.equal(s0, t0) and torch.equal(s1, t1) and bool((s0 != 0).any())


@pytest.mark.parametrize("logn,logq,batch", [(13, 438, 3), (16, 850, 2), (14, 300, 2)])
def test_one_pass_relinearisation_tail_equals_the_two_kernel_form(engine_ctx, logn, logq, batch):
    """gpq_set_fused_tail(ctx, 1): bridge_relin_tail_mfma (front + CRT of Q in one pass per coefficient, Q's residues never in memory)
    against the default two-kernel tail on dense random ciphertexts, whole he_mul and he_swk (c1 without addend), at the headline
    shape too: identical words."""
    torch = _torch()
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(977 + logn)

    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
        top = logq - 1 - 64 * (W - 1)
        big[:, W - 1] = torch.randint(-(1 << (top - 1)), 1 << (top - 1), (batch, n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()

    cts = [centred() for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    outs = []
    try:
        # ... and with / without the inverse transforms pre-multiplying their output by the CRT weights (gpq_set_prescale)
        # (0 = no scaling, 1 = the CRT weights on the limbs of each basis, 2 = also w_j on the limbs above P for the relinearisation front)
        #  3 = every limb of the key switch by the weights of its whole basis: the relinearisation tail as ONE product)
        for fused, prescale in ((False, 3), (False, 2), (True, 2), (False, 0), (True, 0), (False, 1), (True, 1)):
            g.set_fused_tail(fused)
            g.set_prescale(prescale)
            o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
            g.he_mul(o0, o1, *cts, rlk[0], rlk[1], W, logq, dimA, dimB, dimP)
            s0, s1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
            g.he_swk(s0, s1, cts[0], cts[1], rlk[0], rlk[1], W, logq, dimB, dimP)
            torch.cuda.synchronize()
            outs.append((o0, o1, s0, s1))
    finally:
        g.set_fused_tail(False)
        g.set_prescale(3)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert bool((outs[0][0] != 0).any()) and bool((outs[0][3] != 0).any())
