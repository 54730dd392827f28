"""Device MPI<->RNS bridge against the Python-integer restatement of the reference's
libgcrypt code (oracle/bigint_ref.py): rns_decompose, poly_rns2mpi, poly_mul at the
MPI level (the tests/polymul.c KAT entirely on the device) and he_rs."""
import random

import numpy as np
import pytest

from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host
from oracle.bigint_ref import (RnsBasis, centred_mod, mpi_rdiv, mpi_smod, negacyclic_mul, poly_rns2mpi,
                               rns_decompose)

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _edge_values(bits, rng, count):
    """signed values of at most `bits` bits (two's complement), edges first"""
    lim = 1 << (bits - 1)
    vals = [0, 1, -1, lim - 1, -lim, lim // 2, -(lim // 2), (1 << 59) - 1, -(1 << 59), (1 << 64), -(1 << 64) + 1]
    vals = [v for v in vals if -lim <= v < lim]
    while len(vals) < count:
        vals.append(rng.randrange(-lim, lim))
    return vals[:count]


@pytest.mark.parametrize("mfma", [True, False])
@pytest.mark.parametrize("logn,dim,W,bits", [(7, 5, 1, 62), (7, 5, 2, 128), (7, 3, 4, 200), (8, 6, 14, 851), (7, 4, 16, 1024), (7, 2, 28, 1737),
                                             (6, 7, 4, 256), (7, 9, 8, 512), (7, 13, 28, 1737), (7, 8, 32, 2047), (6, 45, 14, 851)])
def test_rns_decompose_matches_floor_mod(engine_ctx, logn, dim, W, bits, mfma):
    """both implementations: bytes x (256^k mod p) on the matrix cores, and the 59-bit-digit Horner on the VALU"""
    g = engine_ctx(logn, max(dim, 6))
    g.set_bridge_mfma(mfma)
    try:
        rng = random.Random(1234 + W)
        polys = [_edge_values(bits, rng, g.n) for _ in range(2)]
        big = np.concatenate([ints_to_big(v, W) for v in polys])
        slab = _torch().empty(2 * dim * g.n, dtype=_torch().int64, device="cuda")
        g.rns_decompose(slab, to_device(big), W, dim)
        got = to_host(slab).reshape(2, dim, g.n)
        for k in range(2):
            for d in range(dim):
                assert [int(x) for x in got[k, d]] == rns_decompose(polys[k], g.p[d]), (k, d)  # src/rns.c:44-45
    finally:
        g.set_bridge_mfma(True)


@pytest.mark.parametrize("dim,W,batch", [(30, 14, 3), (45, 14, 2), (15, 14, 1), (58, 16, 1)])
def test_rns_decompose_matrix_core_equals_valu_at_full_size(engine_ctx, dim, W, batch):
    """n = 2^16: every word pattern incl. all-ones / all-zero bytes (the extremes of the i32 accumulators)"""
    torch = _torch()
    g = engine_ctx(16, 58)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77 + dim)
    big = torch.randint(-(1 << 63), (1 << 63) - 1, (batch, W, g.n), dtype=torch.int64, device="cuda", generator=gen)
    big[0, :, 0:64] = -1                # x = -1: every byte 0xff
    big[0, :, 64:128] = 0
    big[0, :, 128:192] = 0x7f7f7f7f7f7f7f7f
    big[0, :, 192:256] = -0x7f7f7f7f7f7f7f80   # 0x8080...80
    big = big.reshape(-1).contiguous()
    out = []
    try:
        for mfma in (True, False):
            g.set_bridge_mfma(mfma)
            slab = torch.empty(batch * dim * g.n, dtype=torch.int64, device="cuda")
            g.rns_decompose(slab, big, W, dim)
            out.append(slab)
    finally:
        g.set_bridge_mfma(True)
    assert torch.equal(out[0], out[1])
    host = to_host(out[0]).reshape(batch, dim, g.n)
    words = to_host(big).reshape(batch, W, g.n)
    for i in (0, 64, 128, 192, 300):     # spot values against Python integers
        v = big_to_ints(np.ascontiguousarray(words[0][:, i:i + 1]).reshape(-1), W, 1)[0][0]
        assert [int(host[0, d, i]) for d in range(dim)] == [v % g.p[d] for d in range(dim)]


def test_phat_invmp_matches_reference_printout(golden, engine_ctx):
    g = engine_ctx(7, 5)
    for dim, exp in enumerate(golden["phat_invmp_logn7"], start=1):  # tests/polymul.c:106-112 output
        assert [str(v) for v in g.phat_invmp(dim)] == exp


@pytest.mark.parametrize("logn,dim,logq", [(7, 1, 30), (7, 2, 61), (7, 5, 61), (7, 5, 64), (7, 5, 0), (8, 9, 200), (7, 30, 850), (7, 45, 850), (7, 45, 0)])
def test_poly_rns2mpi_matches_bigint(engine_ctx, oracle_ctx, logn, dim, logq):
    g, o = engine_ctx(logn, dim), oracle_ctx(logn, dim)
    basis = RnsBasis(g.p[:dim])
    rng = random.Random(99 + dim)
    n = g.n
    slab = o.gen(5 + dim, dim).reshape(dim, n).copy()
    # edge residues: 0, p-1, and a centred value just around P/2
    half = basis.P_2
    for col, v in enumerate((0, basis.P - 1, half - 1, half, half + 1, 1, (1 << logq) // 2 if logq else 7)):
        for d in range(dim):
            slab[d, col] = v % g.p[d]
    exp = poly_rns2mpi([slab[d] for d in range(dim)], basis, 1 << logq) if logq else \
        [mpi_smod(sum(int(slab[d][i]) * ((basis.phat[d] * basis.phat_invmp[d]) % basis.P) for d in range(dim)) % basis.P, basis.P)
         for i in range(n)]
    Wout = (logq + 63) // 64 if logq else (basis.P.bit_length() + 1 + 63) // 64
    big = _torch().empty(Wout * n, dtype=_torch().int64, device="cuda")
    g.rns_reconstruct(big, Wout, to_device(slab.reshape(-1)), dim, logq)
    assert big_to_ints(to_host(big), Wout, n)[0] == exp
    assert rng is not None


def test_polymul_kat_entirely_on_device(engine_ctx):
    """tests/polymul.c + tests/polymul.gp: n = 128, dimub = 5 limbs, q = 2^61."""
    g = engine_ctx(7, 5)
    N, Q = 128, 1 << 61
    cases = [([i + 2 for i in range(N)], [i + 3 for i in range(N)]),
             ([g.p[0] - i - 1 for i in range(N)], [g.p[1] - i - 1 for i in range(N)])]
    leading = [[382784, 357372, 332350], [18559595904, 18272672062, 17985699960]]
    W = 1
    a = to_device(np.concatenate([ints_to_big(c[0], W) for c in cases]))
    b = to_device(np.concatenate([ints_to_big(c[1], W) for c in cases]))
    r = _torch().empty_like(a)
    g.poly_mul(r, a, b, W, 5, 61)  # src/poly.c:84-107 with dim = polyctx.dimub
    got = big_to_ints(to_host(r), W, N)
    for (ca, cb), res, lead in zip(cases, got, leading):
        assert res == [centred_mod(v, Q) for v in negacyclic_mul(ca, cb)]
        assert [res[127], res[126], res[125]] == lead


@pytest.mark.parametrize("logn,W,s,logql", [(7, 1, 10, 40), (7, 2, 50, 70), (7, 14, 50, 800), (7, 14, 64, 700), (7, 3, 128, 60), (7, 14, 30, 820)])
def test_he_rs_matches_rdiv_smod(engine_ctx, logn, W, s, logql):
    g = engine_ctx(logn, 5)
    rng = random.Random(7 * W + s)
    bits = min(64 * W, logql + s)
    vals = _edge_values(bits, rng, g.n)
    half = 1 << (s - 1)
    # ties and near-ties of the rounding rule: remainder == Delta/2 does NOT round up (src/types.c:124)
    vals[11:17] = [half, half + 1, half - 1, -half, -half + 1, 3 * half]
    c0 = to_device(ints_to_big(vals, W))
    c1 = to_device(ints_to_big([-v for v in vals], W))
    g.he_rs(c0, c1, W, s, logql)
    exp0 = [mpi_smod(mpi_rdiv(v, 1 << s), 1 << logql) for v in vals]      # src/he-rescale.c:45-48
    exp1 = [mpi_smod(mpi_rdiv(-v, 1 << s), 1 << logql) for v in vals]
    assert big_to_ints(to_host(c0), W, g.n)[0] == exp0
    assert big_to_ints(to_host(c1), W, g.n)[0] == exp1


def test_crt_matrix_core_equals_valu_at_full_size(engine_ctx):
    """n = 2^16, 30 and 45 limbs, uniform residues (values all over [0, P)): both fast paths must agree word for word"""
    torch = _torch()
    g = engine_ctx(16, 58)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    for dim, logq, batch in ((30, 850, 2), (45, 850, 1), (15, 438, 2)):
        W = (logq + 63) // 64
        slab = torch.empty((batch, dim, g.n), dtype=torch.int64, device="cuda")
        for d in range(dim):
            slab[:, d, :] = torch.randint(0, g.p[d], (batch, g.n), dtype=torch.int64, device="cuda", generator=gen)
        slab[0, :, :128] = 0                                   # x = 0
        for d in range(dim):
            slab[0, d, 128:256] = g.p[d] - 1                   # x = P - 1 (== -1)
        slab = slab.reshape(-1).contiguous()
        out = []
        try:
            for mfma in (True, False):
                g.set_bridge_mfma(mfma)
                big = torch.empty(batch * W * g.n, dtype=torch.int64, device="cuda")
                g.rns_reconstruct(big, W, slab, dim, logq)
                out.append(big)
        finally:
            g.set_bridge_mfma(True)
        assert torch.equal(out[0], out[1]), (dim, logq)
        vals = big_to_ints(to_host(out[0]), W, g.n)[0]
        assert vals[0] == 0 and vals[200] == -1


def test_roundtrip_full_size(engine_ctx):
    """Size-independent property at the headline shape (n = 2^16, 30 limbs, 851-bit coefficients):
    poly_rns2mpi(rns_decompose(a)) == a for every centred a, including negatives."""
    torch = _torch()
    logn, dim, W, logq = 16, 30, 14, 850
    g = engine_ctx(logn, 45)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    big = torch.randint(-(1 << 62), 1 << 62, (W, g.n), dtype=torch.int64, device="cuda", generator=gen)
    top = torch.randint(-(1 << 16), 1 << 16, (g.n,), dtype=torch.int64, device="cuda", generator=gen)  # 850 - 13*64 = 18 bits
    big[W - 1] = top
    big = big.reshape(-1).contiguous()
    slab = torch.empty(dim * g.n, dtype=torch.int64, device="cuda")
    g.rns_decompose(slab, big, W, dim)
    back = torch.empty_like(big)
    g.rns_reconstruct(back, W, slab, dim, logq)
    assert torch.equal(back, big)
    assert bool((slab >= 0).all())


@pytest.mark.parametrize("mfma", [True, False])
@pytest.mark.parametrize("dim,logq", [(30, 850), (30, 61), (45, 850), (9, 200), (15, 100), (30, 440), (45, 620), (58, 1000), (12, 128)])
def test_fast_crt_path_equals_exact_kernel(engine_ctx, oracle_ctx, dim, logq, mfma):
    """gpq_rns_reconstruct's low-word fast path (fixed-point quotient, flagged coefficients redone exactly)
    against the full-width kernel on the same slabs, including residues of values that sit on the
    rounding boundaries (x = k*P/2 +- small), where the fast path must hand over to the exact one."""
    torch = _torch()
    logn = 8
    g, o = engine_ctx(logn, 58), oracle_ctx(logn, 58)
    g.set_bridge_mfma(mfma)          # the CRT sum on the matrix cores (default) or on the VALU
    basis = RnsBasis(g.p[:dim])
    n = g.n
    slab = o.gen(77, dim).reshape(dim, n).copy()
    half = basis.P_2
    edge = [half, half - 1, half + 1, 0, basis.P - 1, 1, half + (1 << 40), half - (1 << 40), half - (1 << 900) if dim > 20 else half - 5]
    for col, v in enumerate(edge):
        for d in range(dim):
            slab[d, col] = v % g.p[d]
    W = (logq + 63) // 64
    dev = to_device(slab.reshape(-1))
    fast = torch.empty(W * n, dtype=torch.int64, device="cuda")
    exact = torch.empty_like(fast)
    try:
        g.rns_reconstruct(fast, W, dev, dim, logq)
        g.set_exact_crt(True)
        g.rns_reconstruct(exact, W, dev, dim, logq)
    finally:
        g.set_exact_crt(False)
        g.set_bridge_mfma(True)
    assert torch.equal(fast, exact)
    exp = poly_rns2mpi([slab[d][:16] for d in range(dim)], basis, 1 << logq)
    assert big_to_ints(to_host(fast), W, n)[0][:16] == exp


@pytest.mark.parametrize("logn,dim,q", [
    (7, 5, (1 << 61)),                       # power of two through the general path: must equal the KAT
    (7, 5, 3 * (1 << 61) + 7),               # one-word odd modulus, CRT value many times longer
    (7, 5, (1 << 100) - 3),
    (7, 9, None),                            # q = P*q_L as he_genswk uses (src/he-kem.c:80,95), built below
])
def test_poly_mul_general_modulus(engine_ctx, oracle_ctx, logn, dim, q):
    torch = _torch()
    g = engine_ctx(logn, max(dim, 9))
    n = g.n
    if q is None:
        P = RnsBasis(g.p[:3]).P
        q = P << 120                          # P * 2^120
        dim = (q.bit_length() + logn) // 59 + 1
    rng = random.Random(q % 1000003)
    h = q // 2
    a = [rng.randrange(-h, h) for _ in range(n)]
    b = [rng.randrange(-3, 4) for _ in range(n)]           # small second factor, like a secret key
    a[:3] = [0, h - 1, -h]
    W = (q.bit_length() + 64) // 64
    da, db = to_device(ints_to_big(a, W)), to_device(ints_to_big(b, W))
    r = torch.empty_like(da)
    g.poly_mul_general(r, da, db, W, dim, q)
    exp = [mpi_smod(v, q) for v in negacyclic_mul(a, b)]   # |a*b| < n*3*q/2 < P/2 of the dim-limb basis
    assert big_to_ints(to_host(r), W, n)[0] == exp


@pytest.mark.parametrize("logn", [13, 16])
def test_poly_mul_general_modulus_exact_at_full_size(engine_ctx, logn):
    """poly_mul with q = P q_L (odd, 1700+ bits) as he_genswk calls it (src/he-kem.c:95), n up to 2^16: dense uniform a times a
    polynomial with a few small terms (the secret's shape), expected = smod(a * s, q) exactly."""
    torch = _torch()
    g = engine_ctx(logn, 45)
    n = g.n
    P = 1
    for d in range(15):
        P *= g.p[d]
    q = P << 850
    rng = random.Random(66 + logn)
    a = [rng.randrange(-(q >> 1), q >> 1) for _ in range(n)]
    terms = sorted({rng.randrange(n): rng.choice((-1, 1)) for _ in range(24)}.items())
    s = [dict(terms).get(i, 0) for i in range(n)]
    want = [0] * n
    for k, c in terms:
        for i, v in enumerate(a):
            j = i + k
            if j < n:
                want[j] += c * v
            else:
                want[j - n] -= c * v
    want = [mpi_smod(v, q) for v in want]
    dim = (q.bit_length() + logn) // 59 + 1                  # src/he-kem.c:83
    W = q.bit_length() // 64 + 1
    r = torch.empty(W * n, dtype=torch.int64, device="cuda")
    g.poly_mul_general(r, to_device(ints_to_big(a, W)), to_device(ints_to_big(s, W)), W, dim, q)
    assert big_to_ints(to_host(r), W, n)[0] == want


@pytest.mark.parametrize("logn,W,batch", [(6, 1, 1), (7, 14, 3), (13, 7, 2), (16, 14, 2), (9, 33, 1)])
def test_big_transpose_between_word_major_and_rows(engine_ctx, logn, W, batch):
    """gpq_big_transpose: the kernels' layout (word j of coefficient i at j*n + i) <-> rows of W words per coefficient, the layout the
    MPI-typed calls stage host data through."""
    import ctypes as C
    import torch
    g = engine_ctx(logn, 2)
    n = g.n
    rng = np.random.default_rng(logn * 100 + W)
    words = rng.integers(0, 1 << 63, size=(batch, W, n), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(batch, W, n), dtype=np.uint64)
    d_words = to_device(words.reshape(-1))
    d_rows = torch.empty_like(d_words)
    lib = g.lib
    assert lib.gpq_big_transpose(g.h, C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_words.data_ptr()), W, batch, 1, g._stream()) == 0
    assert np.array_equal(to_host(d_rows).reshape(batch, n, W), words.transpose(0, 2, 1))
    back = torch.empty_like(d_words)
    assert lib.gpq_big_transpose(g.h, C.c_void_p(back.data_ptr()), C.c_void_p(d_rows.data_ptr()), W, batch, 0, g._stream()) == 0
    assert torch.equal(back, d_words)


@pytest.mark.parametrize("logn,W,polys", [(6, 1, 1), (8, 2, 3), (13, 7, 2), (16, 14, 2)])
def test_big_addsub_is_wrapping_twos_complement_arithmetic(engine_ctx, logn, W, polys):
    """gpq_big_addsub: a + b, a - b, -a on W-word two's complement slabs (the arithmetic of src/he-add.c:32-140 before its mpi_smod), with the
    carries that run through every word: all-ones + 1, 0 - 1, the most negative value, aliasing of out with either operand."""
    import torch
    g = engine_ctx(logn, 2)
    n = g.n
    rng = random.Random(logn * 31 + W)
    mod = 1 << (64 * W)

    def ints(seed_edges):
        v = [rng.randrange(mod) for _ in range(polys * n)]
        v[:6] = seed_edges
        return v

    a = ints([mod - 1, 0, 1 << (64 * W - 1), (1 << (64 * W - 1)) - 1, 1, mod - 1])
    b = ints([1, 1, 1 << (64 * W - 1), 1, mod - 1, mod - 1])

    def slab(v):        # word-major per polynomial
        out = np.empty((polys, W, n), dtype=np.uint64)
        for k in range(polys):
            for j in range(W):
                out[k, j] = [(v[k * n + i] >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]
        return to_device(out.reshape(-1))

    def back(t):
        w = to_host(t).reshape(polys, W, n)
        return [sum(int(w[k, j, i]) << (64 * j) for j in range(W)) for k in range(polys) for i in range(n)]

    da, db = slab(a), slab(b)
    out = torch.empty_like(da)
    assert back(g.big_addsub(out, da, db, W, 0)) == [(x + y) % mod for x, y in zip(a, b)]
    assert back(g.big_addsub(out, da, db, W, 1)) == [(x - y) % mod for x, y in zip(a, b)]
    assert back(g.big_addsub(out, da, None, W, 2)) == [(-x) % mod for x in a]
    t = da.clone()
    assert back(g.big_addsub(t, t, db, W, 1)) == [(x - y) % mod for x, y in zip(a, b)]       # out aliases a
    t = db.clone()
    assert back(g.big_addsub(t, da, t, W, 0)) == [(x + y) % mod for x, y in zip(a, b)]       # out aliases b

