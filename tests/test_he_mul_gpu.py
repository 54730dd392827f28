"""Parity of the fused he_mul RNS core (tensor stage + key-switch inner product)
with the oracle's replay of src/he-mult.c:116-138 / :58-66 and with the golden
digests the compiled reference produced."""
import numpy as np
import pytest

from gpqhe_amd import to_device, to_host
from oracle.oracle import fnv

pytestmark = pytest.mark.gpu


def _empty_like(t):
    import torch
    return torch.empty_like(t)


def _tensor(g, ins, dim):
    dev = [to_device(x) for x in ins]
    outs = [_empty_like(dev[0]) for _ in range(3)]
    g.he_mul_tensor(outs[0], outs[1], outs[2], dev[0], dev[1], dev[2], dev[3], dim)
    for d, x in zip(dev, ins):
        assert np.array_equal(to_host(d), x), "inputs must be preserved"
    return [to_host(t) for t in outs]


def _keyswitch(g, x, e0, e1, dim):
    dx, d0, d1 = to_device(x), to_device(e0), to_device(e1)
    c0, c1 = _empty_like(dx), _empty_like(dx)
    g.he_keyswitch(c0, c1, dx, d0, d1, dim)
    assert np.array_equal(to_host(dx), x)
    return to_host(c0), to_host(c1)


@pytest.mark.parametrize("logn", ["7", "12", "15", "16"])
def test_golden_he_mul_core(golden, engine_ctx, oracle_ctx, logn):
    kat = golden["he_mul_core_kat"][logn]
    seeds = golden["he_mul_core_kat"]["_seeds"]
    dA, dB = kat["dA"], kat["dB"]
    npr = max(dA, dB)
    o, g = oracle_ctx(int(logn), npr), engine_ctx(int(logn), npr)
    ins = [o.gen(seeds[k], dA) for k in ("a0", "a1", "b0", "b1")]
    assert [fnv(x) for x in ins] == kat["inputs"]
    d0, d1, d2 = _tensor(g, ins, dA)
    assert (fnv(d0), fnv(d1), fnv(d2)) == (kat["d0"], kat["d1"], kat["d2"])
    c0, c1 = _keyswitch(g, o.gen(seeds["d2"], dB), o.gen(seeds["evk0"], dB), o.gen(seeds["evk1"], dB), dB)
    assert (fnv(c0), fnv(c1)) == (kat["c0"], kat["c1"])


@pytest.mark.parametrize("logn,dim,batch,chunk", [(7, 5, 3, 4), (12, 2, 2, 4), (13, 2, 5, 2), (14, 3, 2, 4), (16, 2, 3, 2), (17, 2, 1, 4),
                                                  (17, 44, 1, 4)])          # BASELINE configs[4]: every limb of the n = 2^17 shape
def test_batched_core_matches_oracle(engine_ctx, oracle_ctx, logn, dim, batch, chunk):
    """Batches (including a batch that is not a multiple of the launch chunk) against the oracle."""
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    g.set_chunk(chunk)
    per = dim * o.n
    ins = [np.concatenate([o.gen(100 * s + k, dim) for k in range(batch)]) for s in range(4)]
    d0, d1, d2 = _tensor(g, ins, dim)
    for k in range(batch):
        e0, e1, e2 = o.he_mul_tensor(*[x[k * per:(k + 1) * per].copy() for x in ins], dim)
        assert np.array_equal(d0[k * per:(k + 1) * per], e0)
        assert np.array_equal(d1[k * per:(k + 1) * per], e1)
        assert np.array_equal(d2[k * per:(k + 1) * per], e2)
    x = np.concatenate([o.gen(900 + k, dim) for k in range(batch)])
    ev0, ev1 = o.gen(7000, dim), o.gen(7001, dim)
    c0, c1 = _keyswitch(g, x, ev0, ev1, dim)
    for k in range(batch):
        f0, f1 = o.keyswitch(x[k * per:(k + 1) * per].copy(), ev0, ev1, dim)
        assert np.array_equal(c0[k * per:(k + 1) * per], f0)
        assert np.array_equal(c1[k * per:(k + 1) * per], f1)
    g.set_chunk(4)


@pytest.mark.parametrize("logn,dim,batch,chunk,limb_block", [(13, 7, 5, 2, 3), (16, 6, 3, 2, 4), (17, 44, 2, 1, 7), (17, 44, 1, 4, 1)])
def test_limb_blocked_launch_groups_match_oracle(oracle_ctx, logn, dim, batch, chunk, limb_block):
    """gpq_set_limb_block: the three kernels of the tensor stage / key switch run per block of limbs inside a group of polynomials.
    Blocks that do not divide the limb count, that are cut again by the butterfly-class boundaries of the chain (n = 2^17: wide /
    split / plain limbs inside 44) and a ragged last group of polynomials: bit-exact against the oracle for every setting."""
    import gpqhe_amd
    o = oracle_ctx(logn, dim)
    g = gpqhe_amd.PolyContext(logn, dim)          # own context: the session-wide ones keep their defaults
    g.set_chunk(chunk)
    g.set_limb_block(limb_block)
    per = dim * o.n
    ins = [np.concatenate([o.gen(300 * s + k, dim) for k in range(batch)]) for s in range(4)]
    d0, d1, d2 = _tensor(g, ins, dim)
    x = np.concatenate([o.gen(950 + k, dim) for k in range(batch)])
    ev0, ev1 = o.gen(7100, dim), o.gen(7101, dim)
    c0, c1 = _keyswitch(g, x, ev0, ev1, dim)
    for k in range(batch):
        sl = slice(k * per, (k + 1) * per)
        e = o.he_mul_tensor(*[v[sl].copy() for v in ins], dim)
        f = o.keyswitch(x[sl].copy(), ev0, ev1, dim)
        for got, exp in zip((d0, d1, d2, c0, c1), list(e) + list(f)):
            assert np.array_equal(got[sl], exp)
    g.close()


@pytest.mark.parametrize("logn,dim,batch", [(7, 5, 1), (13, 2, 3), (14, 3, 2), (16, 4, 2), (17, 33, 1)])
def test_poly_mul_limb_loop(engine_ctx, oracle_ctx, logn, dim, batch):
    """src/poly.c:96-103 without rns_decompose (small rings: four transforms; n >= 2^13: strided pass, fused middle, strided pass;
    (17, 33) crosses the three twiddle classes), also with the result written over an operand."""
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    a = np.concatenate([o.gen(41 + 2 * k, dim) for k in range(batch)])
    b = np.concatenate([o.gen(42 + 2 * k, dim) for k in range(batch)])
    per = dim << logn
    want = np.concatenate([o.poly_mul_rns(a[k * per:(k + 1) * per], b[k * per:(k + 1) * per], dim) for k in range(batch)])
    da, db = to_device(a), to_device(b)
    r = _empty_like(da)
    g.poly_mul_rns(r, da, db, dim)
    assert np.array_equal(to_host(r), want)
    da, db = to_device(a), to_device(b)
    g.poly_mul_rns(db, da, db, dim)
    assert np.array_equal(to_host(db), want)


@pytest.mark.parametrize("logn,dim,batch", [(7, 5, 2), (13, 2, 3), (16, 4, 1), (17, 33, 1)])
def test_mulpt_limb_loop(engine_ctx, oracle_ctx, logn, dim, batch):
    """src/he-mult.c:179-185: both ciphertext polynomials times the plaintext polynomial, in place on the operands."""
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    per = dim << logn
    m, x0, x1 = (np.concatenate([o.gen(s + 3 * k, dim) for k in range(batch)]) for s in (51, 52, 53))
    want0 = np.concatenate([o.poly_mul_rns(m[k * per:(k + 1) * per], x0[k * per:(k + 1) * per], dim) for k in range(batch)])
    want1 = np.concatenate([o.poly_mul_rns(m[k * per:(k + 1) * per], x1[k * per:(k + 1) * per], dim) for k in range(batch)])
    dm, d0, d1 = to_device(m), to_device(x0), to_device(x1)
    g.mulpt_rns(d0, d1, dm, d0, d1, dim)
    assert np.array_equal(to_host(d0), want0) and np.array_equal(to_host(d1), want1)


def test_linearity_full_size(engine_ctx, oracle_ctx):
    """Size-independent property at BASELINE's full shape (n=2^16, 30 limbs):
    the tensor stage is bilinear, so tensor(a0+x, a1, b0, b1).d0 == d0 + x*b0."""
    logn, dim = 16, 30
    o, g = oracle_ctx(logn, 45), engine_ctx(logn, 45)
    ins = [o.gen(1000 + s, dim) for s in range(4)]
    x = o.gen(5, dim)
    p = np.repeat(np.array(o.p[:dim], dtype=np.uint64), o.n)
    a0x = (ins[0] + x) % p  # both < p < 2^60: no uint64 overflow
    d0, d1, d2 = _tensor(g, ins, dim)
    e0, e1, e2 = _tensor(g, [a0x, ins[1], ins[2], ins[3]], dim)
    z = np.zeros_like(x)
    f0, f1, f2 = _tensor(g, [x, z, ins[2], ins[3]], dim)
    assert np.array_equal(e0, (d0 + f0) % p)
    assert np.array_equal(e1, (d1 + f1) % p)
    assert np.array_equal(e2, d2) and not f2.any()


def test_repeatable_under_load(engine_ctx, oracle_ctx):
    """The contiguous kernels exchange registers through wave-private LDS regions without workgroup barriers,
    and the strided ones through one barrier: run the full-size core back to back (every CU busy, several
    launch groups in flight) and require identical bits every time; spot-check one ciphertext against the oracle."""
    import torch
    logn, dA, dB, batch = 16, 30, 45, 24
    g, o = engine_ctx(logn, 45), oracle_ctx(logn, 45)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)

    def slab(dim, polys):
        out = torch.empty((polys, dim, g.n), dtype=torch.int64, device="cuda")
        for d in range(dim):
            out[:, d, :] = torch.randint(0, g.p[d], (polys, g.n), dtype=torch.int64, device="cuda", generator=gen)
        return out.reshape(-1)

    ins = [slab(dA, batch) for _ in range(4)]
    x, e0, e1 = slab(dB, batch), slab(dB, 1), slab(dB, 1)
    outs = [torch.empty_like(ins[0]) for _ in range(3)]
    cs = [torch.empty_like(x) for _ in range(2)]
    wsA, wsB = g.tensor_workspace(dA, batch), g.keyswitch_workspace(dB, batch)
    g.set_chunk(8)                      # three launch groups per call
    g.he_mul_tensor(*outs, *ins, dA, wsA)
    g.he_keyswitch(*cs, x, e0, e1, dB, wsB)
    first = [t.clone() for t in outs + cs]
    for _ in range(12):
        g.he_mul_tensor(*outs, *ins, dA, wsA)
        g.he_keyswitch(*cs, x, e0, e1, dB, wsB)
        for got, ref0 in zip(outs + cs, first):
            assert torch.equal(got, ref0), "results changed between identical launches"
    g.set_chunk(16)
    k, perA, perB = batch - 1, dA * g.n, dB * g.n
    d = o.he_mul_tensor(*[to_host(v[k * perA:(k + 1) * perA]) for v in ins], dA)
    c = o.keyswitch(to_host(x[k * perB:(k + 1) * perB]), to_host(e0), to_host(e1), dB)
    for got, exp, per in zip(first, list(d) + list(c), (perA, perA, perA, perB, perB)):
        assert np.array_equal(to_host(got[k * per:(k + 1) * per]), exp)


@pytest.mark.timeout(600)
def test_configs2_full_batch_is_consistent_with_single_ciphertext_runs(engine_ctx, oracle_ctx):
    """BASELINE configs[2] at its full size inside the suite: n = 2^16, 30 / 45 limbs, batch 64 (two launch groups of 32).  Every
    ciphertext of the batch must come out exactly as when it is run alone (batch 1: other grid, other launch-group boundaries), the
    XOR of all outputs is a checksum of checksums across the two runs, and ciphertexts 0 and 63 equal the oracle."""
    import torch
    import gpqhe_amd
    logn, dA, dB, B = 16, 30, 45, 64
    g, o = engine_ctx(logn, dB), oracle_ctx(logn, dB)
    n = g.n
    perA, perB = dA * n, dB * n
    gen = torch.Generator(device="cuda")
    gen.manual_seed(6464)

    def slab(dim, batch):
        s = torch.empty((batch, dim, n), dtype=torch.int64, device="cuda")
        for d in range(dim):
            s[:, d, :] = torch.randint(0, g.p[d], (batch, n), dtype=torch.int64, device="cuda", generator=gen)
        return s.reshape(-1)

    ins = [slab(dA, B) for _ in range(4)]
    x = slab(dB, B)
    e0, e1 = slab(dB, 1), slab(dB, 1)
    d = [torch.empty_like(ins[0]) for _ in range(3)]
    c = [torch.empty_like(x) for _ in range(2)]
    g.he_mul_tensor(d[0], d[1], d[2], *ins, dA)
    g.he_keyswitch(c[0], c[1], x, e0, e1, dB)
    one_d = [torch.empty(perA, dtype=torch.int64, device="cuda") for _ in range(3)]
    one_c = [torch.empty(perB, dtype=torch.int64, device="cuda") for _ in range(2)]
    acc_batch = torch.zeros(perB, dtype=torch.int64, device="cuda")
    acc_single = torch.zeros(perB, dtype=torch.int64, device="cuda")
    for k in range(B):
        sa, sb = slice(k * perA, (k + 1) * perA), slice(k * perB, (k + 1) * perB)
        g.he_mul_tensor(one_d[0], one_d[1], one_d[2], *[v[sa].contiguous() for v in ins], dA)
        g.he_keyswitch(one_c[0], one_c[1], x[sb].contiguous(), e0, e1, dB)
        for i in range(3):
            assert torch.equal(d[i][sa], one_d[i]), (k, i)
            acc_batch[:perA] ^= d[i][sa]
            acc_single[:perA] ^= one_d[i]
        for i in range(2):
            assert torch.equal(c[i][sb], one_c[i]), (k, i)
            acc_batch ^= c[i][sb]
            acc_single ^= one_c[i]
    assert torch.equal(acc_batch, acc_single) and bool((acc_batch != 0).any())
    for k in (0, B - 1):
        sa, sb = slice(k * perA, (k + 1) * perA), slice(k * perB, (k + 1) * perB)
        exp = list(o.he_mul_tensor(*[gpqhe_amd.to_host(v[sa]) for v in ins], dA)) + list(o.keyswitch(gpqhe_amd.to_host(x[sb]), gpqhe_amd.to_host(e0), gpqhe_amd.to_host(e1), dB))
        for got, want, sl in zip(d + c, exp, (sa, sa, sa, sb, sb)):
            assert np.array_equal(gpqhe_amd.to_host(got[sl]), want)
    del ins, x, d, c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("logn,dim,batch", [(7, 5, 2), (13, 3, 3), (16, 30, 2), (17, 40, 1)])
def test_squaring_takes_the_two_transform_path_and_gives_the_same_residues(engine_ctx, oracle_ctx, logn, dim, batch):
    """he_mul(&ct, &ct, &ct, rlk) (src/he-algo.c:151; the repeated squarings of he_exp / he_inv): gpq_he_mul_tensor with b0 = a0, b1 = a1
    runs tensor_sq_mid8 (two forward transforms).  Bit-identical with the general kernel fed copies of the operands and with the oracle;
    all-(p-1) limbs drive the doubled cross term to its bound."""
    import torch
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    per = dim * o.n
    a0 = np.concatenate([o.gen(77 + k, dim) for k in range(batch)])
    a1 = np.concatenate([o.gen(177 + k, dim) for k in range(batch)])
    for d in range(dim):                                   # extremes in the last ciphertext
        a0[(batch - 1) * per + d * o.n:(batch - 1) * per + (d + 1) * o.n] = o.p[d] - 1
        a1[(batch - 1) * per + d * o.n:(batch - 1) * per + (d + 1) * o.n:2] = o.p[d] - 1
    da0, da1 = to_device(a0), to_device(a1)
    sq = [torch.empty_like(da0) for _ in range(3)]
    g.he_mul_tensor(sq[0], sq[1], sq[2], da0, da1, da0, da1, dim)                 # aliased operands: the squaring kernel
    gen = [torch.empty_like(da0) for _ in range(3)]
    g.he_mul_tensor(gen[0], gen[1], gen[2], da0, da1, da0.clone(), da1.clone(), dim)   # distinct buffers: the general kernel
    for s, t in zip(sq, gen):
        assert torch.equal(s, t)
    for k in range(batch):
        sl = slice(k * per, (k + 1) * per)
        exp = o.he_mul_tensor(a0[sl].copy(), a1[sl].copy(), a0[sl].copy(), a1[sl].copy(), dim)
        for s, e in zip(sq, exp):
            assert np.array_equal(to_host(s[sl]), e)
