"""Whole he_mul (+ he_rs) and he_swk at the LAUNCH SHAPES bench.py quotes, against the restated reference.

bench.py's `he_mul_mpi_level` runs batch 64 as two launch groups of 32 ciphertexts, the second group on the peer lane (gpq_set_overlap,
default on): persistent streaming kernels whose rings are refilled across products and groups, per-wave flag words, two contexts.  The
dense full-size tests (test_dense_full_size_gpu.py) run batch 1.  Here the batch is a block of 4 DISTINCT dense ciphertext pairs
repeated 16 times:

* every repeat must equal its first occurrence word for word (whatever slot of whatever group and lane it ran in), lanes on and lanes off;
* the four distinct products -- which sit at slots 0..3, and therefore at the first (0 = A, 32 = A) and last (31 = D, 63 = D) slot of EACH
  group -- equal oracle/bigint_ref.he_mul (src/he-mult.c:88-156 restated), all coefficients; he_rs behind it equals src/he-rescale.c:33-54.

The same for the reference's default shape (n = 2^14, q = 2^438, batch 64: tests/gpqhe.c:1296-1299) and for he_swk
(src/he-automorphism.c:40-85) at BASELINE configs[4]'s shape (n = 2^17, 44 limbs, batch 64)."""
import numpy as np
import pytest
import torch

from gpqhe_amd import to_device, to_host
from oracle import bigint_ref as ref
from oracle import expect

pytestmark = pytest.mark.gpu
BATCH, DISTINCT = 64, 4


def _dense(rng, W, n, logq):
    """one dense centred polynomial as words: uniform in [-q/2, q/2) with the extremes of the range riding along"""
    w = rng.integers(0, 1 << 63, size=(W, n), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(W, n), dtype=np.uint64)
    top = logq - 1 - 64 * (W - 1)
    w[W - 1] = rng.integers(-(1 << top), 1 << top, size=n, dtype=np.int64).view(np.uint64)
    q = 1 << logq
    w[:, :4] = expect.ints_to_words([-(q >> 1), (q >> 1) - 1, 0, -1], W).reshape(W, 4)
    return w.reshape(-1)


def _tile(blocks):
    """[DISTINCT polynomials as words] -> the batch slab: the block repeated BATCH / DISTINCT times"""
    return np.concatenate([blocks[k % DISTINCT] for k in range(BATCH)])


def _same_as_first_occurrence(t, per):
    v = t.view(BATCH // DISTINCT, DISTINCT * per)
    return bool((v == v[0]).all().item())


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("logn,logq", [(16, 850), (14, 438)])
def test_he_mul_and_he_rs_at_the_benchmarked_launch_shape(engine_ctx, oracle_ctx, logn, logq):
    n, W = 1 << logn, logq // 64 + 1
    dimP, dimA, dimB, dimevk = engine_ctx(logn, 20).he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    rng = np.random.default_rng(logn * 1000 + logq)
    blocks = [[_dense(rng, W, n, logq) for _ in range(DISTINCT)] for _ in range(4)]           # ct1.c0, ct1.c1, ct2.c0, ct2.c1
    rlk0, rlk1 = o.gen(3000, dimevk)[: dimB * n], o.gen(3001, dimevk)[: dimB * n]
    dev = [to_device(_tile(b)) for b in blocks]
    k0, k1 = to_device(rlk0), to_device(rlk1)
    per = W * n
    outs = {}
    try:
        for lanes in (1, 0):
            g.set_overlap(lanes)
            o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
            g.he_mul(o0, o1, *dev, k0, k1, W, logq, dimA, dimB, dimP)
            r0, r1 = o0.clone(), o1.clone()
            g.he_rs(r0, r1, W, 50, logq - 50)
            torch.cuda.synchronize()
            for name, t in (("c0", o0), ("c1", o1), ("rs0", r0), ("rs1", r1)):
                assert _same_as_first_occurrence(t, per), "lanes=%d: a repeat of %s differs from its first occurrence" % (lanes, name)
            outs[lanes] = (o0, o1, r0, r1)
        for a, b in zip(outs[1], outs[0]):
            assert torch.equal(a, b), "two lanes and one lane differ"
    finally:
        g.set_overlap(1)
    tasks = [dict(kind="he_mul", logn=logn, dimP=dimP, dimA=dimA, dimB=dimB, W=W, logq=logq, ct=[blocks[s][j] for s in range(4)], rlk0=rlk0, rlk1=rlk1, rs=50)
             for j in range(DISTINCT)]
    want = expect.expect_many(tasks, workers=4)
    got = [to_host(t[: DISTINCT * per]) for t in outs[1]]
    for j in range(DISTINCT):
        for name, a in zip(("c0", "c1", "rs0", "rs1"), got):
            bad = np.flatnonzero(a[j * per:(j + 1) * per] != want[j][name])
            assert bad.size == 0, "ciphertext %d %s: %d words differ from the restated reference, first %s" % (j, name, bad.size, bad[:4])
    # not degenerate: the products are spread over the centred range
    assert len(set(to_host(outs[1][0][:n]).tolist())) > n // 2


@pytest.mark.timeout(1500)
def test_he_swk_at_configs4_shape_batch_64(engine_ctx, oracle_ctx):
    logn, logq = 17, 835
    n, W = 1 << logn, logq // 64 + 1
    dimP = (logq + 1 + logn) // 59 + 1                                          # hectx.dim, src/precomp.c:401
    P = ref.RnsBasis(engine_ctx(logn, dimP).p[:dimP]).P
    dimB = (logq + 1 + (P << logq).bit_length() + logn) // 59 + 1               # src/he-automorphism.c:52
    assert dimB == 44
    g, o = engine_ctx(logn, dimB), oracle_ctx(logn, dimB)
    rng = np.random.default_rng(17 * 835 + 64)
    blocks = [[_dense(rng, W, n, logq) for _ in range(DISTINCT)] for _ in range(2)]           # d0, d1
    swk0, swk1 = o.gen(5002, dimB), o.gen(5003, dimB)
    dev = [to_device(_tile(b)) for b in blocks]
    k0, k1 = to_device(swk0), to_device(swk1)
    per = W * n
    outs = {}
    try:
        for lanes in (1, 0):
            g.set_overlap(lanes)
            o0, o1 = torch.empty_like(dev[0]), torch.empty_like(dev[0])
            g.he_swk(o0, o1, dev[0], dev[1], k0, k1, W, logq, dimB, dimP)
            torch.cuda.synchronize()
            for name, t in (("c0", o0), ("c1", o1)):
                assert _same_as_first_occurrence(t, per), "lanes=%d: a repeat of %s differs from its first occurrence" % (lanes, name)
            outs[lanes] = (o0, o1)
        for a, b in zip(outs[1], outs[0]):
            assert torch.equal(a, b), "two lanes and one lane differ"
    finally:
        g.set_overlap(1)
    tasks = [dict(kind="he_swk", logn=logn, dimP=dimP, dimB=dimB, W=W, logq=logq, d0=blocks[0][j], d1=blocks[1][j], swk0=swk0, swk1=swk1) for j in range(DISTINCT)]
    want = expect.expect_many(tasks, workers=4)
    got = [to_host(t[: DISTINCT * per]) for t in outs[1]]
    for j in range(DISTINCT):
        for name, a in zip(("c0", "c1"), got):
            bad = np.flatnonzero(a[j * per:(j + 1) * per] != want[j][name])
            assert bad.size == 0, "polynomial pair %d %s: %d words differ from the restated reference, first %s" % (j, name, bad.size, bad[:4])
