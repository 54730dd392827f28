"""gpq_he_mul_rs (round 5): he_mul followed by he_rs (src/he-mult.c:88-156, then src/he-rescale.c:33-54 -- BASELINE configs[2] is "he_mul +
he_rescale") with the rounding division riding in the relinearisation tail.  The contract is word-for-word equality with the two calls
gpq_he_mul + gpq_he_rs (which the dense tests pin to the restated reference): on every shape of the streaming-bridge suite, with coefficients
FORCED through the exact kernels behind the streaming tail (they are written unrescaled by those and finished by the masked rescale), across
launch groups and lanes, with the streaming bridge off (two calls inside), and for a Delta of more than one word (two calls inside)."""
import pytest

from tests.test_stream_bridge_gpu import SHAPES, _centred

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _two_calls(g, torch, cts, rlk, W, logql, dims, logDelta):
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    g.he_mul(o0, o1, *cts, rlk[0], rlk[1], W, logql, *dims)
    g.he_rs(o0, o1, W, logDelta, logql - logDelta)
    return o0, o1


def _fused(g, torch, cts, rlk, W, logql, dims, logDelta):
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    g.he_mul_rs(o0, o1, *cts, rlk[0], rlk[1], W, logql, *dims, logDelta)
    return o0, o1


@pytest.mark.parametrize("logn,logqL,logql,batch", SHAPES)
def test_he_mul_rs_equals_he_mul_then_he_rs(engine_ctx, logn, logqL, logql, batch):
    torch = _torch()
    dimP, dimA, dimB, dimevk = engine_ctx(logn, 20).he_dims(logqL, logql)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logqL + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5100 + logn + logql)
    cts = [_centred(torch, gen, batch, W, n, logql) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    dims = (dimA, dimB, dimP)
    try:
        for logDelta in (50, 1, 63, 17):
            want = _two_calls(g, torch, cts, rlk, W, logql, dims, logDelta)
            got = _fused(g, torch, cts, rlk, W, logql, dims, logDelta)
            for name, a, b in zip(("c0", "c1"), want, got):
                assert torch.equal(a, b), "%s, Delta = 2^%d" % (name, logDelta)
        want = _two_calls(g, torch, cts, rlk, W, logql, dims, 50)
        for every in (5, 64, 1):                      # flagged coefficients: exact kernels, then the masked rescale
            g.debug_force_redo(every)
            got = _fused(g, torch, cts, rlk, W, logql, dims, 50)
            for name, a, b in zip(("c0", "c1"), want, got):
                assert torch.equal(a, b), "%s with every %d-th coefficient through the exact kernels" % (name, every)
        g.debug_force_redo(0)
        if logql > 80:                                # a Delta wider than a word: the two calls inside
            want = _two_calls(g, torch, cts, rlk, W, logql, dims, 70)
            got = _fused(g, torch, cts, rlk, W, logql, dims, 70)
            assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
        g.set_stream_bridge(False)                    # round 3's kernels: no tail to ride in, the two calls inside
        got = _fused(g, torch, cts, rlk, W, logql, dims, 50)
        want = _two_calls(g, torch, cts, rlk, W, logql, dims, 50)
        assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
        assert bool((want[0] != 0).any())
    finally:
        g.debug_force_redo(0)
        g.set_stream_bridge(True)


@pytest.mark.parametrize("lanes", [0, 1])
def test_he_mul_rs_across_launch_groups_and_lanes(engine_ctx, lanes):
    torch = _torch()
    logn, logq, batch = 13, 438, 7
    dimP, dimA, dimB, dimevk = engine_ctx(logn, 20).he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(78)
    cts = [_centred(torch, gen, batch, W, n, logq) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    dims = (dimA, dimB, dimP)
    try:
        g.set_chunk(3)
        g.set_overlap(lanes)
        want = _two_calls(g, torch, cts, rlk, W, logq, dims, 50)
        got = _fused(g, torch, cts, rlk, W, logq, dims, 50)
        assert g.last_lanes() == 1 + lanes
        g.debug_force_redo(9)
        forced = _fused(g, torch, cts, rlk, W, logq, dims, 50)
        plain = _two_calls(g, torch, cts, rlk, W, logq, dims, 50)       # a plain he_mul after a fused one: the rescale must not stick to the context
    finally:
        g.debug_force_redo(0)
        g.set_chunk(32)
        g.set_overlap(-1)
    for a, b, c, d in zip(want, got, forced, plain):
        assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)


def test_he_mul_rs_rejects_a_delta_that_is_not_inside_the_modulus(engine_ctx):
    torch = _torch()
    from gpqhe_amd import GpqError
    g = engine_ctx(8, 6)
    t = torch.zeros(2 * g.n, dtype=torch.int64, device="cuda")
    k = torch.zeros(6 * g.n, dtype=torch.int64, device="cuda")
    for bad in (0, 109, 200):
        with pytest.raises(GpqError):
            g.he_mul_rs(t.clone(), t.clone(), t, t, t, t, k, k, 2, 109, 4, 6, 2, bad)


@pytest.mark.parametrize("logn,logq", [(16, 850), (14, 438)])
def test_the_rescale_really_rides_in_the_tail_at_the_benchmarked_shapes(engine_ctx, logn, logq):
    """Equality with the two calls cannot tell a fused call from two calls inside: the per-kernel profile can.  At the headline shape and at the
    reference's default shape gpq_he_mul_rs launches NO rescale kernel; gpq_he_mul + gpq_he_rs launches two (c0, c1)."""
    torch = _torch()
    dimP, dimA, dimB, dimevk = engine_ctx(logn, 20).he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    cts = [_centred(torch, gen, 2, W, n, logq) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    dims = (dimA, dimB, dimP)
    _fused(g, torch, cts, rlk, W, logq, dims, 50)     # (first call at the shape: tables)
    torch.cuda.synchronize()
    try:
        g.profile(True)
        _fused(g, torch, cts, rlk, W, logq, dims, 50)
        torch.cuda.synchronize()
        fused = g.profile_collect()
        _two_calls(g, torch, cts, rlk, W, logq, dims, 50)
        torch.cuda.synchronize()
        two = g.profile_collect()
    finally:
        g.profile(False)
    assert "bridge_rescale" not in fused and "bridge_tail_stream" in fused, fused
    assert two["bridge_rescale"][1] == 2, two
