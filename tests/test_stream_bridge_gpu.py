"""bridge_stream.hpp (round 4): the fused streaming bridge kernels of gpq_he_mul / gpq_he_swk -- CRT(d2hat) -> rns_decompose in one kernel
(src/he-mult.c:140 feeding :59) and the one-product relinearisation tail that makes its addend d0 / d1 from limbs on the spot (:139, :141,
:67-77) -- against round 3's separate kernels (gpq_set_stream_bridge(ctx, 0)), which the dense full-size tests pin to the oracle:
identical words on dense random ciphertexts, at the headline shape, at the reference's default shape, at lower levels, for squarings,
for he_swk with and without an addend, across launch groups; and with coefficients FORCED through the exact kernels behind the streaming
ones (gpq_debug_force_redo: every k-th coefficient is flagged as if it sat in a rounding window), which must not change a word."""
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _centred(torch, gen, batch, W, n, logq):
    """dense random values in [-2^(logq-2), 2^(logq-2)) as W two's-complement words (the words above the value's top word: its sign)"""
    big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
    wt = (logq - 2) // 64
    top = min(logq - 2 - 64 * wt, 62)            # (torch.randint's bounds are int64)
    big[:, wt] = torch.randint(-(1 << top), 1 << top, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
    for j in range(wt + 1, W):
        big[:, j] = big[:, wt] >> 63
    return big.reshape(-1).contiguous()


def _run(g, torch, cts, rlk, W, logql, dimA, dimB, dimP, square=False):
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    if square:
        g.he_mul(o0, o1, cts[0], cts[1], cts[0], cts[1], rlk[0], rlk[1], W, logql, dimA, dimB, dimP)
    else:
        g.he_mul(o0, o1, *cts, rlk[0], rlk[1], W, logql, dimA, dimB, dimP)
    s0, s1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    g.he_swk(s0, s1, cts[0], cts[1], rlk[0], rlk[1], W, logql, dimB, dimP)
    torch.cuda.synchronize()
    return o0, o1, s0, s1


# (logn, logqL, logql, batch): the headline shape (30 / 45 limbs, 14 words), the reference's default (tests/gpqhe.c:1296-1299: 16 / 24 limbs,
# 7 words), a lower level of each (fewer limbs than the instantiation's k steps: zero-padded constant matrices), small rings
SHAPES = [(16, 850, 850, 2), (14, 438, 438, 3), (16, 850, 500, 2), (14, 438, 238, 2), (13, 300, 300, 3), (8, 109, 109, 5), (10, 200, 130, 4)]


@pytest.mark.parametrize("logn,logqL,logql,batch", SHAPES)
def test_streaming_bridge_equals_the_separate_kernels(engine_ctx, logn, logqL, logql, batch):
    torch = _torch()
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logqL + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4100 + logn + logql)
    cts = [_centred(torch, gen, batch, W, n, logql) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    try:
        g.set_stream_bridge(False)
        g.set_lazy_decompose(False)                   # round 3's path: separate kernels, canonical residues between them
        want = _run(g, torch, cts, rlk, W, logql, dimA, dimB, dimP)
        want_sq = _run(g, torch, cts, rlk, W, logql, dimA, dimB, dimP, square=True)
        g.set_stream_bridge(True)
        g.set_lazy_decompose(True)
        got = _run(g, torch, cts, rlk, W, logql, dimA, dimB, dimP)
        got_sq = _run(g, torch, cts, rlk, W, logql, dimA, dimB, dimP, square=True)
        forced = []
        for every in (5, 64, 1):                      # scattered coefficients, one per group of 64, every coefficient
            g.debug_force_redo(every)
            forced.append(_run(g, torch, cts, rlk, W, logql, dimA, dimB, dimP))
    finally:
        g.debug_force_redo(0)
        g.set_stream_bridge(True)
        g.set_lazy_decompose(True)
    names = ("he_mul c0", "he_mul c1", "he_swk c0", "he_swk c1")
    for name, a, b in zip(names, want, got):
        assert torch.equal(a, b), name
    for name, a, b in zip(names, want_sq, got_sq):
        assert torch.equal(a, b), name + " (squaring)"
    for every, res in zip((5, 64, 1), forced):
        for name, a, b in zip(names, want, res):
            assert torch.equal(a, b), "%s with every %d-th coefficient through the exact kernels" % (name, every)
    assert bool((want[0] != 0).any()) and bool((want[3] != 0).any())


def test_streaming_bridge_across_launch_groups(engine_ctx):
    """more ciphertexts than one launch group (gpq_set_chunk): the per-wave flag words and the scratch of a group are reused by the next"""
    torch = _torch()
    logn, logq, batch = 13, 438, 7
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    cts = [_centred(torch, gen, batch, W, n, logq) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    try:
        g.set_chunk(3)
        g.set_stream_bridge(False)
        g.set_lazy_decompose(False)
        want = _run(g, torch, cts, rlk, W, logq, dimA, dimB, dimP)
        g.set_stream_bridge(True)
        g.set_lazy_decompose(True)
        got = _run(g, torch, cts, rlk, W, logq, dimA, dimB, dimP)
        g.debug_force_redo(9)
        forced = _run(g, torch, cts, rlk, W, logq, dimA, dimB, dimP)
    finally:
        g.debug_force_redo(0)
        g.set_stream_bridge(True)
        g.set_lazy_decompose(True)
        g.set_chunk(32)
    for a, b, c in zip(want, got, forced):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_flag_bytes_grow_inside_a_tail_that_cannot_stream():
    """dimA = 3 (q_l = 2^62 under q_L = 2^627): the streaming kernels decline (fewer than 4 limbs), the addend still arrives as limbs, and the
    relinearisation tail makes it with the exact-CRT path -- which may be the call that GROWS the context's flag bytes.  Every argument block
    of that tail must see the grown buffer (found by tools/soak_bridge.py: a block built before the growth kept the outgrown one -- stale
    flags and reads past its end).  A fresh context, a small call, then a larger one with many coefficients forced through the exact paths."""
    import gpqhe_amd
    torch = _torch()
    logn, logqL, logql = 14, 627, 62
    probe = gpqhe_amd.PolyContext(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    probe.close()
    assert dimA == 3
    n, W = 1 << logn, logqL // 64 + 1
    gen = torch.Generator(device="cuda")
    gen.manual_seed(62)
    small = [_centred(torch, gen, 1, W, n, logql) for _ in range(4)]
    large = [_centred(torch, gen, 5, W, n, logql) for _ in range(4)]
    ref = gpqhe_amd.PolyContext(logn, max(dimevk, 20))
    rlk = [torch.cat([torch.randint(0, ref.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    ref.set_stream_bridge(False); ref.set_lazy_decompose(False); ref.set_overlap(False)
    want = _run(ref, torch, large, rlk, W, logql, dimA, dimB, dimP)
    ref.close()
    for lanes in (False, True):
        g = gpqhe_amd.PolyContext(logn, max(dimevk, 20))
        g.set_overlap(lanes)
        g.set_chunk(2)
        g.debug_force_redo(3)
        _run(g, torch, small, rlk, W, logql, dimA, dimB, dimP)          # flag bytes sized for one ciphertext
        got = _run(g, torch, large, rlk, W, logql, dimA, dimB, dimP)    # ... grown inside this call
        g.close()
        for name, a, b in zip(("he_mul c0", "he_mul c1", "he_swk c0", "he_swk c1"), want, got):
            assert torch.equal(a, b), "%s (%s)" % (name, "two lanes" if lanes else "one lane")
