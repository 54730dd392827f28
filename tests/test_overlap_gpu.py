"""gpq_set_overlap (include/gpqhe_hip.h): gpq_he_mul / gpq_he_swk over more than one launch group run every other group on a second internal
stream through a peer context (src/he-mult.c:116-141, :40-85 and src/he-automorphism.c:40-85 per ciphertext are independent: the loop over the
batch is this library's).  Same words as the single-stream order, for multiplications, squarings and key switches, with coefficients forced
through the exact paths, with a ragged last group; the call stays ordered on the caller's stream (inputs written just before it, outputs read
just after it, on a non-default stream, no host synchronisation in between); and it can be captured into a HIP graph with both lanes."""
import pytest

pytestmark = pytest.mark.gpu


def _centred(torch, gen, batch, W, n, logq):
    """dense random values in [-2^(logq-2), 2^(logq-2)) as W two's-complement words (as in test_stream_bridge_gpu.py)"""
    big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
    wt = (logq - 2) // 64
    top = min(logq - 2 - 64 * wt, 62)
    big[:, wt] = torch.randint(-(1 << top), 1 << top, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
    for j in range(wt + 1, W):
        big[:, j] = big[:, wt] >> 63
    return big.reshape(-1).contiguous()


def _setup(engine_ctx, logn, logq, batch, seed):
    import torch
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g = engine_ctx(logn, dimevk)
    n, W = g.n, (logq + 64) // 64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    cts = [_centred(torch, gen, batch, W, n, logq) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    g._test_more = lambda: [_centred(torch, gen, batch, W, n, logq) for _ in range(4)]      # another valid set of operands, same shapes
    return g, cts, rlk, W, (dimA, dimB, dimP)


def _run(g, torch, cts, rlk, W, logq, dims, square=False):
    dimA, dimB, dimP = dims
    o = [torch.empty_like(cts[0]) for _ in range(4)]
    if square:
        g.he_mul(o[0], o[1], cts[0], cts[1], cts[0], cts[1], rlk[0], rlk[1], W, logq, dimA, dimB, dimP)
    else:
        g.he_mul(o[0], o[1], *cts, rlk[0], rlk[1], W, logq, dimA, dimB, dimP)
    g.he_swk(o[2], o[3], cts[2], cts[3], rlk[0], rlk[1], W, logq, dimB, dimP)
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize("logn,logq,batch,chunk", [(13, 438, 7, 2), (14, 438, 5, 1), (16, 850, 3, 1), (10, 130, 9, 4)])
def test_two_lanes_give_the_words_of_one(engine_ctx, logn, logq, batch, chunk):
    import torch
    g, cts, rlk, W, dims = _setup(engine_ctx, logn, logq, batch, 500 + logn + batch)
    try:
        g.set_chunk(chunk)
        g.set_overlap(False)
        want, want_sq = _run(g, torch, cts, rlk, W, logq, dims), _run(g, torch, cts, rlk, W, logq, dims, square=True)
        g.set_overlap(True)
        got, got_sq = _run(g, torch, cts, rlk, W, logq, dims), _run(g, torch, cts, rlk, W, logq, dims, square=True)
        g.debug_force_redo(7)                       # the peer follows the context's settings: its exact paths run too
        forced = _run(g, torch, cts, rlk, W, logq, dims)
        g.debug_force_redo(0)
        g.set_stream_bridge(False)                  # ... and round 3's separate kernels on both lanes
        separate = _run(g, torch, cts, rlk, W, logq, dims)
    finally:
        g.debug_force_redo(0)
        g.set_stream_bridge(True)
        g.set_overlap(True)
        g.set_chunk(32)
    for name, a, b, c, d, e in zip(("he_mul c0", "he_mul c1", "he_swk c0", "he_swk c1"), want, got, forced, separate, zip(want_sq, got_sq)):
        assert torch.equal(a, b), name
        assert torch.equal(a, c), name + " (forced exact paths)"
        assert torch.equal(a, d), name + " (separate kernels)"
        assert torch.equal(e[0], e[1]), name + " (squaring)"
    assert bool((want[0] != 0).any()) and bool((want[3] != 0).any())


def test_the_call_stays_ordered_on_the_callers_stream(engine_ctx):
    """inputs produced on the caller's stream right before the call, outputs consumed on it right after, no host synchronisation: the peer's
    stream must wait for the first and the caller's stream for the peer"""
    import torch
    logn, logq, batch = 13, 438, 6
    g, cts, rlk, W, dims = _setup(engine_ctx, logn, logq, batch, 77)
    dimA, dimB, dimP = dims
    other = g._test_more()
    try:
        g.set_chunk(2)
        g.set_overlap(False)
        want = []
        for src in (cts, other):
            o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
            g.he_mul(o0, o1, *src, rlk[0], rlk[1], W, logq, dimA, dimB, dimP)
            torch.cuda.synchronize()
            want.append((o0, o1))
        g.set_overlap(True)
        side = torch.cuda.Stream()
        ins = [torch.empty_like(t) for t in cts]
        o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
        got = []
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for rnd in range(6):
                src = (cts, other)[rnd & 1]
                for dst, s in zip(ins, src):
                    dst.copy_(s)                     # queued on `side`, not waited for
                g.he_mul(o0, o1, *ins, rlk[0], rlk[1], W, logq, dimA, dimB, dimP)
                got.append((rnd & 1, o0.clone(), o1.clone()))     # queued right behind the call
        torch.cuda.synchronize()
    finally:
        g.set_overlap(True)
        g.set_chunk(32)
    for which, a, b in got:
        assert torch.equal(a, want[which][0]) and torch.equal(b, want[which][1])


def test_two_lanes_in_a_hip_graph(engine_ctx):
    """after a warm-up call (peer context and workspaces exist) the fork / join over the two streams is captured with the launches"""
    import torch
    logn, logq, batch = 13, 438, 4
    g, cts, rlk, W, dims = _setup(engine_ctx, logn, logq, batch, 91)
    dimA, dimB, dimP = dims
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    try:
        g.set_chunk(1)
        nbytes = g.lib.gpq_he_mul_workspace_bytes(g.h, W, dimA, dimB, dimP, batch)
        ws = torch.empty(nbytes // 8 + 8, dtype=torch.int64, device="cuda")

        def call():
            from gpqhe_amd import _native
            from gpqhe_amd.engine import _ptr, _stream
            _native.check(g.lib.gpq_he_mul(g.h, _ptr(o0), _ptr(o1), *[_ptr(v) for v in cts], _ptr(rlk[0]), _ptr(rlk[1]), W, logq, dimA, dimB, dimP,
                                           batch, _ptr(ws), _stream()), "gpq_he_mul")

        call()
        torch.cuda.synchronize()
        want = (o0.clone(), o1.clone())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            call()
        for rnd in range(3):
            o0.zero_(); o1.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(o0, want[0]) and torch.equal(o1, want[1])
        fresh = g._test_more()
        for dst, src in zip(cts, fresh):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        r0, r1 = o0.clone(), o1.clone()
        g.set_overlap(False)
        call()
        torch.cuda.synchronize()
        assert torch.equal(r0, o0) and torch.equal(r1, o1) and not torch.equal(r0, want[0])
    finally:
        g.set_overlap(True)
        g.set_chunk(32)


@pytest.mark.parametrize("logn,dim,batch,chunk", [(13, 5, 7, 2), (16, 4, 5, 2), (17, 3, 3, 1)])
def test_rns_core_on_two_lanes(engine_ctx, logn, dim, batch, chunk):
    """gpq_he_mul_tensor / gpq_keyswitch (the limb loops of src/he-mult.c:116-138, :58-66) over several launch groups: same words on one and two lanes"""
    import torch
    from bench import rand_slab
    g = engine_ctx(logn, dim)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(40 + logn)
    a = [rand_slab(torch, g, dim, batch, gen) for _ in range(5)]
    evk = [rand_slab(torch, g, dim, 1, gen) for _ in range(2)]

    def run():
        d = [torch.empty_like(a[0]) for _ in range(3)]
        g.he_mul_tensor(d[0], d[1], d[2], a[0], a[1], a[2], a[3], dim)
        sq = [torch.empty_like(a[0]) for _ in range(3)]
        g.he_mul_tensor(sq[0], sq[1], sq[2], a[0], a[1], a[0], a[1], dim)
        k = [torch.empty_like(a[0]) for _ in range(2)]
        g.he_keyswitch(k[0], k[1], a[4], evk[0], evk[1], dim)
        torch.cuda.synchronize()
        return d + sq + k

    try:
        g.set_chunk(chunk)
        g.set_overlap(False)
        want = run()
        g.set_overlap(True)
        got = run()
        again = run()
    finally:
        g.set_overlap(True)
        g.set_chunk(32)
    for i, (w, x, y) in enumerate(zip(want, got, again)):
        assert torch.equal(w, x) and torch.equal(w, y), "result %d" % i
    assert bool((want[1] != 0).any()) and bool((want[7] != 0).any())


def test_a_peer_that_cannot_be_created_means_one_lane_not_an_error():
    """when the peer context (or its workspace) cannot be allocated the call must run on the caller's stream alone -- same words,
    no error, and the context stops trying (gpq_debug_fail_peer stands in for the failing allocation); cleared, the lanes come back.
    A context of its own: the session-cached ones may own a peer already."""
    import torch
    import gpqhe_amd
    logn, logq, batch = 13, 300, 5
    probe = gpqhe_amd.PolyContext(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    probe.close()
    g = gpqhe_amd.PolyContext(logn, dimevk)
    try:
        n, W = g.n, (logq + 64) // 64
        gen = torch.Generator(device="cuda")
        gen.manual_seed(31)
        cts = [_centred(torch, gen, batch, W, n, logq) for _ in range(4)]
        rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
        dims = (dimA, dimB, dimP)
        g.set_chunk(2)
        g.set_overlap(False)
        want = _run(g, torch, cts, rlk, W, logq, dims)
        g.set_overlap(True)
        g.debug_fail_peer(True)
        got = _run(g, torch, cts, rlk, W, logq, dims)
        assert g.last_lanes() == 1
        again = _run(g, torch, cts, rlk, W, logq, dims)            # the context does not retry call after call
        assert g.last_lanes() == 1
        g.debug_fail_peer(False)
        back = _run(g, torch, cts, rlk, W, logq, dims)
        assert g.last_lanes() == 2
        for a, b, c, d in zip(want, got, again, back):
            assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    finally:
        g.close()


def test_the_peer_lane_borrows_every_table_and_a_failed_workspace_declines_that_size_only():
    """Round 6: the peer lane owns no read-only device table -- twiddles, split pairs, LimbTab and the bridge's
    constant cache are the parent's, by pointer -- so the second lane costs its workspace only; and a workspace that cannot be allocated declines
    shapes of that size, not the lane (a smaller shape still runs on two); what exists for a lane that has not run yet is a context object and its flag words."""
    import torch
    import gpqhe_amd
    logn, logq = 13, 300
    probe = gpqhe_amd.PolyContext(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    probe.close()
    g = gpqhe_amd.PolyContext(logn, dimevk)
    try:
        n, W = g.n, (logq + 64) // 64
        gen = torch.Generator(device="cuda")
        gen.manual_seed(32)
        rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
        dims = (dimA, dimB, dimP)
        big = [_centred(torch, gen, 8, W, n, logq) for _ in range(4)]
        small = [t[:4 * W * n] for t in big]
        g.set_overlap(False)
        g.set_chunk(4)
        want_big = _run(g, torch, big, rlk, W, logq, dims)         # groups of 4: the larger workspace
        g.set_chunk(2)
        want_small = _run(g, torch, small, rlk, W, logq, dims)     # groups of 2: the smaller one
        assert g.debug_table_bytes(2) == 0                          # one lane so far: no peer exists
        owned = g.debug_table_bytes(0)
        assert owned > 2 * dimevk * n * 8                           # at least the two plain twiddle tables
        # the workspace of the larger shape "does not fit": one lane for it, and no peer was created for nothing before the price was known
        g.set_overlap(True)
        g.set_chunk(4)
        g.debug_fail_peer(2)
        got_big = _run(g, torch, big, rlk, W, logq, dims)
        assert g.last_lanes() == 1
        assert g.debug_table_bytes(2) == 1                          # (the peer itself is cheap and exists; it holds no workspace)
        g.set_chunk(2)
        g.debug_fail_peer(3)                                        # 3: allocations work again, what was declined stays declined
        got_small = _run(g, torch, small, rlk, W, logq, dims)
        assert g.last_lanes() == 2
        g.set_chunk(4)
        again_big = _run(g, torch, big, rlk, W, logq, dims)         # the declined size stays declined
        assert g.last_lanes() == 1
        g.debug_fail_peer(False)                                    # cleared: the larger shape gets its lane
        back_big = _run(g, torch, big, rlk, W, logq, dims)
        assert g.last_lanes() == 2
        for a, b, c, d in zip(want_big, got_big, again_big, back_big):
            assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
        for a, b in zip(want_small, got_small):
            assert torch.equal(a, b)
        # the peer exists now and holds no table of its own; the parent's tables did not grow by a second copy
        assert g.debug_table_bytes(2) == 1 and g.debug_table_bytes(1) == 0 and g.debug_table_bytes(3) == 1
        assert g.debug_table_bytes(0) < owned + (1 << 20)          # (+ the constants the two-lane shapes built at first use, no second set of twiddles)
    finally:
        g.close()
