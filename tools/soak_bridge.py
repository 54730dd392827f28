"""Randomised soak of the streaming bridge (gpqhe_amd/csrc/bridge_stream.hpp) against round 3's separate kernels: random rings (2^13 .. 2^15; now and then 2^8 .. 2^12 and 2^16),
moduli, levels, batch sizes, launch groups, forced-redo strides, one or two lanes (gpq_set_overlap); he_mul, a squaring, he_swk, poly_mul, he_mulpt and he_rs must give
identical words (the reference side of the comparison: separate kernels, canonical residues, one lane).
usage: python tools/soak_bridge.py [configs] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
ctxs = {}


def ctx_for(logn, nprimes):
    key = (logn, nprimes)
    if key not in ctxs:
        ctxs[key] = gpqhe_amd.PolyContext(logn, nprimes)
    return ctxs[key]


def centred(gen, batch, W, n, logq):
    big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
    wt = (logq - 2) // 64
    top = min(logq - 2 - 64 * wt, 62)            # (torch.randint's bounds are int64)
    big[:, wt] = torch.randint(-(1 << top), 1 << top, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
    for j in range(wt + 1, W):
        big[:, j] = big[:, wt] >> 63
    return big.reshape(-1).contiguous()


VERBOSE = os.environ.get("SOAK_VERBOSE") == "1"      # every call announced and waited for: the last line names what a device fault belongs to


def step(what, fn):
    if VERBOSE:
        print("  ...", what, flush=True)
    fn()
    if VERBOSE:
        torch.cuda.synchronize()


def run(g, cts, rlk, W, logql, dims, pt=None):
    dimP, dimA, dimB = dims
    if pt is None:
        pt = (cts[3] >> 20 if W == 1 else cts[3]).clone()       # some plaintext of the same shape
    o = [torch.empty_like(cts[0]) for _ in range(6)]
    step("he_mul", lambda: g.he_mul(o[0], o[1], *cts, rlk[0], rlk[1], W, logql, dimA, dimB, dimP))
    step("squaring", lambda: g.he_mul(o[2], o[3], cts[0], cts[1], cts[0], cts[1], rlk[0], rlk[1], W, logql, dimA, dimB, dimP))
    step("he_swk", lambda: g.he_swk(o[4], o[5], cts[2], cts[3], rlk[0], rlk[1], W, logql, dimB, dimP))
    # the other callers of the bridge (src/poly.c:84-107, src/he-mult.c:159-196, src/he-rescale.c:33-54): poly_mul, he_mulpt, he_rs
    o += [torch.empty_like(cts[0]) for _ in range(3)] + [cts[2].clone(), cts[3].clone()]
    step("poly_mul", lambda: g.poly_mul(o[6], cts[0], cts[1], W, dimA, logql))
    step("he_mulpt", lambda: g.he_mulpt(o[7], o[8], cts[0], cts[1], pt, W, logql, dimA))
    ld = min(50, logql - 8)
    step("he_rs", lambda: g.he_rs(o[9], o[10], W, ld, logql - ld))
    # he_mul + he_rs as one call (round 5): fused into the streaming tail on the side under test, two calls inside on the reference side
    o += [torch.empty_like(cts[0]) for _ in range(2)]
    step("he_mul_rs", lambda: g.he_mul_rs(o[11], o[12], *cts, rlk[0], rlk[1], W, logql, dimA, dimB, dimP, ld))
    torch.cuda.synchronize()
    return o


t0 = time.time()
streamed = 0
for it in range(N):
    logn = rng.choice((13, 13, 14, 14, 15) if rng.random() < 0.85 else (8, 10, 12, 16))     # now and then the small rings (no two-pass transform) and the headline ring
    logqL = rng.randrange(100, 881)
    if logn == 15: logqL = min(logqL, 600)
    logql = logqL if rng.random() < 0.5 else rng.randrange(60, logqL + 1)
    batch, chunk, force = rng.randrange(1, 8), rng.choice((1, 2, 3, 32)), rng.choice((0, 0, 1, 7, 64, 257))
    if logn == 16: batch = min(batch, 3)
    lanes = rng.choice((0, 1, 1))
    probe = ctx_for(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    g = ctx_for(logn, max(dimevk, 20))
    n, W = g.n, logqL // 64 + 1
    gen = torch.Generator(device="cuda"); gen.manual_seed(rng.randrange(1 << 30))
    cts = [centred(gen, batch, W, n, logql) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    if VERBOSE:
        print("config %d: logn %d q_L 2^%d q_l 2^%d dims P/A/B %d/%d/%d W %d batch %d chunk %d force %d lanes %d" % (it, logn, logqL, logql, dimP, dimA, dimB, W, batch, chunk, force, lanes + 1), flush=True)
    g.set_chunk(chunk)
    g.set_stream_bridge(False); g.set_lazy_decompose(False); g.debug_force_redo(0); g.set_overlap(False)
    ref_exact = rng.random() < 0.5                               # half of the time the reference side is the most conservative path: exact CRT, integer-VALU decompose
    g.set_exact_crt(ref_exact); g.set_bridge_mfma(not ref_exact)
    if VERBOSE: print(" separate kernels, one lane", flush=True)
    want = run(g, cts, rlk, W, logql, (dimP, dimA, dimB))
    g.set_exact_crt(False); g.set_bridge_mfma(True)
    # the side under test: the context with its history, or (a third of the time) a FRESH context -- every scratch buffer, table and the peer lane
    # made inside these calls -- after a smaller call of the same kind (buffers grow inside the larger one), on a non-blocking side stream
    fresh = rng.random() < 0.33
    h = gpqhe_amd.PolyContext(logn, max(dimevk, 20)) if fresh else g
    h.set_chunk(chunk); h.set_stream_bridge(True); h.set_lazy_decompose(True); h.debug_force_redo(force); h.set_overlap(lanes)
    # a quarter of the time one of the older kernel families instead of the default (every setting must give the same words, also from a fresh
    # context, after growth, on a side stream): prescale 0 / 1 / 2 (+ the fused tail), the integer-VALU bridge, the exact CRT everywhere
    other = rng.choice((None, None, None, "prescale0", "prescale1", "prescale2", "prescale2+fused", "valu", "exact", "nostream"))
    if other == "prescale0": h.set_prescale(0)
    elif other == "prescale1": h.set_prescale(1)
    elif other == "prescale2": h.set_prescale(2)
    elif other == "prescale2+fused": h.set_prescale(2); h.set_fused_tail(True)
    elif other == "valu": h.set_bridge_mfma(False)
    elif other == "exact": h.set_exact_crt(True)
    elif other == "nostream": h.set_stream_bridge(False)
    if VERBOSE: print(" streaming bridge%s" % (" (fresh context, side stream, small call first)" if fresh else ""), flush=True)
    if fresh:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if batch > 1:
                run(h, [t[: t.numel() // batch].contiguous() for t in cts], rlk, W, logql, (dimP, dimA, dimB))
            got = run(h, cts, rlk, W, logql, (dimP, dimA, dimB))
        h.close()
    else:
        got = run(h, cts, rlk, W, logql, (dimP, dimA, dimB))
    g.debug_force_redo(0); g.set_chunk(32); g.set_overlap(True)
    g.set_prescale(3); g.set_fused_tail(False); g.set_bridge_mfma(True); g.set_exact_crt(False); g.set_stream_bridge(True)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if not torch.equal(a, b)]
    if bad:
        print("MISMATCH at config %d: logn %d logqL %d logql %d dims %s batch %d chunk %d force %d lanes %d fresh %d settings %s outputs %s" % (it, logn, logqL, logql, (dimP, dimA, dimB), batch, chunk, force, lanes + 1, fresh, other, bad), flush=True)
        sys.exit(1)
    if it % 10 == 0:
        print("config %d ok: logn %d q_L 2^%d q_l 2^%d dims P/A/B %d/%d/%d W %d batch %d chunk %d force %d lanes %d%s (%.0f s)" % (it, logn, logqL, logql, dimP, dimA, dimB, W, batch, chunk, force, lanes + 1, (" fresh" if fresh else "") + (" " + other if other else ""), time.time() - t0), flush=True)
print("soak_bridge ok: %d configurations, every word of he_mul / squaring / he_swk / poly_mul / he_mulpt / he_rs / he_mul_rs equal between the streaming bridge and the separate kernels (%.0f s)" % (N, time.time() - t0))
