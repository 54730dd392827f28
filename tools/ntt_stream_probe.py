"""Standalone NTT + INTT pairs (the other half of BASELINE's metric) as ONE stream over the whole batch vs the batch split over two / four
HIP streams with their own contexts, so that the strided (HBM-bound) pass of one part runs under the contiguous (issue-bound) pass of
another and launch tails overlap.  Interleaved on one device; same bytes, same kernels.  """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab


def setup(logn, dim, batch, seed):
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    return ctx, rand_slab(torch, ctx, dim, batch, gen)


def run(parts, dim, iters):
    streams = [torch.cuda.Stream() for _ in parts]
    def once():
        for s, (ctx, slab) in zip(streams, parts):
            with torch.cuda.stream(s):
                ctx.poly_ntt(slab, dim)
                ctx.poly_invntt(slab, dim)
    for _ in range(4): once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for logn, dim, batch, iters in ((15, 10, 64, 40), (16, 30, 64, 20), (14, 24, 64, 40)):
    one = [setup(logn, dim, batch, 1)]
    two = [setup(logn, dim, batch // 2, 2 + i) for i in range(2)]
    four = [setup(logn, dim, batch // 4, 4 + i) for i in range(4)]
    byts = 2 * 16 * (1 << logn) * dim * batch
    for rnd in range(3):
        a, b, c = run(one, dim, iters), run(two, dim, iters), run(four, dim, iters)
        print("n=2^%d %d limbs batch %d, round %d: one stream %.4f ms (%.0f GB/s, %.3f of 8 TB/s) | two streams %.4f ms (%.0f GB/s, %.3f) | four streams %.4f ms (%.0f GB/s, %.3f)"
              % (logn, dim, batch, rnd, a, byts / a / 1e6, byts / a / 8e9, b, byts / b / 1e6, byts / b / 8e9, c, byts / c / 1e6, byts / c / 8e9), flush=True)
    for ctx, _ in one + two + four:
        ctx.close()
