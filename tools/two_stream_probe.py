"""Whole he_mul (device slabs, n = 2^16, q = 2^850) as one stream of 64 ciphertexts vs two streams of 32 with their own contexts:
do the HBM-bound bridge kernels of one half overlap usefully with the VALU / power-bound transforms of the other?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab

LOGN, logq, W = 16, 850, 14
def setup(batch, seed):
    ctx = gpqhe_amd.PolyContext(LOGN, 45)
    dimP, dimA, dimB, _ = ctx.he_dims(logq, logq)
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, ctx.n), dtype=torch.int64, device="cuda", generator=gen)
        big[:, W - 1] = torch.randint(-(1 << 16), 1 << 16, (batch, ctx.n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()
    cts = [centred() for _ in range(4)]
    rlk = rand_slab(torch, ctx, dimB, 1, gen), rand_slab(torch, ctx, dimB, 1, gen)
    outs = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    return ctx, cts, rlk, outs, (dimA, dimB, dimP)

def run(parts, iters=6):
    streams = [torch.cuda.Stream() for _ in parts]
    def once():
        for s, (ctx, cts, rlk, outs, (dA, dB, dP)) in zip(streams, parts):
            with torch.cuda.stream(s):
                ctx.he_mul(outs[0], outs[1], *cts, rlk[0], rlk[1], W, logq, dA, dB, dP)
    for _ in range(2): once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

one = [setup(64, 21)]
two = [setup(32, 21), setup(32, 22)]
four = [setup(16, 21 + i) for i in range(4)]
for rnd in range(3):
    a, b, c = run(one), run(two), run(four)
    print("round %d: one stream x64 %.3f ms (%.0f he_mul/s) | two streams x32 %.3f ms (%.0f/s) | four streams x16 %.3f ms (%.0f/s)"
          % (rnd, a, 64e3 / a, b, 64e3 / b, c, 64e3 / c), flush=True)
