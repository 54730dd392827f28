"""dev tool (round 5): launch-group size of gpq_keyswitch at n = 2^17, 44 limbs, batch 64, one lane and two, interleaved on one device."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpqhe_amd
import bench

torch.cuda.set_device(0)
logn, dim, batch = 17, 44, 64
ctx = gpqhe_amd.PolyContext(logn, dim)
gen = torch.Generator(device="cuda"); gen.manual_seed(17)
x = bench.rand_slab(torch, ctx, dim, batch, gen)
e0, e1 = bench.rand_slab(torch, ctx, dim, 1, gen), bench.rand_slab(torch, ctx, dim, 1, gen)
c0, c1 = torch.empty_like(x), torch.empty_like(x)
t = gpqhe_amd.StreamTimer()
algo = 5 * dim * (8 << logn) * batch
res = {}
for rnd in range(4):
    for chunk in (8, 16, 32, 64):
        for lanes in (0, 1):
            ctx.set_chunk(chunk); ctx.set_overlap(bool(lanes))
            ws = ctx.keyswitch_workspace(dim, batch)
            for _ in range(2):
                ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
            torch.cuda.synchronize()
            t.start()
            for _ in range(4):
                ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
            t.stop()
            res.setdefault((chunk, lanes), []).append(t.elapsed_ms() / 4)
for (chunk, lanes), v in sorted(res.items()):
    m = sorted(v)[len(v) // 2]
    print("chunk %2d lanes %d: %.3f ms per 64 (%s)  %.1f GB/s = %.4f of 8 TB/s" % (chunk, lanes + 1, m, " ".join("%.3f" % a for a in v), algo / m / 1e6, algo / m / 1e6 / 8000))
