"""BASELINE configs[4] shape: key-switch inner product (he_swk loop) at n = 2^17, 44 limbs (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab
logn, dim, batch = 17, 44, 16
ctx = gpqhe_amd.PolyContext(logn, dim)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
x = rand_slab(torch, ctx, dim, batch, gen)
e0, e1 = rand_slab(torch, ctx, dim, 1, gen), rand_slab(torch, ctx, dim, 1, gen)
c0, c1 = torch.empty_like(x), torch.empty_like(x)
ws = ctx.keyswitch_workspace(dim, batch)
for _ in range(2): ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
torch.cuda.synchronize()
ctx.profile(True)
t = gpqhe_amd.StreamTimer(); t.start()
for _ in range(5): ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
t.stop(); ms = t.elapsed_ms() / 5
prof = ctx.profile_collect()
algo = 5 * dim * (8 << logn) * batch
print("keyswitch n=2^17 dim=44 batch=%d: %.3f ms  %.0f key-switches/s  %.0f GB/s algorithmic" % (batch, ms, batch / ms * 1e3, algo / ms / 1e6))
print({k: round(v[0] / v[1], 4) for k, v in prof.items()})
