import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import keyswitch_n17_rate
for b in (16, 32, 64):
    print(keyswitch_n17_rate(torch, gpqhe_amd, b, 5), flush=True)
# the same stage at n=2^16, 45 limbs for comparison
from bench import rand_slab
ctx = gpqhe_amd.PolyContext(16, 45)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for b in (16, 64):
    x = rand_slab(torch, ctx, 45, b, gen); e0, e1 = rand_slab(torch, ctx, 45, 1, gen), rand_slab(torch, ctx, 45, 1, gen)
    c0, c1 = torch.empty_like(x), torch.empty_like(x); ws = ctx.keyswitch_workspace(45, b)
    ctx.he_keyswitch(c0, c1, x, e0, e1, 45, ws)
    t = gpqhe_amd.StreamTimer(); t.start()
    for _ in range(5): ctx.he_keyswitch(c0, c1, x, e0, e1, 45, ws)
    t.stop(); ms = t.elapsed_ms() / 5
    print("n=2^16 45 limbs batch %d: %.3f ms, %.0f /s, %.0f GB/s" % (b, ms, b / ms * 1e3, 5 * 45 * (8 << 16) * b / ms / 1e6))
