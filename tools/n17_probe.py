"""n = 2^17: NTT pair and he_mul tensor stage, per-kernel (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab
logn, dim, batch = 17, 44, 16
ctx = gpqhe_amd.PolyContext(logn, dim)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
s = [rand_slab(torch, ctx, dim, batch, gen) for _ in range(4)]
d = [torch.empty_like(s[0]) for _ in range(3)]
ws = ctx.tensor_workspace(dim, batch)
for name, fn in (("ntt+invntt", lambda: (ctx.poly_ntt(s[0], dim), ctx.poly_invntt(s[0], dim))),
                 ("tensor", lambda: ctx.he_mul_tensor(d[0], d[1], d[2], s[0], s[1], s[2], s[3], dim, ws))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t = gpqhe_amd.StreamTimer(); t.start()
    for _ in range(5): fn()
    t.stop()
    print(name, "n=2^17 dim=44 batch=16: %.3f ms" % (t.elapsed_ms() / 5))
