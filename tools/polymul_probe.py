"""gpq_poly_mul_rns rate (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab
for logn, dim, batch in ((16, 30, 64), (14, 16, 64), (17, 44, 16)):
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    a0, b0 = rand_slab(torch, ctx, dim, batch, gen), rand_slab(torch, ctx, dim, batch, gen)
    a, b, r = a0.clone(), b0.clone(), torch.empty_like(a0)
    for _ in range(2): ctx.poly_mul_rns(r, a, b, dim)
    torch.cuda.synchronize()
    t = gpqhe_amd.StreamTimer(); t.start()
    for _ in range(5): ctx.poly_mul_rns(r, a, b, dim)
    t.stop()
    ms = t.elapsed_ms() / 5
    print("poly_mul_rns n=2^%d dim=%d batch=%d: %.3f ms  %.0f products/s  %.0f GB/s algorithmic (2R+1W)" % (logn, dim, batch, ms, batch / ms * 1e3, 3 * dim * (8 << logn) * batch / ms / 1e6))
    ctx.close()
