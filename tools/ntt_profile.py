"""Per-kernel timing of the standalone NTT path (dev tool)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from gpqhe_amd import _native
if os.environ.get("NTT_LIB"):
    _native.use_variant(os.environ["NTT_LIB"])
from bench import rand_slab
for logn, dim, batch in ((16, 30, 64), (15, 10, 64), (17, 44, 8)):
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    slab = rand_slab(torch, ctx, dim, batch, gen)
    for _ in range(2):
        ctx.poly_ntt(slab, dim); ctx.poly_invntt(slab, dim)
    torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(int(os.environ.get('ITERS', '5'))):
        ctx.poly_ntt(slab, dim); ctx.poly_invntt(slab, dim)
    torch.cuda.synchronize()
    prof = ctx.profile_collect()
    nb = (1 << logn) * 8 * dim * batch
    print(logn, dim, batch, {k: (round(ms / c, 4), round(2 * nb / (ms / c * 1e-3) / 1e9)) for k, (ms, c) in prof.items()})
    ctx.close()
