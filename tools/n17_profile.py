"""rocprofv3 target (dev tool): BASELINE configs[4]'s shape on one lane -- N17_WHAT=keyswitch: the key-switch inner product (gpq_keyswitch, 44 limbs,
n = 2^17, batch N17_BATCH); N17_WHAT=he_swk: the whole he_swk (bench.he_swk_mpi_rate).  One lane (gpq_set_overlap(ctx, 0)): a kernel trace / PMC
pass should time a kernel with nothing running beside it."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpqhe_amd
from gpqhe_amd import _native
if os.environ.get("N17_LIB"):
    _native.use_variant(os.environ["N17_LIB"])
import bench

what, batch, iters = os.environ.get("N17_WHAT", "keyswitch"), int(os.environ.get("N17_BATCH", "64")), int(os.environ.get("N17_ITERS", "5"))
torch.cuda.set_device(0)
if what == "he_swk":
    r = bench.he_swk_mpi_rate(torch, gpqhe_amd, batch=batch, iters=iters, checked=None)
    r.pop("_check", None)
    print(r)
else:
    logn, dim = int(os.environ.get("N17_LOGN", "17")), int(os.environ.get("N17_DIM", "44"))
    ctx = gpqhe_amd.PolyContext(logn, dim)
    ctx.set_overlap(False)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(17)
    x = bench.rand_slab(torch, ctx, dim, batch, gen)
    e0, e1 = bench.rand_slab(torch, ctx, dim, 1, gen), bench.rand_slab(torch, ctx, dim, 1, gen)
    c0, c1 = torch.empty_like(x), torch.empty_like(x)
    ws = ctx.keyswitch_workspace(dim, batch)
    for _ in range(3):
        ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
    torch.cuda.synchronize()
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
    t.stop()
    ms = t.elapsed_ms() / iters
    print({"shape": "n=2^%d, %d limbs, batch %d, one lane" % (logn, dim, batch), "ms_per_batch": round(ms, 3),
           "algo_GBps": round(5 * dim * (8 << logn) * batch / (ms * 1e-3) / 1e9, 1)})
