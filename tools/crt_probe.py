import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab
ctx = gpqhe_amd.PolyContext(16, 45)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
dim, batch, W, logq = 30, 16, 14, 850
slab = rand_slab(torch, ctx, dim, batch, gen)
big = torch.empty(batch * W * ctx.n, dtype=torch.int64, device="cuda")
t = gpqhe_amd.StreamTimer()
for mode in (False, True, False):
    ctx.set_exact_crt(mode)
    ctx.rns_reconstruct(big, W, slab, dim, logq)
    t.start()
    for _ in range(3): ctx.rns_reconstruct(big, W, slab, dim, logq)
    t.stop()
    print("exact" if mode else "fast", round(t.elapsed_ms() / 3, 3), "ms")
ctx.lib.gpq_debug_redo_count.restype = C.c_long
ctx.lib.gpq_debug_redo_count.argtypes = [C.c_void_p, C.c_size_t]
torch.cuda.synchronize()
print("flagged", ctx.lib.gpq_debug_redo_count(ctx.h, batch * ctx.n), "of", batch * ctx.n)
