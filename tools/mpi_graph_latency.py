"""single he_mul (batch 1) on device slabs: eager launches vs a replayed HIP graph (dev tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from gpqhe_amd import _native
from gpqhe_amd.engine import _ptr, _stream
logn, logq, W = 16, 850, 14
g = gpqhe_amd.PolyContext(logn, 45)
dimP, dimA, dimB, _ = g.he_dims(logq, logq)
n = g.n
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
def centred():
    big = torch.randint(-(1 << 62), 1 << 62, (W, n), dtype=torch.int64, device="cuda", generator=gen)
    big[W - 1] = torch.randint(-(1 << 16), 1 << 16, (n,), dtype=torch.int64, device="cuda", generator=gen)
    return big.reshape(-1).contiguous()
ins = [centred() for _ in range(4)]
rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
o0, o1 = torch.empty_like(ins[0]), torch.empty_like(ins[0])
ws = torch.empty(g.lib.gpq_he_mul_workspace_bytes(g.h, W, dimA, dimB, dimP, 1) // 8 + 8, dtype=torch.int64, device="cuda")
def call():
    _native.check(g.lib.gpq_he_mul(g.h, _ptr(o0), _ptr(o1), *[_ptr(v) for v in ins], _ptr(rlk[0]), _ptr(rlk[1]), W, logq, dimA, dimB, dimP, 1, _ptr(ws), _stream()), "he_mul")
for _ in range(3): call()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    call()
for name, f in (("eager", call), ("graph", graph.replay)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 50 * 1e3, 4), "ms per he_mul")
