"""Whole he_mul (src/he-mult.c:88-156 on device slabs) issued call by call against the same calls replayed from a HIP graph (dev tool, GPU box).
The C ABI is capture-safe once a shape has run (tests/test_overlap_gpu.py::test_two_lanes_in_a_hip_graph); this prices what a caller gains by
capturing, at the reference's own default shape (logn 14, q = 2^438: launch-bound) and at the headline shape (logn 16, q = 2^850: not).

    python tools/graph_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gpqhe_amd  # noqa: E402
from gpqhe_amd import _native  # noqa: E402
from gpqhe_amd.engine import _ptr, _stream  # noqa: E402


def centred(gen, batch, W, n, logq):
    big = torch.randint(-(1 << 62), 1 << 62, (batch, W, n), dtype=torch.int64, device="cuda", generator=gen)
    top = logq - 2 - 64 * (W - 1)
    big[:, W - 1] = torch.randint(-(1 << top), 1 << top, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
    return big.reshape(-1).contiguous()


def run(logn, nprimes, logq, batch, lanes, calls_per_graph=4, reps=5):
    g = gpqhe_amd.PolyContext(logn, nprimes)
    dimP, dimA, dimB, _ = g.he_dims(logq, logq)
    n, W = g.n, logq // 64 + 1
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    cts = [centred(gen, batch, W, n, logq) for _ in range(4)]
    rlk = [torch.cat([torch.randint(0, g.p[d], (n,), dtype=torch.int64, device="cuda", generator=gen) for d in range(dimB)]) for _ in range(2)]
    o0, o1 = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    g.set_overlap(lanes)
    ws = torch.empty(g.lib.gpq_he_mul_workspace_bytes(g.h, W, dimA, dimB, dimP, batch) // 8 + 8, dtype=torch.int64, device="cuda")

    def call():
        _native.check(g.lib.gpq_he_mul(g.h, _ptr(o0), _ptr(o1), *[_ptr(v) for v in cts], _ptr(rlk[0]), _ptr(rlk[1]), W, logq, dimA, dimB, dimP, batch, _ptr(ws), _stream()), "gpq_he_mul")

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        want = (o0.clone(), o1.clone())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(calls_per_graph):
                call()
        o0.zero_(); o1.zero_()
        graph.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(o0, want[0]) and torch.equal(o1, want[1]))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            call()
            torch.cuda.synchronize()
        res = {"direct": [], "graph": []}
        for _ in range(reps):                                   # interleaved
            for mode in ("direct", "graph"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    if mode == "graph":
                        graph.replay()
                    else:
                        for _ in range(calls_per_graph):
                            call()
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) / (6 * calls_per_graph))
    d, gr = sorted(res["direct"])[reps // 2], sorted(res["graph"])[reps // 2]
    print("logn %d q=2^%d batch %3d lanes %d: direct %.4f ms per call (%.0f he_mul/s), graph replay %.4f ms (%.0f he_mul/s): %+.1f %%   same words %s"
          % (logn, logq, batch, lanes + 1, d, batch / d * 1e3, gr, batch / gr * 1e3, (d / gr - 1) * 100, same), flush=True)
    g.close()


if __name__ == "__main__":
    torch.cuda.set_device(0)
    for batch in (1, 4, 16, 64):
        for lanes in (0, 1):
            run(14, 24, 438, batch, lanes)
    for batch in (1, 8, 64):
        run(16, 45, 850, batch, 1, calls_per_graph=2)
