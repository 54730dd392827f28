"""Does keeping a launch group's slabs inside the 256 MiB Infinity Cache buy time or power?  (dev tool, GPU box)

The tensor stage (strided forward x4 -> tensor_mid8 -> strided inverse x3) is run `reps` times over
  hot : the SAME buffer set every time   (footprint = one group's inputs + scratch + outputs)
  cold: `sets` different buffer sets in rotation (footprint >> Infinity Cache), identical launches
for group shapes whose hot footprint runs from well inside to well outside the cache.  Kernel code, grids and
launch order are identical, so hot/cold isolates what residency of the slabs is worth.  Power is polled like
bench.py's `power` object.

    python tools/mall_probe.py [reps]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import gpqhe_amd  # noqa: E402
from bench import rand_slab, power_state  # noqa: E402


def run(ctx, dim, batch, sets, reps):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    n = ctx.n
    bufs = []
    for _ in range(sets):
        ins = [rand_slab(torch, ctx, dim, batch, gen) for _ in range(4)]
        outs = [torch.empty_like(ins[0]) for _ in range(3)]
        ws = ctx.tensor_workspace(dim, batch)
        bufs.append((ins, outs, ws))
    foot = (4 + 4 + 3) * dim * batch * n * 8 / 2**20

    def loop(which):
        def step():
            for r in range(reps):
                ins, outs, ws = bufs[0 if which == "hot" else r % sets]
                ctx.he_mul_tensor(outs[0], outs[1], outs[2], *ins, dim, ws)
        return step

    res = {"dim": dim, "batch": batch, "hot_footprint_MiB": round(foot, 1), "mid_workgroups": 32 * batch * dim}
    for which in ("cold", "hot", "cold", "hot"):
        step = loop(which)
        step()
        torch.cuda.synchronize()
        t = gpqhe_amd.StreamTimer()
        t.start()
        step()
        t.stop()
        ms = t.elapsed_ms() / reps
        res.setdefault(which + "_ms", []).append(round(ms, 4))
    for which in ("cold", "hot"):
        pw = power_state(torch, loop(which), seconds=2.0)
        if pw:
            res[which + "_W"], res[which + "_MHz"] = pw["package_W"], pw["sclk_MHz"]
    algo = 7 * dim * batch * n * 8
    res["hot_algo_GBps"] = round(algo / (min(res["hot_ms"]) * 1e-3) / 1e9, 1)
    res["cold_algo_GBps"] = round(algo / (min(res["cold_ms"]) * 1e-3) / 1e9, 1)
    return res


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.cuda.set_device(0)
    ctx = gpqhe_amd.PolyContext(16, 45)
    ctx.set_chunk(64)
    for dim, batch in ((1, 16), (1, 32), (2, 32), (1, 64), (4, 32), (30, 32)):
        foot = 11 * dim * batch * 0.5
        sets = max(2, min(16, int(2048 / foot)))          # cold rotation touches >= 1 GiB (capped at 16 sets)
        if dim == 30:
            sets = 2
        print(json.dumps(run(ctx, dim, batch, sets, reps if dim < 30 else 4)), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
