"""Interleaved A/B on one device (dev tool, round 5): the 13 last limbs of the n = 2^17 chain (c >= 2^27.415) in the 7-mad class (as until round 4:
gpq_set_limb_classes(wide, 31)) against the widened split class (default: all 44 limbs split or wide-split).  gpq_keyswitch at BASELINE configs[4]'s
shape (44 limbs, batch 64), one lane, and a forward + inverse NTT pair of the same slab."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpqhe_amd
import bench

torch.cuda.set_device(0)
logn, dim, batch = 17, 44, 64
ctx = gpqhe_amd.PolyContext(logn, dim)
ctx.set_overlap(False)
gen = torch.Generator(device="cuda"); gen.manual_seed(17)
x = bench.rand_slab(torch, ctx, dim, batch, gen)
e0, e1 = bench.rand_slab(torch, ctx, dim, 1, gen), bench.rand_slab(torch, ctx, dim, 1, gen)
c0, c1 = torch.empty_like(x), torch.empty_like(x)
ws = ctx.keyswitch_workspace(dim, batch)
cs = [p - (1 << 59) for p in ctx.p[:dim]]
nwide = sum(1 for c in cs if c < 134217000)
nold = sum(1 for c in cs if c < 178956971)
print("limbs: %d wide-split, %d more below the old split bound 2^27.415, %d above it" % (nwide, nold - nwide, dim - nold))
ref = None
t = gpqhe_amd.StreamTimer()
rows = {"old (last %d limbs 7-mad)" % (dim - nold): (nwide, nold), "new (all split)": (nwide, 99)}
res = {k: {"ks": [], "ntt": []} for k in rows}
for rnd in range(5):
    for name, (w, s) in rows.items():
        ctx.set_limb_classes(w, s)
        for _ in range(2):
            ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
        torch.cuda.synchronize()
        t.start()
        for _ in range(4):
            ctx.he_keyswitch(c0, c1, x, e0, e1, dim, ws)
        t.stop()
        res[name]["ks"].append(t.elapsed_ms() / 4)
        if ref is None:
            ref = (c0.clone(), c1.clone())
        else:
            assert torch.equal(ref[0], c0) and torch.equal(ref[1], c1), "the two class assignments differ"
        y = x.clone()
        ctx.poly_ntt(y, dim); ctx.poly_invntt(y, dim)
        assert torch.equal(y, x)
        torch.cuda.synchronize()
        t.start()
        for _ in range(3):
            ctx.poly_ntt(y, dim); ctx.poly_invntt(y, dim)
        t.stop()
        res[name]["ntt"].append(t.elapsed_ms() / 3)
algo = 5 * dim * (8 << logn) * batch
for name, r in res.items():
    ks, nt = sorted(r["ks"])[len(r["ks"]) // 2], sorted(r["ntt"])[len(r["ntt"]) // 2]
    print("%-28s keyswitch %.3f ms per 64 (%s)  %.1f GB/s algorithmic = %.4f of 8 TB/s;  NTT pair %.3f ms = %.4f of 8 TB/s" % (
        name, ks, " ".join("%.3f" % v for v in r["ks"]), algo / ks / 1e6, algo / ks / 1e6 / 8000, nt, 2 * 16 * (1 << logn) * dim * batch / nt / 1e6 / 8000))
print("same words under both assignments: yes")
