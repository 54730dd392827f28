"""gpq_set_nt_policy: the slab traffic of the transform kernels with the default cache policy (0), non-temporal (1) and chosen by the launch
group's working set (-1, the product's default), interleaved on ONE device: the RNS core of he_mul at the headline shape and at the
reference's default shape, a squaring, standalone NTT+INTT pairs at configs[1] (fits the Infinity Cache) and at n = 2^16 x 30 limbs (does
not), for several batch sizes around the threshold.  Same words in every mode (asserted on the first round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab

MODES = (0, 1, -1)


def timed(fn, iters):
    for _ in range(2):
        fn()
    t = gpqhe_amd.StreamTimer()
    t.start()
    for _ in range(iters):
        fn()
    t.stop()
    return t.elapsed_ms() / iters


def core_case(logn, dimA, dimB, batch, square=False):
    ctx = gpqhe_amd.PolyContext(logn, dimB)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    a = [rand_slab(torch, ctx, dimA, batch, gen) for _ in range(4)]
    if square:
        a[2], a[3] = a[0], a[1]
    x = rand_slab(torch, ctx, dimB, batch, gen)
    e = [rand_slab(torch, ctx, dimB, 1, gen) for _ in range(2)]
    d = [torch.empty_like(a[0]) for _ in range(3)]
    c = [torch.empty_like(x) for _ in range(2)]
    wsA, wsB = ctx.tensor_workspace(dimA, batch), ctx.keyswitch_workspace(dimB, batch)

    def step():
        ctx.he_mul_tensor(d[0], d[1], d[2], a[0], a[1], a[2], a[3], dimA, wsA)
        ctx.he_keyswitch(c[0], c[1], x, e[0], e[1], dimB, wsB)
    return ctx, step, lambda: [t.clone() for t in d + c], lambda ms: "%.0f he_mul/s" % (batch / ms * 1e3)


def ntt_case(logn, dim, batch):
    ctx = gpqhe_amd.PolyContext(logn, dim)
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    slab = rand_slab(torch, ctx, dim, batch, gen)

    def step():
        ctx.poly_ntt(slab, dim)
        ctx.poly_invntt(slab, dim)
    byts = 2 * 16 * (1 << logn) * dim * batch
    return ctx, step, lambda: [slab.clone()], lambda ms: "%.0f GB/s (%.3f of 8 TB/s)" % (byts / ms / 1e6, byts / ms / 8e9)


CASES = [
    ("RNS core, n=2^16 30/45 limbs, batch 64", lambda: core_case(16, 30, 45, 64), 6),
    ("RNS core, squaring, n=2^16 30/45 limbs, batch 64", lambda: core_case(16, 30, 45, 64, True), 6),
    ("RNS core, n=2^14 16/24 limbs (reference default), batch 64", lambda: core_case(14, 16, 24, 64), 25),
    ("RNS core, n=2^14 16/24 limbs, batch 8", lambda: core_case(14, 16, 24, 8), 100),
    ("NTT+INTT, n=2^15 10 limbs batch 64 (configs[1], 168 MB)", lambda: ntt_case(15, 10, 64), 40),
    ("NTT+INTT, n=2^15 10 limbs batch 96 (252 MB)", lambda: ntt_case(15, 10, 96), 30),
    ("NTT+INTT, n=2^15 10 limbs batch 128 (336 MB)", lambda: ntt_case(15, 10, 128), 30),
    ("NTT+INTT, n=2^15 10 limbs batch 256 (671 MB)", lambda: ntt_case(15, 10, 256), 20),
    ("NTT+INTT, n=2^16 30 limbs batch 64 (1 GB)", lambda: ntt_case(16, 30, 64), 20),
    ("NTT+INTT, n=2^14 24 limbs batch 64 (201 MB)", lambda: ntt_case(14, 24, 64), 40),
]

only = sys.argv[1:]
for name, make, iters in CASES:
    if only and not any(k in name for k in only):
        continue
    ctx, step, snap, fmt = make()
    want = None
    for rnd in range(3):
        parts = []
        for mode in MODES:
            ctx.set_nt_policy(mode)
            ms = timed(step, iters)
            if rnd == 0 and not name.startswith("NTT"):      # (the NTT case transforms in place: its words depend on the pair count, equal in every mode)
                got = snap()
                if want is None:
                    want = got
                assert all(torch.equal(u, v) for u, v in zip(want, got)), (name, mode)
            parts.append("policy %2d: %.4f ms %s" % (mode, ms, fmt(ms)))
        print("%s, round %d: %s" % (name, rnd, " | ".join(parts)), flush=True)
    ctx.close()
