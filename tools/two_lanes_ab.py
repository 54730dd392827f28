"""gpq_set_overlap off / on, interleaved on one device: whole he_mul, a squaring and he_swk on device slabs at the headline shape (n = 2^16,
q = 2^850, 64 ciphertexts = two launch groups) and at the reference's default shape (n = 2^14, q = 2^438), one context, one caller stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gpqhe_amd
from bench import rand_slab


def setup(logn, logq, batch, seed):
    probe = gpqhe_amd.PolyContext(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    probe.close()
    ctx = gpqhe_amd.PolyContext(logn, dimevk)
    W = (logq + 64) // 64
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    def centred():
        big = torch.randint(-(1 << 62), 1 << 62, (batch, W, ctx.n), dtype=torch.int64, device="cuda", generator=gen)
        big[:, W - 1] = torch.randint(-(1 << 16), 1 << 16, (batch, ctx.n), dtype=torch.int64, device="cuda", generator=gen)
        return big.reshape(-1).contiguous()
    cts = [centred() for _ in range(4)]
    rlk = rand_slab(torch, ctx, dimB, 1, gen), rand_slab(torch, ctx, dimB, 1, gen)
    outs = torch.empty_like(cts[0]), torch.empty_like(cts[0])
    return ctx, cts, rlk, outs, W, (dimA, dimB, dimP)


def timed(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "core":              # the RNS core (bench.py's step): tensor 30 limbs + key switch 45 limbs, n = 2^16
        for batch, iters in ((64, 6), (128, 3), (256, 2)):
            ctx = gpqhe_amd.PolyContext(16, 45)
            gen = torch.Generator(device="cuda"); gen.manual_seed(5)
            a = [rand_slab(torch, ctx, 30, batch, gen) for _ in range(4)]
            x = rand_slab(torch, ctx, 45, batch, gen)
            e = [rand_slab(torch, ctx, 45, 1, gen) for _ in range(2)]
            d = [torch.empty_like(a[0]) for _ in range(3)]
            c = [torch.empty_like(x) for _ in range(2)]
            wsA, wsB = ctx.tensor_workspace(30, batch), ctx.keyswitch_workspace(45, batch)
            def step():
                ctx.he_mul_tensor(d[0], d[1], d[2], a[0], a[1], a[2], a[3], 30, wsA)
                ctx.he_keyswitch(c[0], c[1], x, e[0], e[1], 45, wsB)
            for rnd in range(3):
                res = []
                for on in (0, 1):
                    ctx.set_overlap(on)
                    ms = timed(step, iters)
                    res.append("overlap %d: %.3f ms (%.0f he_mul/s)" % (on, ms, batch / ms * 1e3))
                print("RNS core batch %d, round %d: %s" % (batch, rnd, " | ".join(res)), flush=True)
            ctx.close()
        sys.exit(0)
    SWEEP = len(sys.argv) > 1 and sys.argv[1] == "chunks"       # launch-group sizes with two lanes (he_mul, headline shape)
    if SWEEP:
        ctx, cts, rlk, outs, W, (dA, dB, dP) = setup(16, 850, 64, 21)
        fn = lambda: ctx.he_mul(outs[0], outs[1], *cts, rlk[0], rlk[1], W, 850, dA, dB, dP)
        for rnd in range(3):
            res = []
            for on, chunk in ((0, 32), (1, 32), (1, 24), (1, 22), (1, 16), (1, 11), (1, 8), (0, 16)):
                ctx.set_overlap(on); ctx.set_chunk(chunk)
                ms = timed(fn, 6)
                res.append("overlap %d chunk %d: %.3f ms (%.0f/s)" % (on, chunk, ms, 64 / ms * 1e3))
            print("round %d: %s" % (rnd, " | ".join(res)), flush=True)
        ctx.close()
    for logn, logq, batch, iters in (() if SWEEP else ((16, 850, 64, 6), (16, 850, 128, 3), (14, 438, 64, 30))):
        ctx, cts, rlk, outs, W, (dA, dB, dP) = setup(logn, logq, batch, 21)
        kinds = {
            "he_mul": lambda: ctx.he_mul(outs[0], outs[1], *cts, rlk[0], rlk[1], W, logq, dA, dB, dP),
            "squaring": lambda: ctx.he_mul(outs[0], outs[1], cts[0], cts[1], cts[0], cts[1], rlk[0], rlk[1], W, logq, dA, dB, dP),
            "he_swk": lambda: ctx.he_swk(outs[0], outs[1], cts[0], cts[1], rlk[0], rlk[1], W, logq, dB, dP),
        }
        for name, fn in kinds.items():
            for rnd in range(3):
                res = []
                for on in (0, 1):
                    ctx.set_overlap(on)
                    ms = timed(fn, iters)
                    res.append("overlap %d: %.3f ms (%.0f/s)" % (on, ms, batch / ms * 1e3))
                print("n=2^%d q=2^%d batch %d, %s, round %d: %s" % (logn, logq, batch, name, rnd, " | ".join(res)), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
