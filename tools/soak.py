"""Randomised parity soak (dev tool, GPU box): random ring sizes, limb counts, batches, launch-group settings and butterfly-class
restrictions through gpq_ntt / gpq_invntt / gpq_he_mul_tensor / gpq_keyswitch / gpq_poly_mul_rns against the oracle, for SECONDS
seconds.  Prints every configuration it ran and stops at the first mismatch.

    python tools/soak.py [seconds] [seed]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import gpqhe_amd  # noqa: E402
from gpqhe_amd import to_device, to_host  # noqa: E402
from oracle.oracle import OracleCtx  # noqa: E402


def diagnose(name, k, got, exp, sl, tensor, rerun, p):
    """Everything a later reader needs from a mismatch (round 4's record held one line): which words, what they hold, whether a second download
    of the same tensor still differs (transfer or device memory) and whether the same call on the same inputs differs again (deterministic or
    transient)."""
    a, b = got[sl], exp
    bad = np.flatnonzero(a != b)
    print("  differing words: %d of %d; first %s" % (bad.size, a.size, bad[:8].tolist()), flush=True)
    for j in bad[:8]:
        print("    [%d] got %d expected %d%s" % (j, int(a[j]), int(b[j]), "  (got 0)" if a[j] == 0 else "  (got a prime)" if int(a[j]) in p else ""), flush=True)
    if bad.size:
        runs = np.split(bad, np.flatnonzero(np.diff(bad) != 1) + 1)
        print("  contiguous runs: %d, longest %d, span [%d, %d]" % (len(runs), max(len(r) for r in runs), int(bad[0]), int(bad[-1])), flush=True)
    again = to_host(tensor)[sl]
    print("  second download of the same tensor: %s" % ("equal to the first" if np.array_equal(again, a) else
                                                         "DIFFERENT (%d words; now %s the expectation)" % (int((again != a).sum()), "equal to" if np.array_equal(again, b) else "still unlike")), flush=True)
    fresh = to_host(rerun())[sl]
    print("  same call again on the same inputs: %s" % ("bit-exact this time (transient)" if np.array_equal(fresh, b) else
                                                         "differs again (%d words, %s)" % (int((fresh != b).sum()), "the same words" if np.array_equal(fresh, a) else "other words")), flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    torch.cuda.set_device(0)
    oracles, t0, runs = {}, time.time(), 0
    while time.time() - t0 < seconds:
        logn = rng.choice([13, 13, 14, 15, 16, 17])
        dim = rng.choice([1, 2, 3, 5, 7]) if logn < 17 else rng.choice([1, 2, 12, 40])       # n = 2^17: 40 limbs cross the three butterfly classes
        batch = rng.choice([1, 2, 3, 5])
        chunk, lblock = rng.choice([1, 2, 3, 32]), rng.choice([0, 0, 1, 2, 3])
        classes = rng.choice([None, None, (0, 0), (0, 99), (1, 2), (2, 3)])
        key = (logn, dim)
        if key not in oracles:
            if len(oracles) > 6:
                oracles.pop(next(iter(oracles)))
            oracles[key] = OracleCtx(logn, dim)
        o = oracles[key]
        g = gpqhe_amd.PolyContext(logn, dim)
        g.set_chunk(chunk)
        g.set_limb_block(lblock)
        nt = rng.choice([-1, 0, 1, 1])                           # cache policy of the slab traffic: the NT instantiations of every kernel / class
        g.set_nt_policy(nt)
        if classes:
            g.set_limb_classes(*classes)
        per = dim * o.n
        seeds = [rng.randrange(1 << 30) for _ in range(7)]
        ins = [np.concatenate([o.gen(s + k, dim) for k in range(batch)]) for s in seeds[:5]]
        if rng.random() < 0.3:                                   # extremes: all p-1 / all zero limbs
            for d in range(dim):
                ins[0][d * o.n:(d + 1) * o.n] = o.p[d] - 1
                ins[1][d * o.n:(d + 1) * o.n] = 0
        zero_mode = rng.random() < 0.35                          # the reference's representation of zero (src/ntt.c:47): transforms with residues 0
        if zero_mode and not (rng.random() < 0.0):
            nrng = np.random.default_rng(seeds[0])
            for k in range(batch):
                for d in range(dim):
                    if rng.random() < 0.6:
                        base = (k * dim + d) * o.n
                        t = o.ntt(ins[0][base:base + o.n], d)
                        how = rng.choice(["one", "pair", "block", "even", "scatter"])
                        if how == "one":
                            t[rng.randrange(o.n)] = 0
                        elif how == "pair":
                            j = 2 * rng.randrange(o.n // 2); t[j] = t[j + 1] = 0
                        elif how == "block":
                            j = 8 * rng.randrange(o.n // 8); t[j:j + 8] = 0
                        elif how == "even":
                            t[0::2] = 0
                        else:
                            t[nrng.integers(0, o.n, size=max(2, o.n // 64))] = 0
                        ins[0][base:base + o.n] = o.invntt(t % np.uint64(o.p[d]), d)
        ev = [o.gen(seeds[5], dim), o.gen(seeds[6], dim)]
        dev = [to_device(v) for v in ins]
        evd = [to_device(ev[0]), to_device(ev[1])]
        lanes = rng.choice([0, 1, 1])                            # two launch groups in flight (gpq_set_overlap) where a call has more than one
        g.set_overlap(lanes)
        side = torch.cuda.Stream() if rng.random() < 0.4 else None   # a non-blocking side stream: nothing may lean on the null stream's ordering
        torch.cuda.synchronize()
        import contextlib
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            f = dev[0].clone(); g.poly_ntt(f, dim)
            i = dev[1].clone(); g.poly_invntt(i, dim)
            d = [torch.empty_like(dev[0]) for _ in range(3)]
            g.he_mul_tensor(d[0], d[1], d[2], dev[0], dev[1], dev[2], dev[3], dim)
            c = [torch.empty_like(dev[0]) for _ in range(2)]
            g.he_keyswitch(c[0], c[1], dev[4], evd[0], evd[1], dim)
            pa, pb, pr = dev[0].clone(), dev[2].clone(), torch.empty_like(dev[0])
            g.poly_mul_rns(pr, pa, pb, dim)
        torch.cuda.synchronize()
        got = [to_host(t) for t in (f, i, d[0], d[1], d[2], c[0], c[1], pr)]
        for k in range(batch):
            sl = slice(k * per, (k + 1) * per)
            exp = [o.ntt_slab(ins[0][sl].copy(), dim), o.ntt_slab(ins[1][sl].copy(), dim, inverse=True)]
            exp += list(o.he_mul_tensor(*[v[sl].copy() for v in ins[:4]], dim))
            exp += list(o.keyswitch(ins[4][sl].copy(), ev[0], ev[1], dim))
            exp.append(o.poly_mul_rns(ins[0][sl].copy(), ins[2][sl].copy(), dim))
            for name, a, b in zip(("ntt", "invntt", "d0", "d1", "d2", "c0", "c1", "polymul"), got, exp):
                if not np.array_equal(a[sl], b):
                    def rerun(name=name):
                        if name == "ntt":
                            t = dev[0].clone(); g.poly_ntt(t, dim); return t
                        if name == "invntt":
                            t = dev[1].clone(); g.poly_invntt(t, dim); return t
                        if name in ("d0", "d1", "d2"):
                            t = [torch.empty_like(dev[0]) for _ in range(3)]
                            g.he_mul_tensor(t[0], t[1], t[2], dev[0], dev[1], dev[2], dev[3], dim); return t[int(name[1])]
                        if name in ("c0", "c1"):
                            t = [torch.empty_like(dev[0]) for _ in range(2)]
                            g.he_keyswitch(t[0], t[1], dev[4], evd[0], evd[1], dim); return t[int(name[1])]
                        t = torch.empty_like(dev[0]); g.poly_mul_rns(t, dev[0].clone(), dev[2].clone(), dim); return t
                    print("MISMATCH", name, "ciphertext", k, dict(logn=logn, dim=dim, batch=batch, chunk=chunk, limb_block=lblock, classes=classes, nt_policy=nt, lanes=lanes + 1, side_stream=side is not None, seeds=seeds), flush=True)
                    diagnose(name, k, a, b, sl, dict(zip(("ntt", "invntt", "d0", "d1", "d2", "c0", "c1", "polymul"), (f, i, d[0], d[1], d[2], c[0], c[1], pr)))[name], rerun, set(o.p))
                    sys.exit(1)
        runs += 1
        print("ok", dict(logn=logn, dim=dim, batch=batch, chunk=chunk, limb_block=lblock, classes=classes, nt_policy=nt, lanes=lanes + 1, side_stream=side is not None, zero_cases=zero_mode), flush=True)
        g.close()
    print("soak: %d configurations, no mismatch, %.0f s" % (runs, time.time() - t0))


if __name__ == "__main__":
    main()
