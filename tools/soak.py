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


def diagnose(name, k, got, exp, sl, tensor, rerun, p, inputs=(), third=None, oracle_again=None, out=print):
    """Everything a later reader needs from a mismatch (round 4's record held one line).  Observations, in the order that splits the candidate
    causes of the one open record (HISTORY.md R5.1 / R6.1):

      which words, what they hold; a SECOND download of the same tensor (transfer back vs device memory); the INPUT tensors of the call downloaded
      and compared word for word with their host sources (upload / first-touch damage, or something wrote into an input, vs a kernel fault); the
      device's literal src/ntt.c (`third`, ntt / invntt only) on the device-resident input as a third opinion beside the fast path and the oracle;
      the oracle run once more on the same host input (is the checker itself deterministic); the same call again on the same inputs (deterministic
      vs transient).

    `inputs` = [(label, device tensor, host array)], `third` = callable -> device tensor, `oracle_again` = callable -> host array (this slice's
    expectation recomputed).  Returns the set of verdict words the report ends with (tests/test_soak_diagnose.py feeds it every case)."""
    a, b = got[sl], exp
    bad = np.flatnonzero(a != b)
    out("  differing words: %d of %d; first %s" % (bad.size, a.size, bad[:8].tolist()))
    for j in bad[:8]:
        out("    [%d] got %d expected %d%s" % (j, int(a[j]), int(b[j]), "  (got 0)" if a[j] == 0 else "  (got a prime)" if int(a[j]) in p else ""))
    if bad.size:
        runs = np.split(bad, np.flatnonzero(np.diff(bad) != 1) + 1)
        out("  contiguous runs: %d, longest %d, span [%d, %d]" % (len(runs), max(len(r) for r in runs), int(bad[0]), int(bad[-1])))
    verdict = set()
    again = to_host(tensor)[sl]
    if np.array_equal(again, a):
        out("  second download of the same tensor: equal to the first")
    else:
        out("  second download of the same tensor: DIFFERENT (%d words; now %s the expectation)" % (int((again != a).sum()), "equal to" if np.array_equal(again, b) else "still unlike"))
        verdict.add("download-transient" if np.array_equal(again, b) else "device-memory-changing")
    # (i) the inputs as they sit on the device NOW against the host arrays they were uploaded from
    for label, dev_t, host in inputs:
        on_dev = to_host(dev_t)
        host = np.ascontiguousarray(host, dtype=np.uint64)
        if on_dev.shape == host.shape and np.array_equal(on_dev, host):
            out("  input %s on the device: equal to its host source (%d words)" % (label, host.size))
        else:
            diff = np.flatnonzero(on_dev != host) if on_dev.shape == host.shape else np.arange(0)
            inside = diff[(diff >= sl.start) & (diff < sl.stop)] if diff.size else diff
            out("  input %s on the device: DIFFERS from its host source (%d words, first %s; %d of them inside this ciphertext's slice)"
                % (label, diff.size, diff[:8].tolist(), inside.size))
            verdict.add("input-damaged")
    # (ii) the device's own literal src/ntt.c on the device-resident input
    if third is not None:
        t3 = to_host(third())[sl]
        if np.array_equal(t3, b):
            out("  device src/ntt.c as written (gpq_ntt_reference) on the device-resident input: equal to the oracle -> the fast path's output is the odd one")
            verdict.add("third-opinion-with-oracle")
        elif np.array_equal(t3, a):
            out("  device src/ntt.c as written (gpq_ntt_reference) on the device-resident input: equal to the FAST PATH -> the device agrees with itself, "
                "input or oracle is the odd one")
            verdict.add("third-opinion-with-fast-path")
        else:
            out("  device src/ntt.c as written (gpq_ntt_reference): unlike both (%d words off the oracle, %d off the fast path)" % (int((t3 != b).sum()), int((t3 != a).sum())))
            verdict.add("third-opinion-alone")
    # (iii) the checker once more
    if oracle_again is not None:
        b2 = oracle_again()
        if np.array_equal(b2, b):
            out("  oracle run again on the same host input: equal to its first answer")
        else:
            out("  oracle run again on the same host input: DIFFERENT from its first answer (%d words; now %s the device)" % (int((b2 != b).sum()), "equal to" if np.array_equal(b2, a) else "still unlike"))
            verdict.add("oracle-transient")
    fresh = to_host(rerun())[sl]
    if np.array_equal(fresh, b):
        out("  same call again on the same inputs: bit-exact this time (transient)")
        verdict.add("call-transient")
    else:
        out("  same call again on the same inputs: differs again (%d words, %s)" % (int((fresh != b).sum()), "the same words" if np.array_equal(fresh, a) else "other words"))
        verdict.add("call-deterministic")
    # what the combination proves (DESIGN.md section 2 states the three cases)
    if "input-damaged" in verdict:
        reading = "the data was wrong BEFORE the kernels ran (upload / first touch) or an input was overwritten later: not a transform fault"
    elif "oracle-transient" in verdict:
        reading = "the CHECKER changed its answer: host-side fault (memory, the oracle library), not the device"
    elif "download-transient" in verdict:
        reading = "device memory held the right words; the first copy back to the host was wrong"
    elif "call-transient" in verdict:
        reading = "inputs intact on the device, output wrong in device memory, the same launches right the second time: a transient fault of the " \
                  "kernels' stores / ordering (or a later writer), not of their arithmetic"
    else:
        reading = "inputs intact, output wrong again: a deterministic kernel bug -- this configuration is a regression test now"
    out("  reading: " + reading)
    return verdict


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
                    outs = dict(zip(("ntt", "invntt", "d0", "d1", "d2", "c0", "c1", "polymul"), (f, i, d[0], d[1], d[2], c[0], c[1], pr)))
                    feeds = {"ntt": [0], "invntt": [1], "d0": [0, 1, 2, 3], "d1": [0, 1, 2, 3], "d2": [0, 1, 2, 3], "c0": [4], "c1": [4], "polymul": [0, 2]}[name]
                    inputs = [("ins[%d]" % j, dev[j], ins[j]) for j in feeds]
                    if name in ("c0", "c1"):
                        inputs += [("evk0", evd[0], ev[0]), ("evk1", evd[1], ev[1])]
                    third = None
                    if name in ("ntt", "invntt"):
                        def third(name=name):
                            t = dev[0 if name == "ntt" else 1].clone(); g.poly_ntt_reference(t, dim, inverse=(name == "invntt")); return t
                    want = ("ntt", "invntt", "d0", "d1", "d2", "c0", "c1", "polymul").index(name)

                    def oracle_again(want=want, sl=sl):
                        e = [o.ntt_slab(ins[0][sl].copy(), dim), o.ntt_slab(ins[1][sl].copy(), dim, inverse=True)]
                        e += list(o.he_mul_tensor(*[v[sl].copy() for v in ins[:4]], dim))
                        e += list(o.keyswitch(ins[4][sl].copy(), ev[0], ev[1], dim))
                        e.append(o.poly_mul_rns(ins[0][sl].copy(), ins[2][sl].copy(), dim))
                        return e[want]
                    diagnose(name, k, a, b, sl, outs[name], rerun, set(o.p), inputs=inputs, third=third, oracle_again=oracle_again,
                             out=lambda line: print(line, flush=True))
                    sys.exit(1)
        runs += 1
        print("ok", dict(logn=logn, dim=dim, batch=batch, chunk=chunk, limb_block=lblock, classes=classes, nt_policy=nt, lanes=lanes + 1, side_stream=side is not None, zero_cases=zero_mode), flush=True)
        g.close()
    print("soak: %d configurations, no mismatch, %.0f s" % (runs, time.time() - t0))


if __name__ == "__main__":
    main()
