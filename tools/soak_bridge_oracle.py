"""Randomised parity soak of the bridge's callers AGAINST THE RESTATED REFERENCE (dev tool, GPU box).

tools/soak_bridge.py compares the streaming bridge with the library's own separate kernels (fast, any ring); this one compares the default
library with oracle/bigint_ref -- src/he-mult.c:88-156, src/he-automorphism.c:40-85, src/he-rescale.c:33-54 restated with Python integers over the C
oracle's limb loops -- at rings small enough for Python integers (n = 64 ... 512, the streaming kernels' groups of 64 coefficients upward): random
moduli and levels (every limb count / word count the kernels are instantiated for), batches over several launch groups, one or two lanes, forced
exact paths, he_mul, gpq_he_mul_rs with a random Delta, he_swk, he_rs.

    python tools/soak_bridge_oracle.py [configurations] [seed] [share of two-pass rings, e.g. 0.5]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import gpqhe_amd  # noqa: E402
from gpqhe_amd import big_to_ints, ints_to_big, to_device, to_host  # noqa: E402
from oracle import bigint_ref as ref  # noqa: E402
from oracle.oracle import OracleCtx  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG_EVERY = 1.0 - float(sys.argv[3]) if len(sys.argv) > 3 else 1.0       # third argument: share of configurations on a two-pass ring (n = 2^13, 2^14), default none
torch.cuda.set_device(0)
ctxs, oracles = {}, {}


def pair(logn, nprimes):
    key = (logn, nprimes)
    if key not in ctxs:
        if len(ctxs) > 10:
            k = next(iter(ctxs))
            ctxs.pop(k).close(); oracles.pop(k)
        ctxs[key] = gpqhe_amd.PolyContext(logn, nprimes)
        oracles[key] = OracleCtx(logn, nprimes)
    return ctxs[key], oracles[key]


def centred(logq, n):
    half = 1 << (logq - 1)
    edge = [half - 1, -half, 0, 1, -1]
    return [edge[i] if i < len(edge) else rng.randrange(-half, half) for i in range(n)]


t0 = time.time()
for it in range(N):
    logn = rng.choice((6, 6, 7, 8, 9)) if rng.random() < BIG_EVERY else rng.choice((13, 13, 14))   # now and then a two-pass ring (seconds of Python integers per ciphertext)
    logqL = rng.randrange(70, 1001)
    logql = logqL if rng.random() < 0.5 else rng.randrange(61, logqL + 1)
    batch, chunk, force, lanes = rng.randrange(1, 6), rng.choice((1, 2, 32)), rng.choice((0, 0, 1, 5, 64)), rng.choice((0, 1))
    if logn >= 13:
        batch, logqL = min(batch, 2), min(logqL, 900 if logn == 13 else 600)
        logql = min(logql, logqL)
    probe, _ = pair(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logqL, logql)
    g, o = pair(logn, max(dimevk, 20))
    assert ref.he_dims(logn, o.p, logqL, logql) == (dimP, dimA, dimB, dimevk)
    n, W = g.n, logqL // 64 + 1
    k0, k1 = o.gen(rng.randrange(1 << 30), dimB), o.gen(rng.randrange(1 << 30), dimB)
    cts = [[centred(logql, n) for _ in range(batch)] for _ in range(4)]
    dev = [to_device(ints_to_big([v for poly in c for v in poly], W).reshape(W, batch, n).transpose(1, 0, 2).reshape(-1)) for c in cts]
    d0, d1 = to_device(k0), to_device(k1)
    ld = rng.randrange(1, min(63, logql - 8) + 1)
    g.set_chunk(chunk); g.set_overlap(lanes); g.debug_force_redo(force)
    o0, o1, r0, r1, s0, s1 = (torch.empty_like(dev[0]) for _ in range(6))
    g.he_mul(o0, o1, *dev, d0, d1, W, logql, dimA, dimB, dimP)
    g.he_mul_rs(r0, r1, *dev, d0, d1, W, logql, dimA, dimB, dimP, ld)
    g.he_swk(s0, s1, dev[0], dev[1], d0, d1, W, logql, dimB, dimP)
    torch.cuda.synchronize()
    g.set_chunk(32); g.set_overlap(-1); g.debug_force_redo(0)
    got = [big_to_ints(to_host(t), W, n) for t in (o0, o1, r0, r1, s0, s1)]
    for k in range(batch):
        e0, e1 = ref.he_mul(o, (cts[0][k], cts[1][k]), (cts[2][k], cts[3][k]), k0, k1, dimP, dimA, dimB, logql)
        rs = [[ref.mpi_smod(ref.mpi_rdiv(v, 1 << ld), 1 << (logql - ld)) for v in e] for e in (e0, e1)]
        w0, w1 = ref.he_swk(o, cts[0][k], cts[1][k], k0, k1, dimP, dimB, logql)
        for name, a, b in zip(("he_mul c0", "he_mul c1", "he_mul_rs c0", "he_mul_rs c1", "he_swk c0", "he_swk c1"), got, (e0, e1, rs[0], rs[1], w0, w1)):
            if a[k] != b:
                bad = [i for i in range(n) if a[k][i] != b[i]]
                print("MISMATCH %s ciphertext %d at config %d: logn %d q_L 2^%d q_l 2^%d dims P/A/B %d/%d/%d W %d batch %d chunk %d force %d lanes %d Delta 2^%d: %d coefficients, first %s"
                      % (name, k, it, logn, logqL, logql, dimP, dimA, dimB, W, batch, chunk, force, lanes + 1, ld, len(bad), bad[:6]), flush=True)
                sys.exit(1)
    if it % 10 == 0:
        print("config %d ok: logn %d q_L 2^%d q_l 2^%d dims P/A/B %d/%d/%d W %d batch %d chunk %d force %d lanes %d Delta 2^%d (%.0f s)"
              % (it, logn, logqL, logql, dimP, dimA, dimB, W, batch, chunk, force, lanes + 1, ld, time.time() - t0), flush=True)
print("soak_bridge_oracle ok: %d configurations, every coefficient of he_mul / he_mul_rs / he_swk equal to the restated reference (%.0f s)" % (N, time.time() - t0))
