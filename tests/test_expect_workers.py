"""oracle/expect.py (the worker-process front of the restated whole functions, used by the bench-shape parity tests and by bench.py's check of
its own legs): words in / words out must equal bigint_ref called directly, for he_mul (+ he_rs) and he_swk, through real child processes."""
import random

import numpy as np

from oracle import bigint_ref as ref
from oracle import expect


def test_workers_equal_the_direct_call(oracle_ctx):
    logn, logq = 7, 61
    o = oracle_ctx(logn, 8)
    n = o.n
    dimP, dimA, dimB, dimevk = ref.he_dims(logn, o.p, logq, logq)
    rng = random.Random(1)
    q = 1 << logq
    ct = [[rng.randrange(-(q >> 1), q >> 1) for _ in range(n)] for _ in range(4)]
    for c in ct:
        c[:3] = [-(q >> 1), (q >> 1) - 1, -1]
    W = logq // 64 + 1
    rlk0, rlk1 = o.gen(3000, dimB), o.gen(3001, dimB)
    e0, e1 = ref.he_mul(o, (ct[0], ct[1]), (ct[2], ct[3]), rlk0, rlk1, dimP, dimA, dimB, logq)
    s0, s1 = ref.he_swk(o, ct[0], ct[1], rlk0, rlk1, dimP, dimB, logq)
    words = [expect.ints_to_words(c, W) for c in ct]
    assert expect.words_to_ints(words[0], W, n) == ct[0]
    tasks = [dict(kind="he_mul", logn=logn, dimP=dimP, dimA=dimA, dimB=dimB, W=W, logq=logq, ct=words, rlk0=rlk0, rlk1=rlk1, rs=20),
             dict(kind="he_swk", logn=logn, dimP=dimP, dimB=dimB, W=W, logq=logq, d0=words[0], d1=words[1], swk0=rlk0, swk1=rlk1)]
    r = expect.expect_many(tasks, workers=2)
    assert np.array_equal(r[0]["c0"], expect.ints_to_words(e0, W)) and np.array_equal(r[0]["c1"], expect.ints_to_words(e1, W))
    assert expect.words_to_ints(r[0]["rs0"], W, n) == [ref.mpi_smod(ref.mpi_rdiv(v, 1 << 20), 1 << 41) for v in e0]      # src/he-rescale.c:45-48
    assert expect.words_to_ints(r[0]["rs1"], W, n) == [ref.mpi_smod(ref.mpi_rdiv(v, 1 << 20), 1 << 41) for v in e1]
    assert np.array_equal(r[1]["c0"], expect.ints_to_words(s0, W)) and np.array_equal(r[1]["c1"], expect.ints_to_words(s1, W))
    assert expect.expect_many([]) == []
