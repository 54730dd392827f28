"""Whole-function expectations of the restated reference for DENSE operands, computed in worker processes.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): the checker of tests/ and of bench.py's check legs, never the thing measured.
One dense he_mul at n = 2^16, q = 2^850 costs ~22 s of Python integers + the C oracle (bigint_ref.he_mul restates src/he-mult.c:88-156),
so the ciphertexts a test wants checked go to worker processes started with `subprocess` (`python -m oracle.expect in out`: nothing of the
caller is inherited or re-imported -- no GPU state, no `__main__` -- and a worker imports neither torch nor the HIP library).  Words in, words out: big slabs `uint64[W][n]`, two's complement, as the C ABI takes them.
"""
import os as _os
import pickle as _pickle
import subprocess as _subprocess
import sys as _sys
import tempfile as _tempfile

import numpy as _np


def words_to_ints(words, W, n):
    """one big slab uint64[W][n] -> list of signed Python ints (vectorised by word: n = 2^17 in well under a second)"""
    a = _np.asarray(words, dtype=_np.uint64).reshape(W, n)
    vals = [0] * n
    for j in range(W):
        col = a[j].tolist()
        sh = 64 * j
        vals = [v | (c << sh) for v, c in zip(vals, col)]
    top, mod = 1 << (64 * W - 1), 1 << (64 * W)
    return [v - mod if v >= top else v for v in vals]


def ints_to_words(values, W):
    """list of signed ints -> uint64[W][n] (flattened), two's complement"""
    mod = 1 << (64 * W)
    vs = [int(v) % mod for v in values]
    out = _np.empty((W, len(vs)), dtype=_np.uint64)
    for j in range(W):
        sh = 64 * j
        out[j] = _np.array([(v >> sh) & 0xFFFFFFFFFFFFFFFF for v in vs], dtype=_np.uint64)
    return out.reshape(-1)


def _he_mul_task(t):
    """src/he-mult.c:88-156 (+ src/he-rescale.c:33-54 when t['rs'] = log2 Delta) on one ciphertext pair given as words."""
    from oracle import bigint_ref as ref
    from oracle.oracle import OracleCtx
    o = OracleCtx(t["logn"], t["dimB"])
    n, W, logq = o.n, t["W"], t["logq"]
    ct = [words_to_ints(w, W, n) for w in t["ct"]]
    e0, e1 = ref.he_mul(o, (ct[0], ct[1]), (ct[2], ct[3]), t["rlk0"], t["rlk1"], t["dimP"], t["dimA"], t["dimB"], logq)
    out = {"c0": ints_to_words(e0, W), "c1": ints_to_words(e1, W)}
    if t.get("rs"):
        s, ql = t["rs"], 1 << (logq - t["rs"])
        out["rs0"] = ints_to_words([ref.mpi_smod(ref.mpi_rdiv(v, 1 << s), ql) for v in e0], W)       # src/he-rescale.c:45-48
        out["rs1"] = ints_to_words([ref.mpi_smod(ref.mpi_rdiv(v, 1 << s), ql) for v in e1], W)
    return out


def _he_swk_task(t):
    """src/he-automorphism.c:40-85 on one (d0, d1) pair given as words."""
    from oracle import bigint_ref as ref
    from oracle.oracle import OracleCtx
    o = OracleCtx(t["logn"], t["dimB"])
    n, W = o.n, t["W"]
    e0, e1 = ref.he_swk(o, words_to_ints(t["d0"], W, n), words_to_ints(t["d1"], W, n), t["swk0"], t["swk1"], t["dimP"], t["dimB"], t["logq"])
    return {"c0": ints_to_words(e0, W), "c1": ints_to_words(e1, W)}


def _run(t):
    return _he_swk_task(t) if t["kind"] == "he_swk" else _he_mul_task(t)


def expect_many(tasks, workers=4, timeout=1500):
    """[task dict] -> [result dict], in order; at most `workers` child processes at a time (one task needs ~1.5 GB at n = 2^16)."""
    if not tasks:
        return []
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    results = [None] * len(tasks)
    with _tempfile.TemporaryDirectory() as td:
        pending, running = list(enumerate(tasks)), []
        try:
            while pending or running:
                while pending and len(running) < max(1, workers):
                    i, t = pending.pop(0)
                    fin, fout = _os.path.join(td, "in%d.pkl" % i), _os.path.join(td, "out%d.pkl" % i)
                    with open(fin, "wb") as f:
                        _pickle.dump(t, f, protocol=4)
                    running.append((i, fout, _subprocess.Popen([_sys.executable, "-m", "oracle.expect", fin, fout], cwd=root)))
                i, fout, proc = running.pop(0)
                if proc.wait(timeout=timeout) != 0:
                    raise RuntimeError("oracle.expect worker %d exited with %d" % (i, proc.returncode))
                with open(fout, "rb") as f:
                    results[i] = _pickle.load(f)
        finally:
            for _, _, proc in running:
                proc.kill()
    return results


if __name__ == "__main__":
    with open(_sys.argv[1], "rb") as _f:
        _task = _pickle.load(_f)
    with open(_sys.argv[2], "wb") as _f:
        _pickle.dump(_run(_task), _f, protocol=4)
