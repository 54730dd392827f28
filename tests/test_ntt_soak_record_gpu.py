"""The configuration of the one unexplained record, `profiles/r04/v16_soak_rns_long_the_one_record.txt` (round 4, 18:57):

    MISMATCH ntt ciphertext 2 {'logn': 16, 'dim': 1, 'batch': 3, 'chunk': 3, 'limb_block': 0, 'classes': (1, 2), 'nt_policy': 1,
                               'seeds': [843779584, 892266427, 40055201, 69458351, 222764240, 954993035, 482770568]}

rebuilt from the generator that made it (`tools/soak.py` at commit 999519a, `random.Random(29)`; the seeds it draws are asserted below), run as
that process ran it -- a fresh context as the first work of the test, every call on the null stream, one launch group, no peer lane -- and checked
word for word against the oracle AND through the zero watch's debug door (gpq_debug_zero_watch): which (polynomial, limb) units the forward
kernels flagged (src/ntt.c:45-48: the reference stores p for a sum x + t == p) and that the redo kernel cleared exactly those.

What replaying the generator establishes about the record (HISTORY.md, round 5): polynomial 1 is the only one with residues 0 in its transform
(a pair, at 58616 / 58617); polynomial 2 -- the one that mismatched -- has none, its flag is never set and the redo kernel never touches it.
The record is therefore NOT a lost zero flag."""
import random

import numpy as np
import pytest
import torch

import gpqhe_amd
from gpqhe_amd import to_device, to_host

pytestmark = pytest.mark.gpu

RECORD = dict(logn=16, dim=1, batch=3, chunk=3, limb_block=0, classes=(1, 2), nt_policy=1,
              seeds=[843779584, 892266427, 40055201, 69458351, 222764240, 954993035, 482770568])


def _replay(oracle_ctx):
    """tools/soak.py (999519a) from random.Random(29) up to the device calls; returns the configuration, the inputs and where the zeros went."""
    rng = random.Random(29)
    logn = rng.choice([13, 13, 14, 15, 16, 17])
    dim = rng.choice([1, 2, 3, 5, 7]) if logn < 17 else rng.choice([1, 2, 12, 40])
    batch = rng.choice([1, 2, 3, 5])
    chunk, lblock = rng.choice([1, 2, 3, 32]), rng.choice([0, 0, 1, 2, 3])
    classes = rng.choice([None, None, (0, 0), (0, 99), (1, 2), (2, 3)])
    o = oracle_ctx(logn, dim)
    nt = rng.choice([-1, 0, 1, 1])
    seeds = [rng.randrange(1 << 30) for _ in range(7)]
    cfg = dict(logn=logn, dim=dim, batch=batch, chunk=chunk, limb_block=lblock, classes=classes, nt_policy=nt, seeds=seeds)
    ins = [np.concatenate([o.gen(s + k, dim) for k in range(batch)]) for s in seeds[:5]]
    assert not rng.random() < 0.3                 # no "extremes" in this configuration
    assert rng.random() < 0.35                    # zero mode
    rng.random()
    nrng = np.random.default_rng(seeds[0])
    zeros = {}
    for k in range(batch):
        for d in range(dim):
            if rng.random() < 0.6:
                base = (k * dim + d) * o.n
                t = o.ntt(ins[0][base:base + o.n], d)
                how = rng.choice(["one", "pair", "block", "even", "scatter"])
                if how == "one":
                    j = rng.randrange(o.n); t[j] = 0
                elif how == "pair":
                    j = 2 * rng.randrange(o.n // 2); t[j] = t[j + 1] = 0
                elif how == "block":
                    j = 8 * rng.randrange(o.n // 8); t[j:j + 8] = 0
                elif how == "even":
                    j = -1; t[0::2] = 0
                else:
                    j = -2; t[nrng.integers(0, o.n, size=max(2, o.n // 64))] = 0
                zeros[(k, d)] = (how, j)
                ins[0][base:base + o.n] = o.invntt(t % np.uint64(o.p[d]), d)
    ev = [o.gen(seeds[5], dim), o.gen(seeds[6], dim)]
    return o, cfg, ins, ev, zeros


def test_the_recorded_configuration_words_and_flags(oracle_ctx):
    o, cfg, ins, ev, zeros = _replay(oracle_ctx)
    assert cfg == RECORD                                          # the generator reproduces the record's configuration and seeds
    assert zeros == {(1, 0): ("pair", 58616)}                     # ... and says where the zeros are: polynomial 1 only
    logn, dim, batch = cfg["logn"], cfg["dim"], cfg["batch"]
    per = dim * o.n
    want_ntt = o.ntt_slab(ins[0], dim)
    p0 = np.uint64(o.p[0])
    assert [int((want_ntt[k * per:(k + 1) * per] == p0).sum()) for k in range(batch)] == [0, 2, 0]
    assert int((want_ntt == 0).sum()) == 0

    torch.cuda.set_device(0)
    assert torch.cuda.current_stream().cuda_stream == 0           # the record's process ran on the null stream
    g = gpqhe_amd.PolyContext(logn, dim)                           # a fresh context, as in the first configuration of that process
    try:
        g.set_chunk(cfg["chunk"]); g.set_limb_block(cfg["limb_block"]); g.set_nt_policy(cfg["nt_policy"])
        g.set_limb_classes(*cfg["classes"])
        g.debug_zero_watch(True)
        dev = [to_device(v) for v in ins]
        f = dev[0].clone(); g.poly_ntt(f, dim)
        before, after = g.debug_zero_flags(batch * dim)           # (waits for the device)
        got_first = to_host(f)
        i = dev[1].clone(); g.poly_invntt(i, dim)
        d = [torch.empty_like(dev[0]) for _ in range(3)]
        g.he_mul_tensor(d[0], d[1], d[2], dev[0], dev[1], dev[2], dev[3], dim)
        c = [torch.empty_like(dev[0]) for _ in range(2)]
        g.he_keyswitch(c[0], c[1], dev[4], to_device(ev[0]), to_device(ev[1]), dim)
        pa, pb, pr = dev[0].clone(), dev[2].clone(), torch.empty_like(dev[0])
        g.poly_mul_rns(pr, pa, pb, dim)
        torch.cuda.synchronize()
        got = [to_host(t) for t in (f, i, d[0], d[1], d[2], c[0], c[1], pr)]
        # flags: set by the forward kernels for polynomial 1 alone, all cleared by the redo kernel
        assert before.tolist() == [0, 1, 0], before
        assert after.tolist() == [0, 0, 0], after
        # words: the transform right after its own call, and again after everything the soak queued behind it (nothing later may touch it)
        assert np.array_equal(got_first, want_ntt), np.flatnonzero(got_first != want_ntt)[:8]
        assert np.array_equal(got[0], got_first)
        for k in range(batch):
            sl = slice(k * per, (k + 1) * per)
            exp = [want_ntt[sl], o.ntt_slab(ins[1][sl].copy(), dim, inverse=True)]
            exp += list(o.he_mul_tensor(*[v[sl].copy() for v in ins[:4]], dim))
            exp += list(o.keyswitch(ins[4][sl].copy(), ev[0], ev[1], dim))
            exp.append(o.poly_mul_rns(ins[0][sl].copy(), ins[2][sl].copy(), dim))
            for name, a, b in zip(("ntt", "invntt", "d0", "d1", "d2", "c0", "c1", "polymul"), got, exp):
                assert np.array_equal(a[sl], b), (name, k, np.flatnonzero(a[sl] != b)[:8])
        # the device's own literal src/ntt.c:37-52 on the same input agrees as well (a third opinion beside the two-pass kernels and the oracle)
        r = dev[0].clone(); g.poly_ntt_reference(r, dim)
        assert np.array_equal(to_host(r), want_ntt)
        # round 6: the observation the record could not make -- every INPUT, as it sits on the device after all eight calls, is word for word its
        # host source (nothing was damaged on the way in, nothing wrote into an input); tools/soak.py prints the same at any mismatch
        for j, (t, h) in enumerate(zip(dev, ins)):
            assert np.array_equal(to_host(t), h), ("input %d changed on the device" % j, np.flatnonzero(to_host(t) != h)[:8])
    finally:
        g.close()


@pytest.mark.parametrize("logn,dim,batch", [(16, 1, 3), (13, 3, 5), (15, 2, 3), (17, 2, 3)])
def test_no_call_of_the_soak_sequence_writes_outside_its_operands(oracle_ctx, logn, dim, batch):
    """the polynomial that differed is the LAST 512 KiB of `f` and borders the clone gpq_invntt transforms in place -- can any of the
    later calls address below its base pointer or beyond its end?  Every operand of the soak's call sequence carved out of ONE allocation with a
    canary band on each side (the record's geometry first: odd polynomial count, one limb, nt 1): after the eight calls every band still holds its
    canary, every input still holds its host source, and the outputs are the oracle's."""
    o = oracle_ctx(logn, dim)
    n = o.n
    per = dim * n
    words = batch * per
    BAND = 4096                                                   # 32 KiB: keeps every operand 128-byte aligned
    CANARY = -0x0123456789abcdf                                   # int64 view of a word no residue can be
    names = ["in0", "in1", "in2", "in3", "in4", "f", "i", "d0", "d1", "d2", "c0", "c1", "pa", "pb", "pr"]
    torch.cuda.set_device(0)
    arena = torch.full((len(names) * (words + BAND) + BAND,), CANARY, dtype=torch.int64, device="cuda")
    view = {nm: arena[BAND + j * (words + BAND): BAND + j * (words + BAND) + words] for j, nm in enumerate(names)}
    ins = [np.concatenate([o.gen(900 + 10 * s + k, dim) for k in range(batch)]) for s in range(5)]
    ev = [o.gen(77, dim), o.gen(78, dim)]
    g = gpqhe_amd.PolyContext(logn, dim)
    try:
        g.set_chunk(3); g.set_nt_policy(1)
        for j in range(5):
            view["in%d" % j].copy_(to_device(ins[j]))
        evd = [to_device(ev[0]), to_device(ev[1])]
        view["f"].copy_(view["in0"]); g.poly_ntt(view["f"], dim)
        view["i"].copy_(view["in1"]); g.poly_invntt(view["i"], dim)
        g.he_mul_tensor(view["d0"], view["d1"], view["d2"], view["in0"], view["in1"], view["in2"], view["in3"], dim)
        g.he_keyswitch(view["c0"], view["c1"], view["in4"], evd[0], evd[1], dim)
        view["pa"].copy_(view["in0"]); view["pb"].copy_(view["in2"])
        g.poly_mul_rns(view["pr"], view["pa"], view["pb"], dim)
        torch.cuda.synchronize()
        host = to_host(arena).view(np.int64)
        for j in range(len(names) + 1):
            band = host[j * (words + BAND): j * (words + BAND) + BAND]
            assert (band == CANARY).all(), ("canary band %d (before %s) was written" % (j, names[j] if j < len(names) else "the end"),
                                            np.flatnonzero(band != CANARY)[:8])
        for j in range(5):
            assert np.array_equal(to_host(view["in%d" % j]), ins[j]), "input %d changed" % j
        assert np.array_equal(to_host(view["f"]), o.ntt_slab(ins[0], dim))
        assert np.array_equal(to_host(view["i"]), o.ntt_slab(ins[1], dim, inverse=True))
        for k in range(batch):
            sl = slice(k * per, (k + 1) * per)
            exp = list(o.he_mul_tensor(*[v[sl].copy() for v in ins[:4]], dim)) + list(o.keyswitch(ins[4][sl].copy(), ev[0], ev[1], dim))
            exp.append(o.poly_mul_rns(ins[0][sl].copy(), ins[2][sl].copy(), dim))
            for nm, b in zip(("d0", "d1", "d2", "c0", "c1", "pr"), exp):
                assert np.array_equal(to_host(view[nm])[sl], b), (nm, k)
    finally:
        g.close()


@pytest.mark.parametrize("logn,dim,batch", [(12, 2, 3), (13, 3, 5), (16, 3, 5), (17, 2, 3)])
def test_zero_flags_are_set_exactly_where_the_output_holds_a_residue_zero(oracle_ctx, logn, dim, batch):
    """The debug door over several (polynomial, limb) units and kernel geometries: the forward kernels flag exactly the units whose transform
    holds a residue 0 (where the reference stores 0 or p, src/ntt.c:45-48), the redo kernel clears exactly those, and an odd polynomial count
    exercises the partner-less workgroup row of contig_pass / contig_pass8."""
    o = oracle_ctx(logn, dim)
    n = o.n
    rng = random.Random(1000 * logn + dim)
    slab = np.concatenate([o.gen(70 + k, dim) for k in range(batch)])
    want_flags = np.zeros(batch * dim, dtype=np.uint32)
    for k in range(batch):
        for d in range(dim):
            if rng.random() < 0.5:
                base = (k * dim + d) * n
                t = o.ntt(slab[base:base + n], d)
                t[rng.randrange(n)] = 0
                slab[base:base + n] = o.invntt(t % np.uint64(o.p[d]), d)
                want_flags[k * dim + d] = 1
    assert 0 < want_flags.sum() < batch * dim
    want = o.ntt_slab(slab, dim)
    g = gpqhe_amd.PolyContext(logn, dim)
    try:
        g.debug_zero_watch(True)
        dev = to_device(slab)
        g.poly_ntt(dev, dim)
        before, after = g.debug_zero_flags(batch * dim)
        assert before.tolist() == want_flags.tolist()
        assert not after.any()
        assert np.array_equal(to_host(dev), want)
    finally:
        g.close()


def test_the_recorded_configuration_as_the_first_gpu_work_of_a_fresh_process():
    """The record was the FIRST configuration of its process: the same check once more in a child process that has done nothing on the GPU before
    (device out of idle, freshly mapped memory, first context).  One run per suite run -- not a loop."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle.oracle import OracleCtx\n"
            "import tests.test_ntt_soak_record_gpu as m\n"
            "cache = {}\n"
            "m.test_the_recorded_configuration_words_and_flags(lambda logn, dim: cache.setdefault((logn, dim), OracleCtx(logn, dim)))\n"
            "print('fresh process: words and flags ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "fresh process: words and flags ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
