"""The reference's representation of zero in `ntt` output (src/ntt.c:45-48 keeps [0, p]: a sum leg x + t == p is stored as p)
through every door that hands forward-transform output to a caller: gpq_ntt, the reference-named `ntt` symbol from a C host,
the evaluation keys gpq_he_genswk stores (src/he-kem.c:103-110) -- each bit for bit against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from gpqhe_amd import ints_to_big, to_device, to_host
from oracle import bigint_ref as ref
from tests.zero_cases import slab_of_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("logn,dim", [(1, 2), (3, 3), (7, 5), (12, 2), (13, 3), (14, 2), (15, 2), (16, 3), (17, 2)])
def test_gpq_ntt_stores_p_where_the_reference_does(engine_ctx, oracle_ctx, logn, dim):
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    names, slab = slab_of_cases(o, dim, 100 + logn)
    want = o.ntt_slab(slab, dim)
    per = dim * o.n
    assert sum(int((want[k * per:(k + 1) * per].reshape(dim, -1)[d] == np.uint64(o.p[d])).sum()) for k in range(len(names)) for d in range(dim)) > 0
    dev = to_device(slab)
    g.poly_ntt(dev, dim)
    got = to_host(dev)
    for k, name in enumerate(names):
        assert np.array_equal(got[k * per:(k + 1) * per], want[k * per:(k + 1) * per]), name
    # the inverse of the reference's words is the input again (p is a legal input word of poly_rns_mul, not of gpq_invntt:
    # canonicalise as barrett_reduce would), and a later call on zero-free data finds the flags cleared
    canon = to_device(np.concatenate([got.reshape(-1, o.n)[i] % np.uint64(o.p[i % dim]) for i in range(len(names) * dim)]))
    g.poly_invntt(canon, dim)
    assert np.array_equal(to_host(canon), slab)
    rnd = np.concatenate([o.gen(5 + k, dim) for k in range(3)])
    dev = to_device(rnd)
    g.poly_ntt(dev, dim)
    assert np.array_equal(to_host(dev), o.ntt_slab(rnd, dim))


@pytest.mark.parametrize("logn,dim", [(13, 2), (16, 2)])
def test_only_the_limbs_with_a_zero_are_touched_by_the_redo(engine_ctx, oracle_ctx, logn, dim):
    """A batch in which one limb of one polynomial has a zero: every other limb must equal the plain kernels' output, that one the oracle's."""
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    n = o.n
    slab = np.concatenate([o.gen(40 + k, dim) for k in range(4)])
    t = o.ntt(slab[(2 * dim + 1) * n:(2 * dim + 2) * n], 1)
    t[6] = 0
    slab[(2 * dim + 1) * n:(2 * dim + 2) * n] = o.invntt(t, 1)
    want = o.ntt_slab(slab, dim)
    assert int(want[(2 * dim + 1) * n + 6]) == o.p[1]
    dev = to_device(slab)
    g.poly_ntt(dev, dim)
    assert np.array_equal(to_host(dev), want)


@pytest.mark.parametrize("logn,dim", [(1, 1), (4, 2), (10, 2), (13, 2), (16, 1), (17, 1)])
def test_reference_kernel_is_src_ntt_c_for_any_words(engine_ctx, oracle_ctx, logn, dim):
    """gpq_ntt_reference = src/ntt.c:37-73 as written: canonical data, the word p, words far outside [0, p] (unsigned wrap-around)."""
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    n = o.n
    rng = np.random.default_rng(7 * logn)
    canon = np.concatenate([o.gen(3, dim), o.gen(4, dim)])
    with_p = canon.copy()
    for d in range(dim):
        with_p[d * n + rng.integers(0, n, size=max(1, n // 8))] = o.p[d]
        with_p[(dim + d) * n:(dim + d + 1) * n] = o.p[d]                 # a limb of nothing but p
    garbage = rng.integers(0, 1 << 63, size=2 * dim * n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=2 * dim * n, dtype=np.uint64)
    garbage[:8] = np.uint64(0xFFFFFFFFFFFFFFFF)
    for data in (canon, with_p, garbage):
        for inverse in (False, True):
            dev = to_device(data)
            g.poly_ntt_reference(dev, dim, inverse=inverse)
            assert np.array_equal(to_host(dev), o.ntt_slab(data, dim, inverse=inverse)), (logn, inverse)


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dropin_zero") / "dropin_host")
    lib_dir = os.path.join(ROOT, "gpqhe_amd")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "dropin_host.c"), "-L", lib_dir, "-lgpqhe_hip",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


@pytest.mark.parametrize("logn", [7, 13, 16])
def test_reference_named_ntt_symbol_from_c(host_binary, oracle_ctx, tmp_path, logn):
    """`ntt(a, rns)` / `invntt(a, rns)` of include/gpqhe_hip_compat.h from a C host: zero cases, inputs that already hold p
    (the output of an earlier ntt), out-of-domain words."""
    o = oracle_ctx(logn, 1)
    n, p = o.n, o.p[0]
    names, slab = slab_of_cases(o, 1, 300 + logn)
    rng = np.random.default_rng(logn)
    twice = o.ntt_slab(slab, 1)                                            # ntt of ntt output: inputs holding p
    wild = rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    for tag, data in (("cases", slab), ("twice", twice), ("wild", wild)):
        for direction in ("fwd", "inv"):
            fin, fout = str(tmp_path / (tag + ".in")), str(tmp_path / (tag + direction + ".out"))
            data.tofile(fin)
            res = subprocess.run([host_binary, "xform", str(logn), "1", direction, fin, fout], capture_output=True, text=True, timeout=300)
            assert res.returncode == 0, res.stderr
            got = np.fromfile(fout, dtype=np.uint64)
            assert np.array_equal(got, o.ntt_slab(data, 1, inverse=(direction == "inv"))), (tag, direction)


@pytest.mark.parametrize("logn,logq", [(7, 120), (13, 438)])
def test_he_genswk_keys_hold_p_where_the_reference_stores_it(engine_ctx, oracle_ctx, logn, logq):
    """src/he-kem.c:103-110 stores ntt output: a key polynomial whose transform has a zero on limb 0 (e = X - x_i for an
    evaluation point x_i at a sum-leg position, p1 = 0, s = 0) must come out with p there, and the all-zero p1 as zeros."""
    import torch
    n = 1 << logn
    probe = engine_ctx(logn, 20)
    dimP, dimA, dimB, dimevk = probe.he_dims(logq, logq)
    g, o = engine_ctx(logn, dimevk), oracle_ctx(logn, dimevk)
    X = np.zeros(n, dtype=np.uint64); X[1] = 1
    points = o.ntt(X, 0)                                                   # the evaluation point of every output position (limb 0)
    pos = 4
    e = [0] * n
    e[0], e[1] = o.p[0] - int(points[pos]), 1
    zero = [0] * n
    P = ref.RnsBasis(o.p[:dimP]).P
    W = (P << logq).bit_length() // 64 + 1
    want0 = o.ntt_slab(np.array([v % o.p[d] for d in range(dimevk) for v in e], dtype=np.uint64), dimevk)
    assert int(want0[pos]) == o.p[0]
    dev = [to_device(ints_to_big(v, W)) for v in (zero, zero, e, zero)]   # p1, s, e, sp
    evk0 = torch.empty(dimevk * n, dtype=torch.int64, device="cuda")
    evk1 = torch.empty_like(evk0)
    g.he_genswk(evk0, evk1, *dev, W, dimP, logq, dimevk)
    assert np.array_equal(to_host(evk0), want0)
    assert not to_host(evk1).any()


@pytest.mark.parametrize("logn,dim", [(7, 3), (12, 2), (13, 3), (14, 2), (15, 2), (16, 3), (17, 2)])
def test_gpq_invntt_takes_the_words_gpq_ntt_hands_out(engine_ctx, oracle_ctx, logn, dim):
    """the library's own forward output (which may hold p for a residue 0, src/ntt.c:47) straight into gpq_invntt, as the
    reference's invntt accepts it (its domain is [0, p]) -- no canonicalisation in between -- and p-heavy extremes: a limb that is all p,
    every second word p, p at random places.  Expected: the oracle's inverse of the same residues."""
    o, g = oracle_ctx(logn, dim), engine_ctx(logn, dim)
    n, per = o.n, dim * o.n
    names, slab = slab_of_cases(o, dim, 900 + logn)
    fwd = to_device(slab)
    g.poly_ntt(fwd, dim)
    words = to_host(fwd).copy()
    assert any(int((words.reshape(-1, n)[i] == np.uint64(o.p[i % dim])).sum()) for i in range(len(names) * dim))
    g.poly_invntt(fwd, dim)                                           # the round trip through the reference's words is the identity
    assert np.array_equal(to_host(fwd), slab)
    rng = np.random.default_rng(logn)
    extremes = []
    for kind in range(3):
        poly = o.gen(70 + kind, dim).reshape(dim, n).copy()
        for d in range(dim):
            if kind == 0: poly[d, :] = np.uint64(o.p[d])
            elif kind == 1: poly[d, ::2] = np.uint64(o.p[d])
            else: poly[d, rng.random(n) < 0.1] = np.uint64(o.p[d])
        extremes.append(poly.reshape(-1))
    ext = np.concatenate(extremes)
    canon = np.concatenate([ext.reshape(-1, n)[i] % np.uint64(o.p[i % dim]) for i in range(3 * dim)])
    want = np.concatenate([o.invntt(canon[i * n:(i + 1) * n].copy(), i % dim) for i in range(3 * dim)])
    dev_p = to_device(ext)
    g.poly_invntt(dev_p, dim)
    assert np.array_equal(to_host(dev_p), want)                     # p and 0 are the same residue to the inverse transform
    assert per == dim * n
