"""The reference-named link-time symbols, driven from a C host exactly as GPQHE's
own poly_mul limb loop drives them (tests/c/dropin_host.c), checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import fnv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dropin") / "dropin_host")
    lib_dir = os.path.join(ROOT, "gpqhe_amd")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "dropin_host.c"), "-L", lib_dir, "-lgpqhe_hip",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


@pytest.mark.parametrize("logn,dim", [(7, 5), (13, 2), (16, 2)])
def test_c_host_limb_loop(host_binary, oracle_ctx, logn, dim):
    seed = 77
    res = subprocess.run([host_binary, str(logn), str(dim), str(seed)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = dict(line.split(None, 1) for line in res.stdout.strip().splitlines())
    o = oracle_ctx(logn, dim)
    n = o.n
    a, b = o.gen(seed, dim), o.gen(seed + 1, dim)
    ah, bh = o.ntt_slab(a, dim), o.ntt_slab(b, dim)
    mul = np.concatenate([o.rns_mul(ah[d * n:(d + 1) * n], bh[d * n:(d + 1) * n], d) for d in range(dim)])
    add = np.concatenate([o.rns_add(ah[d * n:(d + 1) * n], bh[d * n:(d + 1) * n], d) for d in range(dim)])
    r = o.ntt_slab(mul, dim, inverse=True)
    assert got["ntt_b"] == fnv(bh)
    assert got["mul"] == fnv(r) and got["alias"] == fnv(r)
    assert got["add"] == fnv(add)
    assert np.array_equal(r, o.poly_mul_rns(a, b, dim))
    p0 = o.p[0]
    prod = (p0 - 1) * (p0 - 2)
    assert got["barrett"].split()[0] == str(prod % p0)
    assert got["barrett"].split()[2] == str(prod * pow(1 << 64, -1, p0) % p0)
