"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: there is no GPU sanitizer on this
pool): tests/c/oracle_sanitized.c compiles oracle/gpqhe_oracle.c into one program with both sanitizers, replays the he_mul
RNS-core known-answer test and the single-limb NTT of SURVEY.md 8c, and must reproduce the reference's digests without a report."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    out = str(tmp_path_factory.mktemp("san") / "oracle_sanitized")
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-o", out, os.path.join(ROOT, "tests", "c", "oracle_sanitized.c"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed: " + r.stderr.splitlines()[-1])
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize("logn", ["7", "12"])
def test_oracle_reproduces_the_goldens_under_asan_and_ubsan(binary, logn):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_8c.json")))
    kat, ntt = golden["he_mul_core_kat"][logn], golden["ntt_kat_seed1_limb0"][logn]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([binary, logn, str(max(kat["dA"], kat["dB"])), str(kat["dA"]), str(kat["dB"])], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    got = dict(line.split() for line in r.stdout.splitlines() if len(line.split()) == 2)
    assert [got["in%d" % i] for i in range(4)] == kat["inputs"]
    for k in ("d0", "d1", "d2", "c0", "c1"):
        assert got[k] == kat[k], k
    assert got["ntt"] == ntt["ntt"] and got["roundtrip"] == "ok"
