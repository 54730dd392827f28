"""The host-only half of the MPI-typed surface under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (there is no GPU
sanitizer on this pool): tests/c/convert_sanitized.cpp compiles gpqhe_amd/csrc/mpi_convert.hpp -- the libgcrypt <-> big-slab
conversions of poly_mul / he_mul / he_rs (src/poly.h:86-87, src/gpqhe.h:136-147 carry MPI polynomials), their worker threads, and the
multiword helpers behind polyctx_init / hectx_init -- into one program with both sanitizers and drives it with real libgcrypt
integers: round trips over signs, zeros and every width up to the slab's (single- and multi-threaded), the word-major two's-
complement layout the device kernels read, and the helpers against libgcrypt's own arithmetic."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("san") / "convert_sanitized")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "convert_sanitized.cpp"), "-ldl", "-l:libgcrypt.so.20", "-pthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("sanitize" in r.stderr or "libgcrypt" in r.stderr):
        pytest.skip("sanitizer runtime or libgcrypt runtime not installed: " + r.stderr.splitlines()[-1])
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.timeout(300)
def test_conversions_and_context_helpers_under_asan_and_ubsan(binary):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([binary], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    word = r.stdout.split()
    assert word[0] == "ok" and int(word[1]) > 80000


@pytest.mark.timeout(400)
def test_conversion_threads_under_tsan(tmp_path):
    """The same program under ThreadSanitizer: the conversion pool (tasks that keep to their thread, the caller working beside its helpers)
    and the threaded conversions race-free."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    out = str(tmp_path / "convert_tsan")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "convert_sanitized.cpp"), "-ldl", "-l:libgcrypt.so.20", "-pthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("sanitize" in r.stderr or "tsan" in r.stderr or "libgcrypt" in r.stderr):
        pytest.skip("sanitizer runtime or libgcrypt runtime not installed: " + r.stderr.splitlines()[-1])
    assert r.returncode == 0, r.stderr
    r = subprocess.run([out], capture_output=True, text=True, timeout=380)
    if "FATAL: ThreadSanitizer" in r.stderr and "mmap" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory here: " + r.stderr.splitlines()[0])
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
    assert r.stdout.split()[0] == "ok"

