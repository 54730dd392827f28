"""The link story INTEGRATION.md tells, tested where it matters: a GPQHE build has its own shared object with STRONG
definitions of polyctx, hectx, polyctx_init, hectx_init, poly_rns_alloc ... (src/Makefile:55-58); between two shared objects
ld.so takes the first definition in search order, weak or not.  libgpqhe_hip.so therefore defines none of those names (they live in
libgpqhe_hip_ctx.so, for hosts that are not GPQHE) and only reads `polyctx` / `hectx`: with a stand-in GPQHE library
(tests/c/link_fake.c) in front of the engine, behind it, and with the engine dlopen()ed RTLD_LOCAL, the program's calls reach
GPQHE's definitions and the engine is bound to the very same `polyctx` / `hectx` objects.  CPU tier: no device is touched."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.join(ROOT, "gpqhe_amd")
INC = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "tests", "c")


@pytest.fixture(scope="module")
def fake(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("link"))
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-fPIC", "-shared", "-I", INC, os.path.join(SRC, "link_fake.c"), "-o", os.path.join(d, "libfakegpqhe.so")])
    return d


def _run(exe, *args):
    res = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    return dict(line.split(None, 1) for line in res.stdout.strip().splitlines())


def _expect_gpqhe_everywhere(out):
    assert out["calls"] == "polyctx_init 1 hectx_init 1 poly_rns_alloc 1 poly_rns_free 1"          # the program's calls ran GPQHE's code
    assert out["functions"] == "polyctx_init libfakegpqhe.so hectx_init libfakegpqhe.so poly_rns_alloc libfakegpqhe.so"
    assert out["polyctx"] == "host 1 fake 1 engine 1 n 32"                                         # one object, seen by all three
    assert out["hectx"] == "host 1 fake 1 engine 1 slots 4"


@pytest.mark.parametrize("order", ["gpqhe_first", "engine_first"])
def test_both_link_orders_bind_to_gpqhes_own_definitions(fake, order):
    exe = os.path.join(fake, "host_" + order)
    libs = ["-lfakegpqhe", "-lgpqhe_hip"] if order == "gpqhe_first" else ["-lgpqhe_hip", "-lfakegpqhe"]
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-DLINKED", "-I", INC, os.path.join(SRC, "link_host.c"), "-L", fake, "-L", LIB_DIR] + libs +
                          ["-ldl", "-Wl,-rpath," + fake, "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    _expect_gpqhe_everywhere(_run(exe, "linked"))


def test_engine_opened_with_dlopen_local_sees_the_programs_context(fake):
    """The Python harness loads the engine like this (ctypes = RTLD_LOCAL).  The engine's references to polyctx / hectx are
    resolved in the global scope first: GPQHE's library, loaded with the program, provides them."""
    exe = os.path.join(fake, "host_dlopen")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-I", INC, os.path.join(SRC, "link_host.c"), "-L", fake, "-lfakegpqhe", "-ldl",
                           "-Wl,-rpath," + fake, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    _expect_gpqhe_everywhere(_run(exe, "dlopen", os.path.join(LIB_DIR, "libgpqhe_hip.so")))


def test_the_engine_library_defines_no_context_symbol():
    """nm -D: polyctx / hectx are undefined weak references of libgpqhe_hip.so, and the init / alloc names do not appear at all."""
    out = subprocess.run(["nm", "-D", os.path.join(LIB_DIR, "libgpqhe_hip.so")], capture_output=True, text=True, check=True).stdout
    syms = {ln.split()[-1]: ln.split()[-2] for ln in out.splitlines() if ln.split()}
    assert syms.get("polyctx") in ("w", "v") and syms.get("hectx") in ("w", "v")
    for name in ("polyctx_init", "polyctx_exit", "hectx_init", "hectx_exit", "poly_mpi_alloc", "poly_mpi_free", "poly_rns_alloc", "poly_rns_free", "GPQHE_TWO"):
        assert name not in syms, name
    ctx = subprocess.run(["nm", "-D", os.path.join(LIB_DIR, "libgpqhe_hip_ctx.so")], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in ctx.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "WVTDB"}
    assert {"polyctx", "hectx", "GPQHE_TWO", "polyctx_init", "hectx_init", "poly_mpi_alloc", "poly_rns_alloc"} <= defined
