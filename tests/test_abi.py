"""The C-ABI library loads and exports every symbol include/*.h declares.
No compute calls here (no GPU in this tier)."""
import os
import re

import gpqhe_amd
from gpqhe_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text):
        names.add(m.group(1))
    return names - {"defined"}


def test_library_loads_and_exports_all_declared_symbols():
    lib = gpqhe_amd.load()
    declared = _declared("gpqhe_hip.h") | _declared("gpqhe_hip_compat.h")
    assert {"gpq_ctx_create", "gpq_ntt", "gpq_he_mul_tensor", "ntt", "invntt", "poly_rns_mul", "poly_ntt",
            "montgomery_reduce", "barrett_reduce"} <= declared
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, "declared in include/ but not exported: %s" % missing
    bound = set(_native.SIGNATURES) | set(_native.EXPORTED_ONLY)
    assert declared <= bound, "not bound in gpqhe_amd/_native.py: %s" % sorted(declared - bound)
    # the context / storage names of the reference live in libgpqhe_hip_ctx.so and nowhere else: the engine library must not
    # be able to shadow GPQHE's own precomp.o / poly.o, whatever the link order (tests/test_link_order.py)
    import ctypes
    ctx_declared = _declared("gpqhe_hip_ctx.h") | {"polyctx", "hectx", "GPQHE_TWO"}
    assert ctx_declared == set(_native.CTX_EXPORTS)
    assert not [n for n in sorted(ctx_declared) if hasattr(lib, n)], "libgpqhe_hip.so defines a context symbol"
    ctx = ctypes.CDLL(_native.CTX_LIB_PATH)
    assert not [n for n in sorted(ctx_declared) if not hasattr(ctx, n)]


def test_scalar_helpers_match_reference_semantics(golden):
    lib = gpqhe_amd.load()
    for logn, k in golden["p0_constants"].items():
        p0 = int(golden["prime_chain"][logn]["first"][0])
        assert str(lib.montgomery_inv(p0)) == k["pinv_mont"]   # src/reduce.c:36-48
        assert str(lib.barrett_inv(p0)) == k["pinv_barr"]      # src/reduce.c:75-78
    assert lib.gpq_dimub(16, 850) == 58 and lib.gpq_dimub(7, 61) == 5  # src/precomp.c:357


def test_no_cpu_fallback_when_library_missing(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libgpqhe_hip.so")
    try:
        _native.load()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must fail loudly without the HIP library")


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gpqhe_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "gpqhe_oracle" not in src, f
