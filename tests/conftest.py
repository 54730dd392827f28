import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--variant", default=None, help="path of another build of libgpqhe_hip.so to run the GPU tests through "
                     "(e.g. the UBSan-instrumented host side, tools/gpu_ubsan.sh); without it the tests load the product, whatever the environment says")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--variant"):
        from gpqhe_amd import _native
        _native.use_variant(config.getoption("--variant"))


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "survey_8c.json")) as f:
        return json.load(f)


_ORACLES = {}


@pytest.fixture(scope="session")
def oracle_ctx():
    """Session cache of oracle contexts keyed by (logn, nprimes)."""
    from oracle.oracle import OracleCtx

    def get(logn, nprimes):
        key = (logn, nprimes)
        if key not in _ORACLES:
            _ORACLES[key] = OracleCtx(logn, nprimes)
        return _ORACLES[key]

    return get


_ENGINES = {}


@pytest.fixture(scope="session")
def engine_ctx():
    """Session cache of device contexts keyed by (logn, nprimes)."""
    import gpqhe_amd

    def get(logn, nprimes):
        key = (logn, nprimes)
        if key not in _ENGINES:
            _ENGINES[key] = gpqhe_amd.PolyContext(logn, nprimes)
        return _ENGINES[key]

    return get
