import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
