"""Short runs of the two randomised parity soaks (tools/soak.py: the RNS kernels against the oracle; tools/soak_bridge.py: the bridge's callers on the
default kernels, lanes, fresh contexts, side streams and older kernel families against the separate kernels / the exact CRT).  The long runs are
under profiles/r04/v16_soak_*.txt; these keep the generators themselves alive and catch a regression in any of the randomised dimensions."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert "MISMATCH" not in r.stdout and "memory access fault" not in (r.stdout + r.stderr).lower(), tail
    return r.stdout


def test_bridge_soak_short():
    out = _run([os.path.join(ROOT, "tools", "soak_bridge.py"), "120", "5"], 600)
    assert "soak_bridge ok: 120 configurations" in out


def test_rns_soak_short():
    out = _run([os.path.join(ROOT, "tools", "soak.py"), "25", "3"], 600)
    assert "no mismatch" in out


def test_bridge_oracle_soak_short():
    """the bridge's callers against the restated reference (oracle/bigint_ref) on random small rings, moduli, levels, groups, lanes, forced exact paths"""
    out = _run([os.path.join(ROOT, "tools", "soak_bridge_oracle.py"), "40", "7"], 600)
    assert "soak_bridge_oracle ok: 40 configurations" in out
