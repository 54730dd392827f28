"""The mismatch handler of tools/soak.py, fed every case it has to tell apart (CPU; torch CPU tensors stand for the device's).

The one open parity record (profiles/r04/v16_soak_rns_long_the_one_record.txt, HISTORY.md R5.1 / R6.1) was one line: it could not separate "the kernels wrote wrong
words" from "the input was already wrong on the device" from "the copy back was wrong" from "the checker was wrong".  A recurrence now prints the
observation that splits them; this test is the proof that the handler draws the right conclusion from each combination."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def soak():
    spec = importlib.util.spec_from_file_location("soak_tool", os.path.join(ROOT, "tools", "soak.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64).copy())


class _Case:
    """three polynomials of 64 words; polynomial 2 (the record's) is the one that differs"""

    def __init__(self):
        rng = np.random.default_rng(6)
        self.n, self.k = 64, 2
        self.host_in = rng.integers(1, 1 << 59, size=3 * self.n, dtype=np.uint64)
        self.exp_all = self.host_in * np.uint64(3) + np.uint64(1)          # stands for the transform
        self.sl = slice(self.k * self.n, (self.k + 1) * self.n)
        self.exp = self.exp_all[self.sl].copy()
        self.bad = self.exp_all.copy()
        self.bad[self.sl.start + 5:self.sl.start + 9] ^= np.uint64(0x10)


def _run(soak, c, *, got, tensor, dev_in, rerun, third, oracle_again):
    lines = []
    v = soak.diagnose("ntt", c.k, got, c.exp, c.sl, tensor, rerun, {7}, inputs=[("ins[0]", dev_in, c.host_in)], third=third,
                      oracle_again=oracle_again, out=lines.append)
    return v, "\n".join(lines)


def test_kernel_transient_is_told_from_everything_else(soak):
    c = _Case()
    v, text = _run(soak, c, got=c.bad, tensor=_t(c.bad), dev_in=_t(c.host_in), rerun=lambda: _t(c.exp_all),
                   third=lambda: _t(c.exp_all), oracle_again=lambda: c.exp.copy())
    assert v == {"third-opinion-with-oracle", "call-transient"}
    assert "differing words: 4 of 64; first [5, 6, 7, 8]" in text
    assert "contiguous runs: 1, longest 4, span [5, 8]" in text
    assert "input ins[0] on the device: equal to its host source (192 words)" in text
    assert "equal to the oracle" in text and "transient fault of the kernels' stores" in text


def test_deterministic_kernel_bug(soak):
    c = _Case()
    v, text = _run(soak, c, got=c.bad, tensor=_t(c.bad), dev_in=_t(c.host_in), rerun=lambda: _t(c.bad),
                   third=lambda: _t(c.exp_all), oracle_again=lambda: c.exp.copy())
    assert v == {"third-opinion-with-oracle", "call-deterministic"}
    assert "differs again (4 words, the same words)" in text and "deterministic kernel bug" in text


def test_damaged_input_on_the_device(soak):
    """upload / first-touch damage: the device transformed what it was given -- its literal src/ntt.c agrees with the fast path, not with the oracle"""
    c = _Case()
    dev_in = c.host_in.copy()
    dev_in[c.sl.start + 3] ^= np.uint64(1)
    v, text = _run(soak, c, got=c.bad, tensor=_t(c.bad), dev_in=_t(dev_in), rerun=lambda: _t(c.bad),
                   third=lambda: _t(c.bad), oracle_again=lambda: c.exp.copy())
    assert "input-damaged" in v and "third-opinion-with-fast-path" in v
    assert "DIFFERS from its host source (1 words, first [131]; 1 of them inside this ciphertext's slice)" in text
    assert "wrong BEFORE the kernels ran" in text


def test_wrong_copy_back(soak):
    c = _Case()
    v, text = _run(soak, c, got=c.bad, tensor=_t(c.exp_all), dev_in=_t(c.host_in), rerun=lambda: _t(c.exp_all),
                   third=lambda: _t(c.exp_all), oracle_again=lambda: c.exp.copy())
    assert "download-transient" in v
    assert "second download of the same tensor: DIFFERENT (4 words; now equal to the expectation)" in text
    assert "first copy back to the host was wrong" in text


def test_checker_changed_its_answer(soak):
    """the device was right all along: got == the oracle's SECOND answer"""
    c = _Case()
    first = c.exp.copy()
    first[0] ^= np.uint64(2)
    lines = []
    v = soak.diagnose("ntt", c.k, c.exp_all, first, c.sl, _t(c.exp_all), lambda: _t(c.exp_all), {7}, inputs=[("ins[0]", _t(c.host_in), c.host_in)],
                      third=lambda: _t(c.exp_all), oracle_again=lambda: c.exp.copy(), out=lines.append)
    assert "oracle-transient" in v and "third-opinion-with-fast-path" in v
    assert any("now equal to the device" in s for s in lines) and any("CHECKER changed its answer" in s for s in lines)
