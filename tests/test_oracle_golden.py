"""Pins the CPU oracle (oracle/gpqhe_oracle.c) to the reference: every value in
tests/golden/survey_8c.json was produced by the compiled reference during the
survey; tests/polymul.gp:17-18 pins p_0, p_1 of logn=7."""
import numpy as np
import pytest

from oracle.oracle import fnv, lib


@pytest.mark.parametrize("logn", ["7", "12", "15", "16", "17"])
def test_prime_chain_and_constants(golden, oracle_ctx, logn):
    rec = golden["prime_chain"][logn]
    c = oracle_ctx(int(logn), rec["count"])
    assert [str(p) for p in c.p[: len(rec["first"])]] == rec["first"]
    x = 0
    for p in c.p:
        x ^= p
    assert str(x) == rec["xor_all"]
    if not 10 <= int(logn) <= 15:  # inside 10..15 logqub comes from the security table (src/precomp.c:338-340)
        assert lib().orc_dimub(int(logn), rec["logq"]) == rec["count"]
    k = golden["p0_constants"][logn]
    z, zi = c.zetas(0), c.zetas(0, inverse=True)
    assert str(c.const("pinv_mont", 0)) == k["pinv_mont"]
    assert str(c.const("pinv_barr", 0)) == k["pinv_barr"]
    assert str(c.const("ninv", 0)) == k["ninv"]
    assert str(z[c.n // 2]) == k["zetas_n_2"]
    assert str(z[1]) == k["zetas_1"]
    assert str(zi[1]) == k["zetas_inv_1"]


def test_polymul_gp_primes(oracle_ctx):
    # hard-coded in the reference's tests/polymul.gp:17-18
    c = oracle_ctx(7, 5)
    assert c.p[0] == 576460752303434497 and c.p[1] == 576460752303436801


@pytest.mark.parametrize("logn", ["7", "12", "15", "16", "17"])
def test_single_limb_ntt_kat(golden, oracle_ctx, logn):
    kat = golden["ntt_kat_seed1_limb0"][logn]
    c = oracle_ctx(int(logn), golden["prime_chain"][logn]["count"])
    a = c.gen(1, 1)
    assert fnv(a) == kat["input"]
    out = c.ntt(a, 0)
    assert fnv(out) == kat["ntt"]
    assert [str(v) for v in out[:3]] == kat["out012"]
    assert np.array_equal(c.invntt(out, 0), a)


@pytest.mark.parametrize("logn", ["7", "12", "15"])
def test_he_mul_core_kat(golden, oracle_ctx, logn):
    kat = golden["he_mul_core_kat"][logn]
    seeds = golden["he_mul_core_kat"]["_seeds"]
    dA, dB = kat["dA"], kat["dB"]
    c = oracle_ctx(int(logn), golden["prime_chain"][logn]["count"])
    ins = [c.gen(seeds[k], dA) for k in ("a0", "a1", "b0", "b1")]
    assert [fnv(x) for x in ins] == kat["inputs"]
    d0, d1, d2 = c.he_mul_tensor(*ins, dA)
    assert (fnv(d0), fnv(d1), fnv(d2)) == (kat["d0"], kat["d1"], kat["d2"])
    c0, c1 = c.keyswitch(c.gen(seeds["d2"], dB), c.gen(seeds["evk0"], dB), c.gen(seeds["evk1"], dB), dB)
    assert (fnv(c0), fnv(c1)) == (kat["c0"], kat["c1"])


def test_reduce_edge_cases(oracle_ctx):
    """montgomery_reduce / barrett_reduce (src/reduce.c) against Python integers."""
    import ctypes as C
    L = lib()
    c = oracle_ctx(7, 5)
    rng = np.random.default_rng(7)
    for p in c.p[:2]:
        pinv_m, pinv_b = L.orc_montgomery_inv(p), L.orc_barrett_inv(p)
        assert (p * pinv_m) % (1 << 64) == 1
        assert pinv_b == (1 << 120) // p
        Rinv = pow(1 << 64, -1, p)
        vals = [0, 1, p - 1, p - 2] + [int(v) % p for v in rng.integers(0, 2**63, 32, dtype=np.uint64)]
        for a in vals:
            for b in (0, 1, p - 1, vals[-1]):
                prod = a * b
                # by-value u128 is not expressible in ctypes: go through the slab ops instead
                ra = c.rns_mul(np.full(c.n, a, np.uint64), np.full(c.n, b, np.uint64), c.p.index(p))
                assert int(ra[0]) == prod % p
                rs = c.rns_add(np.full(c.n, a, np.uint64), np.full(c.n, b, np.uint64), c.p.index(p))
                assert int(rs[0]) == (a + b) % p
        assert Rinv * (1 << 64) % p == 1


def test_golden_fixture_is_a_faithful_transcription_of_the_survey(golden):
    """tests/golden/survey_8c.json carries the values SURVEY.md section 8c recorded from the compiled reference
    (the reference cannot be rebuilt here: no <gcrypt.h>).  The fixture names, per entry, the SURVEY.md line its values
    were taken from (`_survey_rows`); every number and digest must stand ON that line -- location, not mere occurrence --
    so the pin is the survey's capture of the reference's output and not something re-derived from our own code."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = open(os.path.join(root, "SURVEY.md")).read().split("\n")
    rows = golden["_survey_rows"]
    wrong, checked = [], 0

    def leaves(v):
        if isinstance(v, dict):
            for k, x in v.items():
                if not k.startswith("_"):
                    yield from leaves(x)
        elif isinstance(v, list):
            for x in v:
                yield from leaves(x)
        elif isinstance(v, str):
            yield v

    for group in ("prime_chain", "p0_constants", "ntt_kat_seed1_limb0", "he_mul_core_kat"):
        for key, rec in golden[group].items():
            if key.startswith("_"):
                continue
            line = lines[rows["%s/%s" % (group, key)] - 1]
            assert line.lstrip().startswith("| %s " % key) or line.lstrip().startswith("| **%s" % key), line[:40]   # the row of that logn
            for v in leaves(rec):
                checked += 1
                if v not in line:
                    wrong.append((group, key, v))
    for i, rec in enumerate(golden["phat_invmp_logn7"]):
        if i:
            for v in rec:
                checked += 1
                if v not in lines[rows["phat_invmp_logn7/%d" % i] - 1]:
                    wrong.append(("phat_invmp_logn7", i, v))
    assert not wrong, wrong
    assert checked >= 125
    assert "section 8c" in golden["_survey_rows"]["_what"] and lines[rows["prime_chain/7"] - 3].startswith("| logn | p_0")
