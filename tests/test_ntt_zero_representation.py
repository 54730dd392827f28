"""CPU tier: the oracle reproduces the reference's representation of zero in `ntt` output (src/ntt.c:45-48), and the
rule the HIP path relies on -- every residue is canonical except that a residue 0 may be stored as p -- holds."""
import numpy as np
import pytest

from tests.zero_cases import limb_cases


def test_logn1_sum_leg_equal_to_p_is_stored_as_p(oracle_ctx):
    o = oracle_ctx(1, 2)
    for d in range(2):
        p = o.p[d]
        R = pow(2, 64, p)
        w = int(o.zetas(d)[1]) * pow(R, -1, p) % p          # standard-form twiddle of the only stage
        out = o.ntt(np.array([p - w, 1], dtype=np.uint64), d)
        # a[0] + t = (p - w) + 1*w = p  ->  src/ntt.c:47 keeps p;  a[0] - t = p - 2w (mod p), canonical
        assert int(out[0]) == p and int(out[1]) == (p - 2 * w) % p


@pytest.mark.parametrize("logn", [1, 3, 7, 12, 13])
def test_reference_words_are_canonical_except_zero_as_p(oracle_ctx, logn):
    o = oracle_ctx(logn, 2)
    rng = np.random.default_rng(logn)
    seen_p = 0
    for d in range(2):
        p = o.p[d]
        for name, a in limb_cases(o, d, rng):
            out = o.ntt(a, d)
            assert int(out.max()) <= p, name
            res = out % np.uint64(p)
            # the residues are the transform's: inverse of the canonicalised output gives the input back
            assert np.array_equal(o.invntt(res, d), a), name
            seen_p += int((out == np.uint64(p)).sum())
            if name.startswith("all-zero"):
                assert not out.any()
            if name == "zero at sum-leg position 0" and o.n >= 2:
                assert int(out[0]) == p, "a sum leg x + t == p with t != 0 is stored as p (src/ntt.c:47)"
            if name.startswith("zero at difference-leg") :
                assert int(out[1]) == 0, "x - t with x == t is 0 (src/ntt.c:46)"
    assert seen_p > 0
