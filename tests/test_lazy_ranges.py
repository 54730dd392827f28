"""Integer model of the split-twiddle multiply and of the wide-split forward stages (gpqhe_amd/csrc/modarith.hpp):
every intermediate fits the register it lives in and the lazy ranges close, for the largest c each class admits.
CPU only; the kernels themselves are checked bit for bit on the GPU (tests/test_ntt_gpu.py)."""
import random

import pytest

M64 = (1 << 64) - 1
SPLIT_CMAX = 306000000   # GPQ_SPLIT_CMAX
WIDE_CMAX = 134217000    # GPQ_WIDE_CMAX


def mulmod_split(a, wx, wy, c):
    """mulmod_split(): returns T' with a*w == T' + (c+1) (mod p); asserts what the device code relies on."""
    al, ah = a & 0x7FFFFFFF, a >> 31
    assert ah < (1 << 32)
    t0 = al * (wx & 0xFFFFFFFF) + ah * (wy & 0xFFFFFFFF)
    assert t0 <= M64
    t1 = al * (wx >> 32) + (t0 >> 32) + ah * (wy >> 32)
    assert t1 <= M64
    th = t1 >> 27
    assert th < (1 << 32)
    ntl = ((~t0) & 0xFFFFFFFF) | (((~t1) & 0x7FFFFFF) << 32)
    r = c * th + ntl
    assert r <= M64
    return r


def pairs(p, w):
    return p - w, p - ((w << 31) % p)


@pytest.mark.parametrize("c", [4849665, 113508353, 178956970, 218103809, SPLIT_CMAX - 1])
def test_split_multiply_below_3p_for_operands_below_6p(c):
    """mulmod_split for the split class (round 5: forward data < 6p, inverse data < 3p, one conditional subtraction of 3p per butterfly):
    a multiplicand below 6p gives th <= 3.5 * 2^30 and a product term below 3p for every c the class admits (c < 2^30 / 3.5)."""
    p = (1 << 59) + c
    rnd = random.Random(c)
    for a in [0, 1, 6 * p - 1, 6 * p - 2, 4 * p - 1, (1 << 61) - 1, 1 << 61, (1 << 61) + (1 << 60), 3 * (1 << 60) + (1 << 31) - 1] + [rnd.randrange(6 * p) for _ in range(3000)]:
        for w in (1, p - 1, rnd.randrange(1, p)):
            t = mulmod_split(a, *pairs(p, w), c) + c + 1
            assert t < 3 * p and t % p == a * w % p
    for wx in (p - 1, 1):
        for wy in (p - 1, 1):
            assert mulmod_split(6 * p - 1, wx, wy, c) + c + 1 < 3 * p


@pytest.mark.parametrize("c", [4849665, 178956970, 218103809, SPLIT_CMAX - 1])
def test_split_stage_ranges_close(c):
    """ct_bfly / gs_bfly_split / gs_last of the split class as modarith.hpp writes them: forward x, y < 6p stay < 6p, inverse x, y < 3p stay
    < 3p, the last inverse stage (n^-1 folded in) canonicalises products below 3p with two conditional subtractions."""
    p = (1 << 59) + c
    c1 = c + 1
    kx0, kx1x, kyx = c1, (c1 - 3 * p) & M64, 3 * p - 2 * c1
    rnd = random.Random(c + 7)
    edge = [0, 1, p - 1, p, 3 * p - 1, 3 * p, 6 * p - 1, 6 * p - 2, (1 << 61) - 1, 1 << 61, 3 * (1 << 60)]
    for _ in range(4000):
        w = rnd.choice([1, 2, p - 1, rnd.randrange(1, p)])
        wx, wy = pairs(p, w)
        x = rnd.choice(edge + [rnd.randrange(6 * p)]) % (6 * p)
        y = rnd.choice(edge + [rnd.randrange(6 * p)]) % (6 * p)
        T = mulmod_split(y, wx, wy, c)
        assert (T + c1) % p == y * w % p and T + c1 < 3 * p
        xs = (x + (kx1x if x >= 3 * p else kx0)) & M64
        x2, y2 = (xs + T) & M64, (xs + kyx - T) & M64
        assert x2 % p == (x + y * w) % p and y2 % p == (x - y * w) % p
        assert x2 < 6 * p and y2 < 6 * p
        # inverse
        x, y = x % (3 * p), y % (3 * p)
        v, d = x + y, x + 3 * p - y
        assert 0 < d < 6 * p
        x2 = v - 3 * p if v >= 3 * p else v
        y2 = mulmod_split(d, wx, wy, c) + c1
        assert x2 < 3 * p and x2 % p == (x + y) % p and y2 < 3 * p and y2 % p == (x - y) * w % p
        for val, arg in ((x + y, x + y), (x - y, d)):
            t = mulmod_split(arg, wx, wy, c) + c1
            assert t < 3 * p
            t = t - 2 * p if t >= 2 * p else t
            t = t - p if t >= p else t
            assert t == val * w % p


def test_product_operands_of_the_split_class_fit_the_general_multiply():
    """mid kernels, split limbs: left operand < 2p, right operand < 6p (forward data as it comes): the 7-mad multiply's th fits 32 bits and
    the product leaves below 4p for the largest c of the class."""
    c = SPLIT_CMAX - 1
    p = (1 << 59) + c
    a, w = 2 * p - 1, 6 * p - 1
    a0, a1, w0, w1 = a & 0xFFFFFFFF, a >> 32, w & 0xFFFFFFFF, w >> 32
    assert a0 * w1 + ((a0 * w0) >> 32) + a1 * w0 <= M64
    x = a * w
    xh = x >> 59
    assert xh <= M64
    t = c * xh
    assert (t >> 59) < (1 << 32)
    tprime = (x & ((1 << 59) - 1)) + c * (t >> 59) + ((1 << 59) - 1 - (t & ((1 << 59) - 1)))
    assert tprime + c + 1 < 4 * p and (tprime + c + 1) % p == x % p


def _general_multiply(a, w, c):
    """mulmod_raw_t / mulmod_lazy (the 7-mad form) with every register bound asserted; returns the lazily reduced product"""
    p = (1 << 59) + c
    a0, a1, w0, w1 = a & 0xFFFFFFFF, a >> 32, w & 0xFFFFFFFF, w >> 32
    mid = a0 * w1 + ((a0 * w0) >> 32)
    assert mid <= M64
    mid += a1 * w0
    assert mid <= M64                                              # the middle column of the 64 x 64 product
    hi = a1 * w1 + (mid >> 32)
    assert hi <= M64
    x = a * w
    xh = x >> 59
    assert xh <= M64
    t = c * xh
    th = t >> 59
    assert th < (1 << 32)                                          # the second fold is ONE 32 x 32 multiply
    r = (x & ((1 << 59) - 1)) + c * th + ((1 << 59) - 1 - (t & ((1 << 59) - 1))) + c + 1
    assert r <= M64 and r % p == x % p
    return r


def test_products_of_the_split_class_over_random_operands():
    """the widened class leaves 2.5 % between 12 c^2 and 2^60; one edge pair guarded it.  Random left < 2p, right < 6p for the
    largest c (and the real chain's), plus the corners: th fits 32 bits, the product leaves below 4p, d1 = sum of two below 8p."""
    rng = random.Random(2859)
    for c in (SPLIT_CMAX - 1, 218103809, 4849665):
        p = (1 << 59) + c
        corners = [(2 * p - 1, 6 * p - 1), (2 * p - 1, 0), (0, 6 * p - 1), (p, 3 * p), (2 * p - 1, 4 * p), (1, 6 * p - 1)]
        worst = 0
        for a, w in corners + [(rng.randrange(2 * p), rng.randrange(6 * p)) for _ in range(20000)]:
            r = _general_multiply(a, w, c)
            assert 0 < r < 4 * p
            worst = max(worst, r)
        assert 2 * worst < 8 * p                                   # d1 = c0 c1' + c1 c0' enters inv_from8
    # the compile-time guards beside GPQ_SPLIT_CMAX (modarith.hpp) say the same in closed form
    cmax = SPLIT_CMAX
    assert 7 * (1 << 29) * cmax < 1 << 60 and 12 * cmax < 1 << 32 and 12 * cmax * cmax < 1 << 60


@pytest.mark.parametrize("c", [4849665, 113508353, WIDE_CMAX - 1])
def test_wide_split_stage_ranges_close(c):
    """ct_bfly_wide: stage A (no subtraction) takes x, y < 6p to x' < 8p, y' <= 8p - c - 2; stage B (subtract 4p) takes
    x, y < 8p back below 6p; every multiplicand that can occur keeps th in 32 bits and the product term below 2p."""
    p = (1 << 59) + c
    c1 = c + 1
    kx0, kx1, kys = c1, (c1 - 4 * p) & M64, 2 * p - 2 * c1
    rnd = random.Random(c)
    edge = [0, 1, (1 << 62) - 1, 1 << 62, (1 << 62) + (1 << 31) - 2, 8 * p - c - 2, 6 * p - 1, 4 * p, 4 * p - 1]
    for wx in (p - 1, 1, p // 2):
        for wy in (p - 1, 1, p // 3):
            for a in edge:
                assert mulmod_split(a, wx, wy, c) + c1 < 2 * p
    for _ in range(4000):
        w = rnd.choice([1, 2, p - 1, rnd.randrange(1, p)])
        wx, wy = pairs(p, w)
        for kind, lim in (("A", 6 * p), ("B", 8 * p)):
            x = rnd.choice(edge + [lim - 1, rnd.randrange(lim)]) % lim
            y = min(rnd.choice(edge + [lim - 1, rnd.randrange(lim)]), lim - 1 if kind == "A" else 8 * p - c - 2)
            T = mulmod_split(y, wx, wy, c)
            assert (T + c1) % p == y * w % p and T + c1 < 2 * p
            xs = (x + (kx1 if kind == "B" and x >= 4 * p else kx0)) & M64
            x2, y2 = (xs + T) & M64, (xs + kys - T) & M64
            assert x2 % p == (x + y * w) % p and y2 % p == (x - y * w) % p
            if kind == "A":
                assert x2 < 8 * p and y2 <= 8 * p - c - 2
            else:
                assert x2 < 6 * p and y2 < 6 * p


def test_product_operands_of_the_wide_class_fit_the_general_multiply():
    """tensor_mid: left operand < 4p, right operand < 6p for wide limbs: the 7-mad multiply's columns and folds stay in range."""
    c = WIDE_CMAX - 1
    p = (1 << 59) + c
    a, w = 4 * p - 1, 6 * p - 1
    a0, a1, w0, w1 = a & 0xFFFFFFFF, a >> 32, w & 0xFFFFFFFF, w >> 32
    mid = a0 * w1 + ((a0 * w0) >> 32) + a1 * w0
    assert mid <= M64
    x = a * w
    xh = x >> 59
    assert xh <= M64
    t = c * xh
    assert (t >> 59) < (1 << 32)
    tprime = (x & ((1 << 59) - 1)) + c * (t >> 59) + ((1 << 59) - 1 - (t & ((1 << 59) - 1)))
    assert tprime + c + 1 < 4 * p and (tprime + c + 1) % p == x % p


@pytest.mark.parametrize("c", [4849665, 113508353, WIDE_CMAX - 1])
def test_wide_gs_stage_ranges_close(c):
    """gs_bfly_wide: inverse data of the wide class lives in [0, 4p).  v = x + y < 8p; d = x + 4p - y is a legal multiplicand
    (8p - 1 <= 2^62 + 2^31 - 2); the product leg comes out below 2p, the sum leg below 4p after ONE conditional subtraction of
    4p -- and needs none when both inputs are product legs of the stage before (< 2p each): the case the kernels skip statically.
    gs_last takes the same ranges to canonical outputs."""
    p = (1 << 59) + c
    c1 = c + 1
    assert 8 * p - 1 <= (1 << 62) + (1 << 31) - 2
    rnd = random.Random(c + 1)
    edge = [0, 1, p - 1, p, 2 * p - 1, 2 * p, 4 * p - 1, 4 * p - 2, (1 << 61) - 1, 1 << 61]
    for _ in range(4000):
        w = rnd.choice([1, 2, p - 1, rnd.randrange(1, p)])
        wx, wy = pairs(p, w)
        for both_product_legs in (False, True):
            lim = 2 * p if both_product_legs else 4 * p
            x = rnd.choice(edge + [rnd.randrange(lim)]) % lim
            y = rnd.choice(edge + [rnd.randrange(lim)]) % lim
            v, d = x + y, x + 4 * p - y
            assert 0 < d < 8 * p and d <= M64
            x2 = v if both_product_legs else (v - 4 * p if v >= 4 * p else v)
            y2 = mulmod_split(d, wx, wy, c) + c1
            assert x2 < 4 * p and x2 % p == (x + y) % p
            assert y2 < 2 * p and y2 % p == (x - y) * w % p
        # last stage with the n^-1 scaling folded in: both legs are multiplied, then one subtraction of p
        x, y = rnd.randrange(4 * p), rnd.randrange(4 * p)
        for val, arg in ((x + y, (x + y)), (x - y, x + 4 * p - y)):
            t = mulmod_split(arg, wx, wy, c) + c1
            assert t < 2 * p
            r = t - p if t >= p else t
            assert r == val * w % p
