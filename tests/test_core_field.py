"""CPU tests of the single-source field arithmetic (vgen_amd/csrc/core/fe.h, 9 limbs x 29 bits).

The header is compiled for the host by g++ (tests/native) — the same source hipcc compiles into the
kernels — and checked against Python integers, including adversarial limb patterns at the largest
magnitudes each function documents (limb overflow is the classic failure of reduced-radix code and
random canonical inputs do not exercise it).
"""
import ctypes
import os
import random
import subprocess

import pytest

from conftest import locked_make

HERE = os.path.dirname(os.path.abspath(__file__))
P = 2**256 - 2**32 - 977
M29 = (1 << 29) - 1


@pytest.fixture(scope="module")
def core():
    locked_make("-s", "-C", os.path.join(HERE, "native"))
    return ctypes.CDLL(os.path.join(HERE, "native", "libcoretest.so"))


A9 = ctypes.c_uint32 * 9
A8 = ctypes.c_uint32 * 8


def val(limbs):
    return sum(int(x) << (29 * i) for i, x in enumerate(limbs))


def limbs_of(v):
    out = [(v >> (29 * i)) & M29 for i in range(8)]
    out.append(v >> 232)
    return out


# A product leaves its top fold un-rippled in limbs 0 and 1 (fe.h, "weakly normalised"):
W0, W1 = M29 + (1 << 23), M29 + (1 << 16)


def rand_limbs(rng, mag, style):
    """limbs of magnitude `mag` = a sum of `mag` weakly normalised values: n[0] <= mag*(2^29 + 2^23),
    n[1] <= mag*(2^29 + 2^16), n[2..7] <= mag*(2^29-1), n[8] <= mag*2^24"""
    his, top = [mag * W0, mag * W1] + [mag * M29] * 6, mag * (1 << 24)
    if style == "max":
        return his + [top]
    if style == "min":
        return [0] * 9
    if style == "mixed":
        return [rng.choice([0, 1, hi, hi - 1, rng.randrange(hi + 1)]) for hi in his] + \
               [rng.choice([0, top, top - 1, rng.randrange(top + 1)])]
    return [rng.randrange(hi + 1) for hi in his] + [rng.randrange(top + 1)]


def check_mag1(limbs):
    """strictly normalised limbs (fe_normalize / fe_normalize_weak / fe_canonicalize)"""
    assert all(0 <= x <= M29 for x in limbs[:8]) and 0 <= limbs[8] <= (1 << 24), limbs


def check_weak(limbs):
    """the form a product comes out in"""
    assert limbs[0] <= W0 and limbs[1] <= W1 and all(x <= M29 for x in limbs[2:8]) and limbs[8] < (1 << 24), limbs


@pytest.mark.parametrize("ma,mb", [(1, 1), (2, 1), (1, 2), (2, 3), (3, 2), (6, 1), (1, 6), (2, 2)])
def test_mul_all_magnitudes(core, ma, mb):
    rng = random.Random(ma * 10 + mb)
    styles = ["max", "min", "mixed", "rand"]
    for sa in styles:
        for sb in styles:
            for _ in range(40 if "rand" in (sa, sb) or "mixed" in (sa, sb) else 1):
                a, b = rand_limbs(rng, ma, sa), rand_limbs(rng, mb, sb)
                r = A9()
                core.core_fe_mul(A9(*a), A9(*b), r)
                check_weak(list(r))
                assert val(r) % P == (val(a) * val(b)) % P, (a, b)


def test_sqr(core):
    rng = random.Random(5)
    for style in ["max", "min"] + ["mixed"] * 200 + ["rand"] * 200:
        a = rand_limbs(rng, 1, style)
        r = A9()
        core.core_fe_sqr(A9(*a), r)
        check_weak(list(r))
        assert val(r) % P == (val(a) ** 2) % P


@pytest.mark.parametrize("mag", [1, 2, 3, 5, 7])
def test_normalize(core, mag):
    rng = random.Random(mag)
    specials = [limbs_of(P), limbs_of(P - 1), limbs_of(P + 1), limbs_of(2**256 - 1), limbs_of(0),
                limbs_of(2**256 - 2**32 - 978), [M29] * 8 + [1 << 24], [0] * 8 + [1 << 24],
                limbs_of(P)[:8] + [limbs_of(P)[8] + (1 << 24)]]
    cases = specials + [rand_limbs(rng, mag, s) for s in ["max", "min"] + ["mixed"] * 300 + ["rand"] * 300]
    for a in cases:
        r = A9()
        core.core_fe_normalize(A9(*a), r, 0)
        assert val(r) == val(a) % P and val(r) < P, a
        check_mag1(list(r))
        assert list(r) == limbs_of(val(a) % P)
        w = A9()
        core.core_fe_normalize(A9(*a), w, 1)
        check_mag1(list(w))
        assert val(w) % P == val(a) % P


@pytest.mark.parametrize("mag", [1, 2, 3, 4, 5, 6])
def test_neg(core, mag):
    rng = random.Random(mag + 100)
    for style in ["max", "min"] + ["mixed"] * 100 + ["rand"] * 100:
        a = rand_limbs(rng, mag, style)
        r = A9()
        core.core_fe_neg(A9(*a), mag, r)
        assert (val(r) + val(a)) % P == 0
        assert all(0 <= x <= (mag + 1) * W0 for x in list(r)[:8]) and r[8] <= (mag + 1) * (1 << 24)
        assert all(x < 2**32 for x in r)


def inverse_vectors(rng, n_random):
    """Inputs for the inversion tests: edge values, powers of two and their neighbours / negatives (long runs of even
    or odd divsteps), values whose gcd walk is slow, random ones."""
    v = [0, 1, 2, 3, P - 1, P - 2, P - 3, (P + 1) // 2, (P - 1) // 2, 2**255, 2**256 % P, 977, 2**32 + 977, 2**32 + 976]
    v += [(1 << k) % P for k in range(1, 256)] + [P - (1 << k) for k in range(256)] + [(1 << k) - 1 for k in range(2, 257, 7)]
    v += [pow(3, k, P) for k in range(1, 40)] + [P // k for k in range(2, 40)]
    v += [rng.randrange(1, P) for _ in range(n_random)]
    return [x % P for x in v]


def test_inv(core):
    """fe_inv = divsteps (safegcd) inversion: canonical a^-1 against Python, for canonical and for weakly normalised /
    higher-magnitude inputs; the Fermat ladder (fe_inv_fermat) agrees."""
    rng = random.Random(9)
    for v in inverse_vectors(rng, 4000):
        r = A9()
        core.core_fe_inv(A9(*limbs_of(v)), r)
        check_mag1(list(r))
        assert val(r) == (pow(v, -1, P) if v else 0), hex(v)
    for mag in (1, 2, 3, 7):
        for style in ["max", "mixed", "rand", "rand", "mixed"]:
            for _ in range(40):
                a = rand_limbs(rng, mag, style)
                r = A9()
                core.core_fe_inv(A9(*a), r)
                assert val(r) == (pow(val(a), -1, P) if val(a) % P else 0), a
    for v in [1, 2, P - 1, P - 2, 2**255, 977] + [rng.randrange(1, P) for _ in range(60)]:
        r = A9()
        core.core_fe_inv_fermat(A9(*limbs_of(v)), r)
        assert (val(r) * v) % P == 1
    r = A9()
    core.core_fe_inv_fermat(A9(*limbs_of(0)), r)
    assert val(r) % P == 0


def test_inversion_final_lift_over_the_whole_range_of_d(core):
    """The last step of fe_inv turns d, anywhere in (-2p, p) as signed 9 x 29 limbs, into the canonical sign(f) * d.  The
    inversion itself practically never produces a d near -2p, so the lift is driven directly: both ends of the interval and
    their neighbourhoods (round 3's lift by 2p wrapped limb 8 for f > 0 and d within ~2^233 of -2p), zero, +-1, +-p and
    random values, for both signs of f."""
    import ctypes
    I9 = ctypes.c_int * 9

    def signed_limbs(d):
        # limbs 0..7 in [0, 2^29), limb 8 carries the sign: d = sum n_i 2^(29 i)
        n = [(d >> (29 * i)) & 0x1FFFFFFF for i in range(8)]
        n.append(d >> 232)          # arithmetic shift: floor division, negative for negative d
        assert sum(x << (29 * i) for i, x in enumerate(n)) == d and -2**26 < n[8] < 2**26
        return n
    rng = random.Random(77)
    ds = [0, 1, -1, P - 1, P - 2, -P, -P + 1, -P - 1, -2 * P + 1, -2 * P + 2, -2 * P + 2**40, -2 * P + 2**232, -2 * P + 2**233 + 12345,
          -2 * P + 2**240, -(1 << 256), -(1 << 256) - 977, -(1 << 256) + 1, 2**255, -2**255]
    ds += [rng.randrange(-2 * P + 1, P) for _ in range(3000)] + [-2 * P + 1 + rng.randrange(2**234) for _ in range(500)]
    ds += [P - 1 - rng.randrange(2**200) for _ in range(200)]
    for d in ds:
        assert -2 * P < d < P
        for f_top in (0x00FFFFFF, 1, 0, -1, -0x01000000):      # sign word of f: >= 0 or < 0
            r = A9()
            core.core_fe_divsteps_lift(I9(*signed_limbs(d)), f_top, r)
            check_mag1(list(r))
            assert val(r) == (d if f_top >= 0 else -d) % P, (hex(d), f_top)


def test_divsteps_batch_is_the_textbook_recurrence(core):
    """One batch of 29 divsteps (masks, low 32 bits only) against the definition on Python integers: the matrix t must
    satisfy t [f, g] = 2^29 [f', g'] for the (f', g') the recurrence reaches, and delta must agree."""
    rng = random.Random(29)
    T4 = ctypes.c_int * 4
    for _ in range(3000):
        f = rng.getrandbits(rng.choice([8, 31, 64, 256])) | 1
        g = rng.getrandbits(rng.choice([1, 8, 31, 64, 256])) * rng.choice([1, 1, 2, 16, 2**20])
        if rng.random() < 0.3:
            f, g = -f, -g if rng.random() < 0.5 else g
        twice_delta = rng.choice([1, 1, 3, -1, -5, 7, 29, -41])          # 2 delta, odd
        zeta = -(twice_delta + 1) // 2                                    # zeta = -(delta + 1/2)
        t = T4()
        z2 = core.core_fe_divsteps29(zeta, f & 0xFFFFFFFF, g & 0xFFFFFFFF, t)
        ff, gg, td = f, g, twice_delta
        for _i in range(29):
            if td > 0 and gg & 1:
                td, ff, gg = 2 - td, gg, (gg - ff) // 2
            elif gg & 1:
                td, gg = td + 2, (gg + ff) // 2
            else:
                td, gg = td + 2, gg // 2
        u, v, q, r = list(t)
        assert u * f + v * g == ff << 29 and q * f + r * g == gg << 29
        assert z2 == -(td + 1) // 2
        assert abs(u) + abs(v) <= 1 << 29 and abs(q) + abs(r) <= 1 << 29


def test_word_conversion_roundtrip(core):
    rng = random.Random(11)
    for v in [0, 1, P - 1, 2**256 - 1, 2**255] + [rng.randrange(2**256) for _ in range(200)]:
        w, back = A8(), A9()
        core.core_fe_words(A9(*limbs_of(v)), w, back)
        assert sum(int(x) << (32 * i) for i, x in enumerate(w)) == v
        assert list(back) == limbs_of(v)


@pytest.mark.parametrize("ma,mb,mc", [(1, 1, 3), (1, 3, 2), (3, 1, 3), (2, 3, 3), (1, 1, 0)])
def test_mul_add_and_sqr_add(core, ma, mb, mc):
    rng = random.Random(ma * 100 + mb * 10 + mc)
    styles = ["max", "min", "mixed", "rand"]
    for sa in styles:
        for sb in styles:
            for sc in styles:
                for _ in range(10 if "rand" in (sa, sb, sc) or "mixed" in (sa, sb, sc) else 1):
                    a, b = rand_limbs(rng, ma, sa), rand_limbs(rng, mb, sb)
                    c = rand_limbs(rng, mc, sc) if mc else [0] * 9
                    r = A9()
                    core.core_fe_mul_add(A9(*a), A9(*b), A9(*c), r, 0)
                    check_weak(list(r))
                    assert val(r) % P == (val(a) * val(b) + val(c)) % P
                    if ma == 1:
                        core.core_fe_mul_add(A9(*a), A9(*a), A9(*c), r, 1)
                        check_weak(list(r))
                        assert val(r) % P == (val(a) ** 2 + val(c)) % P


def test_canonicalize_weakly_normalised_inputs(core):
    rng = random.Random(321)
    specials = [limbs_of(P), limbs_of(P - 1), limbs_of(P + 1), limbs_of(2**256 - 1), limbs_of(0), limbs_of(1),
                limbs_of(2**256 - 2**32 - 978), [M29] * 8 + [1 << 24], [0] * 8 + [1 << 24], [M29] * 8 + [(1 << 24) - 1],
                limbs_of(P)[:8] + [1 << 24], [5] + [0] * 7 + [(1 << 24) + 1], limbs_of(P + 2**32 + 976),
                limbs_of(2**256 - 1)[:8] + [(1 << 24) + 1]]
    cases = specials + [[rng.choice([0, 1, M29, M29 - 1, rng.randrange(M29 + 1)]) for _ in range(8)] +
                        [rng.choice([0, 1 << 24, (1 << 24) - 1, (1 << 24) + 1, rng.randrange((1 << 24) + 2)])]
                        for _ in range(3000)]
    # the form products come out in: limbs 0 and 1 carry an un-rippled top fold, n[8] < 2^24 — incl. values that
    # reach p or 2^256 only once the carries are propagated
    top = [M29] * 6 + [(1 << 24) - 1]
    cases += [[W0, W1] + top, [W0, 0] + top, [0, W1] + top, [M29 + 1, M29] + top, [M29 - 976, M29 - 8] + top,
              [M29 - 975, M29 - 8] + top, [M29 + 1 - 977, M29 - 8] + top]
    cases += [[rng.choice([0, M29, W0, W0 - 1, rng.randrange(W0 + 1)]), rng.choice([0, M29, W1, rng.randrange(W1 + 1)])] +
              [rng.choice([0, M29, M29 - 1, rng.randrange(M29 + 1)]) for _ in range(6)] +
              [rng.choice([0, (1 << 24) - 1, rng.randrange(1 << 24)])] for _ in range(3000)]
    for a in cases:
        r = A9()
        core.core_fe_canonicalize(A9(*a), r)
        assert list(r) == limbs_of(val(a) % P), a
        assert core.core_fe_parity_weak(A9(*a)) == (val(a) % P) & 1, a   # parity without the representative
        core.core_fe_canonicalize_product(A9(*a), r)                      # one carry pass + rare slow path
        assert list(r) == limbs_of(val(a) % P), a
