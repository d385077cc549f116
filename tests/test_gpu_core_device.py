"""GPU unit tests of the single-source core AS HIPCC COMPILES IT (vgen_amd/csrc/core/fe.h, hash.h).

tests/test_core_field.py / test_core_hash.py run the same headers built by g++.  Some code exists only in the
device build: the wave-uniform slow path of fe_canonicalize_product behind VG_ANY_LANE (a ballot: the full
fe_canonicalize runs for the whole wave when ANY lane needs it), the s_mov "opaque" multipliers of the
column-form multiplication, the v_bitop3 / v_alignbit instruction selection of the hash rounds.  The product
kernels reach the slow path only by chance (~2^-23 per product), so here a test-only kernel
(tests/native/core_dev.hip, one input per lane) runs the adversarial vectors of the CPU tests with the lanes
arranged so that the slow path fires in some waves and not in others, and the results are compared with Python
integers / the oracle.  Counterpart of the reference's WGSL field code, src/shaders/field.wgsl:18-210.
"""
import ctypes
import os
import random
import subprocess

import pytest

from conftest import locked_make

from test_core_field import M29, P, W0, W1, check_mag1, check_weak, limbs_of, rand_limbs, val

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
(OP_MUL, OP_SQR, OP_MUL_ADD, OP_SQR_ADD, OP_CANON_PRODUCT, OP_CANON, OP_PARITY_WEAK, OP_INV, OP_NORMALIZE,
 OP_NORMALIZE_WEAK, OP_NEG, OP_MUL_THEN_CANON_PRODUCT, OP_TO_WORDS, OP_INV_FERMAT) = range(14)
H_PUB33, H_PUB65, H_SCRIPT22, H_KECCAK = range(4)


@pytest.fixture(scope="module")
def dev():
    locked_make("-s", "-C", os.path.join(HERE, "native"), "libcoredev.so")
    lib = ctypes.CDLL(os.path.join(HERE, "native", "libcoredev.so"))
    assert lib.coredev_device_count() >= 1, "no HIP device: the gpu-marked tests need an MI355X"
    return lib


def run_fe(dev, op, a, b=None, c=None):
    n = len(a)
    flat = lambda rows: (ctypes.c_uint32 * (9 * n))(*[x for r in rows for x in r]) if rows is not None else None
    out = (ctypes.c_uint32 * (9 * n))()
    rc = dev.coredev_fe(op, n, flat(a), flat(b), flat(c), out)
    assert rc == 0, f"coredev_fe failed: {rc}"
    return [list(out[9 * i:9 * i + 9]) for i in range(n)]


STYLES = ["max", "min", "mixed", "rand"]


@pytest.mark.parametrize("ma,mb", [(1, 1), (2, 3), (3, 2), (6, 1), (1, 6), (2, 2)])
def test_device_mul_all_magnitudes(dev, ma, mb):
    rng = random.Random(ma * 10 + mb)
    a, b = [], []
    for sa in STYLES:
        for sb in STYLES:
            for _ in range(64):
                a.append(rand_limbs(rng, ma, sa))
                b.append(rand_limbs(rng, mb, sb))
    for x, y, r in zip(a, b, run_fe(dev, OP_MUL, a, b)):
        check_weak(r)
        assert val(r) % P == (val(x) * val(y)) % P, (x, y)


def test_device_sqr_mul_add_sqr_add(dev):
    rng = random.Random(5)
    a = [rand_limbs(rng, 1, s) for s in ["max", "min"] + ["mixed"] * 500 + ["rand"] * 500]
    for x, r in zip(a, run_fe(dev, OP_SQR, a)):
        check_weak(r)
        assert val(r) % P == (val(x) ** 2) % P
    for ma, mb, mc in [(1, 1, 3), (1, 3, 2), (3, 1, 3), (2, 3, 3)]:
        a = [rand_limbs(rng, ma, rng.choice(STYLES)) for _ in range(1024)]
        b = [rand_limbs(rng, mb, rng.choice(STYLES)) for _ in range(1024)]
        c = [rand_limbs(rng, mc, rng.choice(STYLES)) for _ in range(1024)]
        for x, y, z, r in zip(a, b, c, run_fe(dev, OP_MUL_ADD, a, b, c)):
            check_weak(r)
            assert val(r) % P == (val(x) * val(y) + val(z)) % P
        if ma == 1:
            for x, z, r in zip(a, c, run_fe(dev, OP_SQR_ADD, a, None, c)):
                check_weak(r)
                assert val(r) % P == (val(x) ** 2 + val(z)) % P


def weak_cases(rng):
    """Weakly normalised inputs (the form products come out in), ordinary and adversarial (value >= p or >= 2^256
    once the un-rippled carries of limbs 0 and 1 are propagated: the rare lanes fe_canonicalize_product must fix)."""
    top = [M29] * 6 + [(1 << 24) - 1]
    hard = [limbs_of(P), limbs_of(P + 1), limbs_of(2**256 - 1), limbs_of(P + 2**32 + 976), [W0, W1] + top, [W0, 0] + top,
            [0, W1] + top, [M29 + 1, M29] + top, [M29 - 975, M29 - 8] + top, [M29 + 1 - 977, M29 - 8] + top,
            [M29] * 8 + [(1 << 24) - 1]]
    hard += [[rng.choice([M29, W0, W0 - 1, rng.randrange(M29 - 2000, W0 + 1)]), rng.choice([M29, W1, rng.randrange(M29 - 16, W1 + 1)])] + top
             for _ in range(200)]
    easy = [limbs_of(rng.randrange(P)) for _ in range(64)] + [limbs_of(0), limbs_of(1), limbs_of(P - 1), limbs_of(2**255)]
    easy += [[rng.randrange(W0 + 1), rng.randrange(W1 + 1)] + [rng.randrange(M29 + 1) for _ in range(6)] + [rng.randrange(1 << 23)]
             for _ in range(400)]
    return easy, hard


def test_device_canonicalize_product_slow_path_fires_per_wave(dev):
    """Waves are 64 consecutive inputs.  Layout: waves of ordinary values only (ballot false: fast path), waves with
    exactly one adversarial lane at a random position (ballot true: the slow path runs on 63 lanes that did not
    need it and must be the identity there), waves of adversarial values only, and a ragged last wave."""
    rng = random.Random(321)
    easy, hard = weak_cases(rng)
    cases = []
    for w in range(24):
        wave = [rng.choice(easy) for _ in range(64)]
        if w % 3 == 1:
            wave[rng.randrange(64)] = rng.choice(hard)
        elif w % 3 == 2:
            wave = [rng.choice(hard) for _ in range(64)]
        cases += wave
    cases += [rng.choice(hard)] + [rng.choice(easy) for _ in range(9)]      # ragged tail: 10 lanes
    for x, r in zip(cases, run_fe(dev, OP_CANON_PRODUCT, cases)):
        assert r == limbs_of(val(x) % P), x
    for x, r in zip(cases, run_fe(dev, OP_CANON, cases)):
        assert r == limbs_of(val(x) % P), x
    for x, r in zip(cases, run_fe(dev, OP_PARITY_WEAK, cases)):
        assert r[0] == (val(x) % P) & 1, x


def test_device_mul_then_canonicalize_product_as_the_kernels_chain_them(dev):
    rng = random.Random(77)
    # products whose value lands just below / at / above p and 2^256: x * 1, x * x^-1 * t, ...
    targets = [P - 1, 0, 1, 2**256 - P, 2**255, P - 2**32] + [rng.randrange(P) for _ in range(250)]
    a, b = [], []
    for t in targets:
        u = rng.randrange(1, P)
        a.append(limbs_of(u))
        b.append(limbs_of(t * pow(u, -1, P) % P))       # u * b == t (mod p)
    for x, y, r in zip(a, b, run_fe(dev, OP_MUL_THEN_CANON_PRODUCT, a, b)):
        assert r == limbs_of(val(x) * val(y) % P)


@pytest.mark.parametrize("mag", [1, 3, 7])
def test_device_normalize_neg(dev, mag):
    rng = random.Random(mag)
    cases = [limbs_of(P), limbs_of(P - 1), limbs_of(P + 1), limbs_of(2**256 - 1), [M29] * 8 + [1 << 24]]
    cases += [rand_limbs(rng, mag, s) for s in ["max", "min"] + ["mixed"] * 300 + ["rand"] * 300]
    for x, r in zip(cases, run_fe(dev, OP_NORMALIZE, cases)):
        assert r == limbs_of(val(x) % P), x
    for x, r in zip(cases, run_fe(dev, OP_NORMALIZE_WEAK, cases)):
        check_mag1(r)
        assert val(r) % P == val(x) % P
    m = min(mag, 6)
    cases = [rand_limbs(rng, m, s) for s in ["max", "min"] + ["mixed"] * 100 + ["rand"] * 100]
    for x, r in zip(cases, run_fe(dev, OP_NEG, cases, [[m] + [0] * 8 for _ in cases])):
        assert (val(r) + val(x)) % P == 0 and all(v < 2**32 for v in r)


def test_device_inverse_and_word_conversion(dev):
    rng = random.Random(9)
    from test_core_field import inverse_vectors
    # the divsteps inversion as hipcc compiles it (signed 64-bit multiply-adds, the wave-uniform early exit: the lanes of
    # a wave need different numbers of batches, and vectors that finish at once sit beside ones that need them all)
    vals = inverse_vectors(rng, 1500)
    rows = run_fe(dev, OP_INV, [limbs_of(v) for v in vals])
    for v, r in zip(vals, rows):
        check_mag1(r)
        assert val(r) == (pow(v, -1, P) if v else 0), hex(v)
    # weakly normalised / higher-magnitude inputs, as the product trees hand them over
    cases = [rand_limbs(rng, m, s) for m in (1, 2, 7) for s in ["max", "mixed", "rand"] for _ in range(64)]
    for x, r in zip(cases, run_fe(dev, OP_INV, cases)):
        assert val(r) == (pow(val(x), -1, P) if val(x) % P else 0)
    # ... and the Fermat ladder kept as its cross-check
    some = vals[:256]
    for v, r in zip(some, run_fe(dev, OP_INV_FERMAT, [limbs_of(v) for v in some])):
        assert (val(r) * v) % P == (1 if v else 0)
    vals = [0, 1, P - 1, 2**256 - 1, 2**255] + [rng.randrange(2**256) for _ in range(200)]
    for v, r in zip(vals, run_fe(dev, OP_TO_WORDS, [limbs_of(v) for v in vals])):
        assert r == limbs_of(v)


def test_device_hashes_match_the_oracle(dev):
    from oracle import pyoracle as vo
    rng = random.Random(21)
    keys = [1, 2, 3, N - 1, 2**255, 0xFF] + [rng.randrange(1, N) for _ in range(250)]
    pubs = [vo.pubkey(k) for k in keys]
    n = len(pubs)
    xs, ys = b"".join(p[1:33] for p in pubs), b"".join(p[33:65] for p in pubs)
    prefix = bytes(2 + (p[64] & 1) for p in pubs)

    def run(op, x, y, pre):
        out = ctypes.create_string_buffer(20 * n)
        assert dev.coredev_hash(op, n, x, y, pre, out) == 0
        return [out.raw[20 * i:20 * i + 20] for i in range(n)]

    h33 = run(H_PUB33, xs, None, prefix)
    assert h33 == [vo.hash160(bytes([prefix[i]]) + pubs[i][1:33]) for i in range(n)]
    assert run(H_PUB65, xs, ys, None) == [vo.hash160(p) for p in pubs]
    assert run(H_SCRIPT22, b"".join(h + bytes(12) for h in h33), None, None) == [vo.hash160(b"\x00\x14" + h) for h in h33]
    assert run(H_KECCAK, xs, ys, None) == [vo.keccak256(p[1:])[12:] for p in pubs]
