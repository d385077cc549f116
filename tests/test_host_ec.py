"""CPU tests of the host-side EC code of libvgen_hip.so (fixed-base multiplication, stride tables)
against the oracle and the OpenSSL fixtures."""
import ctypes
import json
import os
import random
import subprocess

import pytest

from conftest import locked_make

from oracle import pyoracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
OSSL = json.load(open(os.path.join(HERE, "golden", "openssl_keys.json")))
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


@pytest.fixture(scope="module")
def core():
    locked_make("-s", "-C", os.path.join(HERE, "native"))
    lib = ctypes.CDLL(os.path.join(HERE, "native", "libcoretest.so"))
    lib.core_stride_table.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p]
    return lib


def test_mul_gen_against_openssl_fixtures(core):
    for vec in OSSL["full"]:
        out = ctypes.create_string_buffer(64)
        assert core.core_mul_gen(bytes.fromhex(vec["key"]), out) == 1
        assert out.raw.hex() == vec["pub65"][2:]


def test_mul_gen_random_against_oracle(core):
    rng = random.Random(31)
    for _ in range(100):
        k = rng.randrange(1, N)
        out = ctypes.create_string_buffer(64)
        assert core.core_mul_gen(k.to_bytes(32, "big"), out) == 1
        assert out.raw == vo.pubkey(k)[1:]
    out = ctypes.create_string_buffer(64)
    assert core.core_mul_gen(bytes(32), out) == 0
    assert core.core_mul_gen(N.to_bytes(32, "big"), out) == 0  # n*G = infinity


def test_stride_table(core):
    first, step, count = 8, 16, 300
    buf = ctypes.create_string_buffer(64 * count)
    core.core_stride_table(first, step, count, buf)
    for e in list(range(0, 20)) + [count - 1, 137, 256]:
        assert buf.raw[64 * e:64 * e + 64] == vo.pubkey(first + e * step)[1:]
    # a chunk boundary (4096) is crossed here
    count = 4100
    buf = ctypes.create_string_buffer(64 * count)
    core.core_stride_table(1, 1, count, buf)
    for e in (0, 1, 4094, 4095, 4096, 4097, 4099):
        assert buf.raw[64 * e:64 * e + 64] == vo.pubkey(1 + e)[1:]


def test_branch_free_window_accumulation_used_by_keys_kernel(core):
    """gej_add_ge_nz (no special-case handling) is safe for low-to-high unsigned windows of a valid key."""
    rng = random.Random(77)
    keys = [1, 2, 15, 16, 17, 0xF0, 0x100, N - 1, N - 2, 2**255, 0x1111111111111111, (1 << 252) - 1] + \
           [rng.randrange(1, N) for _ in range(12)]
    for k in keys:
        out = ctypes.create_string_buffer(64)
        assert core.core_mul_windows_nz(k.to_bytes(32, "big"), out) == 1
        assert out.raw == vo.pubkey(k)[1:], hex(k)


def test_taproot_output_key_device_algorithm(core):
    """core/taproot.h (the code the P2TR kernels run) against the oracle's BIP-341 restatement."""
    rng = random.Random(99)
    keys = [1, 2, 3, N - 1, 0x0C28FCA386C7A227600B2FE50B7CAE11EC86D3BF1FBE471BE89827E19D72AA1D] + \
           [rng.randrange(1, N) for _ in range(25)]
    for k in keys:
        out = ctypes.create_string_buffer(32)
        assert core.core_taproot_from_key(k.to_bytes(32, "big"), out) == 1
        assert out.raw == vo.payload(vo.FMT_P2TR, k), hex(k)
    assert vo.segwit_addr("bc", 1, vo.payload(vo.FMT_P2TR, 1)) == "bc1pmfr3p9j00pfxjh0zmgp99y8zftmd3s5pmedqhyptwy6lm87hf5sspknck9"


def test_seq_base_points_cache_walks_jumps_and_rewinds(core):
    """host_seq_points: per-dispatch uniform points (kb + j)*G from the cache — single incremental steps, the
    eight-dispatch look-ahead once the stride has repeated, a change of stride, a jump and a rewind."""
    S = 8
    base = 0x3a8ae174e51b7b1117ab406c6570970f453c4376b6d381977db7c02fb5a993e0
    stride = 1 << 20
    walk = [base + i * stride for i in range(30)]                  # fills and drains the look-ahead three times
    walk += [walk[-1] + 3 * stride * (i + 1) for i in range(12)]   # another stride
    walk += [walk[-1] + (1 << 70), walk[-1] + (1 << 70) + stride]  # a jump too large for the incremental path
    walk += [base + 5 * stride + i * stride for i in range(12)]    # rewind, then the old stride again
    walk += [N - 100, 1, 2, 3]
    blob = b"".join(k.to_bytes(32, "big") for k in walk)
    out = ctypes.create_string_buffer(64 * S * len(walk))
    assert core.core_seq_points_walk(blob, len(walk), S, out) == len(walk)
    for c, kb in enumerate(walk):
        for j in (0, 1, S - 1):
            assert out.raw[64 * (c * S + j):64 * (c * S + j) + 64] == vo.pubkey(kb + j)[1:], (c, j)


def test_sixteen_bit_window_multiplication_used_by_keys_and_taproot_kernels(core):
    """ec_mul_gen_w16 (15 branch-free mixed additions over the 16-bit table the device builds itself): random keys,
    keys with zero digits (skipped windows, late first non-zero digit), single-digit keys, n - 1."""
    rng = random.Random(77)
    keys = [1, 2, 0xFFFF, 0x10000, 0x10001, 2**16 * 0xABCD, 2**240, 2**255 + 1, N - 1, N - 2, 0xFFFF << 112, (1 << 200) + (1 << 16)]
    keys += [rng.randrange(1, N) for _ in range(60)]
    keys += [rng.randrange(1, 2**64) << (16 * rng.randrange(0, 12)) for _ in range(20)]
    for k in keys:
        out = ctypes.create_string_buffer(64)
        assert core.core_mul_w16((k % N).to_bytes(32, "big"), out) == 1, hex(k)
        assert out.raw == vo.pubkey(k % N)[1:], hex(k)


def signed_window_edge_keys(st, rng):
    """Scalars that walk the corners of the signed-window recoding (core/ec.h: ec_mul_gen_signed): digits exactly at, one below
    and one above the sign threshold 2^(st-1), carries that ripple through all-ones windows, zero digits after a carry, the
    top window's largest magnitude (which stands for the scalar 2^256 itself: taken mod n), and n - 1."""
    half = 1 << (st - 1)
    nw = -(-257 // st)
    keys = [1, 2, half - 1, half, half + 1, (1 << st) - 1, 1 << st, (1 << st) + 1, N - 1, N - 2, 2**255, 2**255 + 1, 2**256 - 2**129]
    for w in range(nw - 1):
        for d in (half - 1, half, half + 1, (1 << st) - 1):
            keys.append(d << (st * w))
            keys.append((d << (st * w)) + 1)
    # all windows all-ones below some point: a carry that ripples to the top
    for w in range(1, nw):
        keys.append((1 << (st * w)) - 1)
        keys.append(((1 << (st * w)) - 1) ^ (1 << (st * (w - 1))))
    top = st * (nw - 1)
    # raw top digit all ones + a carry from the window below: magnitude 2^(256 - top) = the scalar 2^256
    keys.append((1 << 256) - (1 << top) + (half + 1) * (1 << (top - st)) + 5)
    keys.append((1 << 256) - (1 << top) + ((1 << st) - 1) * (1 << (top - st)))
    keys += [rng.randrange(1, N) for _ in range(40)]
    keys += [rng.randrange(1, 2**64) << rng.randrange(0, 192) for _ in range(20)]
    return [k for k in keys if 0 < k < N]


@pytest.mark.parametrize("st", [25, 27, 29])
def test_signed_window_multiplication(core, st):
    """ec_mul_gen_signed<ST>: windows of ST bits with digits in [-(2^(ST-1) - 1), 2^(ST-1)], a table of magnitudes, (x, p - y) for
    negative digits — 8 additions per multiplication at 29 bits where the unsigned 24-bit table needs 10.  The table is mapped
    without backing store (up to 138 GB of address space) and filled only where the key walks; results against the oracle."""
    rng = random.Random(st)
    got_any = False
    for k in signed_window_edge_keys(st, rng):
        out = ctypes.create_string_buffer(64)
        rc = core.core_mul_signed(st, k.to_bytes(32, "big"), out)
        if rc == -1 and not got_any:
            pytest.skip(f"cannot map {st}-bit table's address space here")
        assert rc == 1, (st, hex(k), rc)
        got_any = True
        assert out.raw == vo.pubkey(k)[1:], (st, hex(k))
