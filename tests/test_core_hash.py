"""CPU tests of the single-source hash block functions (vgen_amd/csrc/core/hash.h) against the oracle."""
import ctypes
import os
import random
import subprocess

import pytest

from conftest import locked_make

from oracle import pyoracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


@pytest.fixture(scope="module")
def core():
    locked_make("-s", "-C", os.path.join(HERE, "native"))
    return ctypes.CDLL(os.path.join(HERE, "native", "libcoretest.so"))


def pubs():
    rng = random.Random(21)
    keys = [1, 2, 3, N - 1, 2**255, 0xFF] + [rng.randrange(1, N) for _ in range(60)]
    return [vo.pubkey(k) for k in keys]


def test_hash160_compressed(core):
    for p in pubs():
        out = ctypes.create_string_buffer(20)
        core.core_hash160_pub33(2 + (p[64] & 1), p[1:33], out)
        assert out.raw == vo.hash160(bytes([2 + (p[64] & 1)]) + p[1:33])


def test_hash160_uncompressed(core):
    for p in pubs():
        out = ctypes.create_string_buffer(20)
        core.core_hash160_pub65(p[1:33], p[33:65], out)
        assert out.raw == vo.hash160(p)


def test_hash160_p2sh_script(core):
    for p in pubs():
        h = vo.hash160(bytes([2 + (p[64] & 1)]) + p[1:33])
        out = ctypes.create_string_buffer(20)
        core.core_hash160_script22(h, out)
        assert out.raw == vo.hash160(b"\x00\x14" + h)


def test_keccak_address(core):
    for p in pubs():
        out = ctypes.create_string_buffer(20)
        core.core_keccak_addr(p[1:33], p[33:65], out)
        assert out.raw == vo.keccak256(p[1:])[12:]


def test_host_sha256_portable_and_sha_extension_paths_agree_with_hashlib(core):
    """host_sha256 (host/encode.cpp) — what the host's Base58Check checksums run through, four blocks per match — by both of its
    block functions: the portable one (the single-source core/hash.h, with the host's plain-expression truth tables) and the x86 SHA
    extensions where the CPU has them (run-time CPUID; the call with allow = 1 falls back by itself elsewhere).  Every length
    around the one- and two-block padding boundaries, and long messages."""
    import hashlib
    rng = random.Random(5)
    for length in list(range(0, 200)) + [255, 256, 257, 1000, 4096, 65537]:
        for _ in range(3):
            m = rng.randbytes(length)
            want = hashlib.sha256(m).digest()
            for allow in (0, 1):
                out = ctypes.create_string_buffer(32)
                core.core_host_sha256(m, ctypes.c_ulong(length), out, allow)
                assert out.raw == want, (length, allow)
