#!/usr/bin/env python3
"""Generate tests/golden/openssl_keys.json with OpenSSL libcrypto as an independent second source.

Run in the BUILD container only (needs libcrypto.so.3 with the secp256k1 curve and the deprecated
one-shot RIPEMD160()).  The output is committed; the GPU box and the test-suite only read the JSON.

For each scalar k (edge cases + seeded random): pub65 = k*G from EC_POINT_mul, and
hash160 = RIPEMD160(SHA256(.)) of the compressed and uncompressed SEC1 encodings.
Nothing here imports the oracle or the product library: it pins them from outside.
"""
import ctypes
import ctypes.util
import hashlib
import json
import os
import random

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
NID_secp256k1 = 714

lib = ctypes.CDLL(ctypes.util.find_library("crypto") or "libcrypto.so.3")
lib.EC_GROUP_new_by_curve_name.restype = ctypes.c_void_p
lib.EC_POINT_new.restype = ctypes.c_void_p
lib.EC_POINT_new.argtypes = [ctypes.c_void_p]
lib.BN_bin2bn.restype = ctypes.c_void_p
lib.BN_bin2bn.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
lib.BN_CTX_new.restype = ctypes.c_void_p
lib.BN_free.argtypes = [ctypes.c_void_p]
lib.EC_POINT_mul.argtypes = [ctypes.c_void_p] * 6
lib.EC_POINT_point2oct.restype = ctypes.c_size_t
lib.EC_POINT_point2oct.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
lib.RIPEMD160.restype = ctypes.c_void_p
lib.RIPEMD160.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]

group = lib.EC_GROUP_new_by_curve_name(NID_secp256k1)
assert group, "secp256k1 not available in this libcrypto"
bnctx = lib.BN_CTX_new()


def pub65(k: int) -> bytes:
    kb = k.to_bytes(32, "big")
    bn = lib.BN_bin2bn(kb, 32, None)
    pt = lib.EC_POINT_new(group)
    assert lib.EC_POINT_mul(group, pt, bn, None, None, bnctx) == 1
    buf = ctypes.create_string_buffer(65)
    n = lib.EC_POINT_point2oct(group, pt, 4, buf, 65, bnctx)
    assert n == 65
    lib.BN_free(bn)
    return buf.raw


def rmd160(b: bytes) -> bytes:
    out = ctypes.create_string_buffer(20)
    lib.RIPEMD160(b, len(b), out)
    return out.raw


def h160(b: bytes) -> bytes:
    return rmd160(hashlib.sha256(b).digest())


def main():
    assert rmd160(b"").hex() == "9c1185a5c5e9fc54612808977ee8f548b2258d31"
    assert rmd160(b"abc").hex() == "8eb208f7e05d987a9b044a8e98c6b087f15a0bfc"
    rng = random.Random(0x76_67_65_6E)  # "vgen"
    edge = [1, 2, 3, 15, 16, 17, 255, 256, 0xFFFF, 2**32 - 1, 2**32, 2**64 - 1, 2**64, 2**65,
            2**66 - 1, 2**128, 2**255, N - 1, N - 2, N - 3, N // 2, N // 2 + 1,
            int.from_bytes(hashlib.sha256(b"correct horse battery staple").digest(), "big"),
            0x0C28FCA386C7A227600B2FE50B7CAE11EC86D3BF1FBE471BE89827E19D72AA1D]
    edge += [2**k for k in range(8, 256, 31)] + [2**k - 1 for k in range(9, 256, 37)]
    full, short = [], []
    for k in edge + [rng.randrange(1, N) for _ in range(40)]:
        p = pub65(k)
        c = bytes([2 + (p[64] & 1)]) + p[1:33]
        full.append({"key": f"{k:064x}", "pub65": p.hex(), "h160c": h160(c).hex(),
                     "h160u": h160(p).hex()})
    # runs of consecutive scalars (sequential mode) and more random ones, hash160 only
    bases = [rng.randrange(1, N - 64) for _ in range(6)] + [2**65, N - 40]
    for b in bases:
        for i in range(32):
            k = b + i
            if not 0 < k < N:
                continue
            p = pub65(k)
            c = bytes([2 + (p[64] & 1)]) + p[1:33]
            short.append([f"{k:064x}", h160(c).hex(), h160(p).hex()])
    for _ in range(200):
        k = rng.randrange(1, N)
        p = pub65(k)
        c = bytes([2 + (p[64] & 1)]) + p[1:33]
        short.append([f"{k:064x}", h160(c).hex(), h160(p).hex()])
    out = {"source": "OpenSSL libcrypto EC_POINT_mul(secp256k1) + SHA256 + RIPEMD160",
           "full": full, "short_fields": ["key", "h160c", "h160u"], "short": short}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "openssl_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes;", len(full), "full,", len(short), "short")


if __name__ == "__main__":
    main()
