// rnd.h — the counter-based scalar stream of the "independent random key per candidate" mode, single source for the
// HIP kernels and the host (match re-derivation, CPU tests).
//
// The reference's CPU hot loop draws 32 bytes per candidate from a thread-local StdRng seeded from OS entropy
// (src/scanner.rs:144-152) and skips draws that are not valid scalars (SecretKey::from_slice, src/address.rs:93).  A
// device cannot share one sequential generator among a million lanes, and a reproducible run needs a seed the reference
// does not have; so the stream is counter-based: candidate `index` of stream `stream` under `seed` is
//     key = SHA-256("vgen-mi355x-rand" || seed[24] || u32le(stream) || u64le(index))           (52 bytes, one block)
// read as a big-endian scalar; 0 and values >= n are skipped exactly like the reference's invalid draws.  The oracle
// restates the same function (oracle/vo_scan.c: random_key; stream = its worker thread), so a device scan and the
// oracle's walk test identical private keys in identical order.
//
// The seed is 24 BYTES (round 4; it was a u64).  Every key this mode returns is a function of (seed, stream, index), and
// stream and index are small: a key found under a 64-bit seed carries ~64 bits of secret however wide its scalar looks —
// whoever knows the address can enumerate seeds and replay the search (the Profanity flaw with a wider seed).  An unseeded
// scan therefore draws 192 bits of OS entropy (the reference seeds a 256-bit StdRng from the OS, src/scanner.rs:144); the
// 64-bit `seed` of the C ABI — bytes 0..7 of the 24, little-endian, the rest zero — is for reproducible runs and tests, NOT
// for keys that will hold value.  24 bytes rather than 32 keep the message in ONE SHA-256 block: the draw costs the device
// what it did before.
#pragma once
#include "hash.h"

namespace vg {

// k: the scalar as eight little-endian words (k[0] least significant).  Validity (0 < k < n) is the caller's test.
struct RndSeed {
    u32 w[6];   // the 24 seed bytes as six little-endian words (w[0] = bytes 0..3)
};

// the u64 seeds of the C ABI: bytes 0..7, little-endian; bytes 8..23 zero
VG_HD RndSeed rnd_seed_from_u64(unsigned long long seed) {
    RndSeed s;
    s.w[0] = (u32)seed;
    s.w[1] = (u32)(seed >> 32);
    s.w[2] = s.w[3] = s.w[4] = s.w[5] = 0;
    return s;
}

VG_HD void rnd_scalar(const RndSeed &seed, u32 stream, u32 index_lo, u32 index_hi, u32 k[8]) {
    u32 w[16];
    w[0] = 0x7667656eu;   // "vgen"
    w[1] = 0x2d6d6933u;   // "-mi3"
    w[2] = 0x3535782du;   // "55x-"
    w[3] = 0x72616e64u;   // "rand"
#pragma unroll
    for (int i = 0; i < 6; i++) w[4 + i] = bswap32(seed.w[i]);
    w[10] = bswap32(stream);
    w[11] = bswap32(index_lo);
    w[12] = bswap32(index_hi);
    w[13] = 0x80000000u;
    w[14] = 0;
    w[15] = 52 * 8;
    u32 st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = SHA256_IV[i];
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = st[7 - i];
}

}  // namespace vg
