// rnd.h — the counter-based scalar stream of the "independent random key per candidate" mode, single source for the
// HIP kernels and the host (match re-derivation, CPU tests).
//
// The reference's CPU hot loop draws 32 bytes per candidate from a thread-local StdRng seeded from OS entropy
// (src/scanner.rs:144-152) and skips draws that are not valid scalars (SecretKey::from_slice, src/address.rs:93).  A
// device cannot share one sequential generator among a million lanes, and a reproducible run needs a seed the reference
// does not have; so the stream is counter-based: candidate `index` of stream `stream` under `seed` is
//     key = SHA-256("vgen-mi355x-rand" || u64le(seed) || u32le(stream) || u64le(index))        (36 bytes, one block)
// read as a big-endian scalar; 0 and values >= n are skipped exactly like the reference's invalid draws.  The oracle
// restates the same function (oracle/vo_scan.c: random_key; stream = its worker thread), so a device scan and the
// oracle's walk test identical private keys in identical order.
#pragma once
#include "hash.h"

namespace vg {

// k: the scalar as eight little-endian words (k[0] least significant).  Validity (0 < k < n) is the caller's test.
VG_HD void rnd_scalar(u32 seed_lo, u32 seed_hi, u32 stream, u32 index_lo, u32 index_hi, u32 k[8]) {
    u32 w[16];
    w[0] = 0x7667656eu;   // "vgen"
    w[1] = 0x2d6d6933u;   // "-mi3"
    w[2] = 0x3535782du;   // "55x-"
    w[3] = 0x72616e64u;   // "rand"
    w[4] = bswap32(seed_lo);
    w[5] = bswap32(seed_hi);
    w[6] = bswap32(stream);
    w[7] = bswap32(index_lo);
    w[8] = bswap32(index_hi);
    w[9] = 0x80000000u;
#pragma unroll
    for (int i = 10; i < 15; i++) w[i] = 0;
    w[15] = 36 * 8;
    u32 st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = SHA256_IV[i];
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = st[7 - i];
}

}  // namespace vg
