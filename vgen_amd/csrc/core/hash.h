// hash.h — SHA-256, RIPEMD-160 and Keccak-f[1600] block functions, single source for the HIP
// kernels (hipcc, gfx950) and the host side / CPU tests (g++).
//
// These replace, on the device, what the reference computes in src/shaders/sha256.wgsl:43-170
// (one-block SHA-256 of the 33-byte compressed key) and src/shaders/ripemd160.wgsl:10-100
// (one-block RIPEMD-160 of the 32-byte digest), and add the Keccak-256 the reference only has on
// the CPU (src/address.rs:100-102).  Everything is written as straight-line code over registers:
// no lookup tables for the RIPEMD message order / rotation amounts (the WGSL indexes arrays at run
// time), a rolling 16-word SHA-256 schedule instead of w[64], and boolean functions written so that
// hipcc selects v_bitop3_b32 (any 3-input boolean, full rate on gfx950) and v_alignbit_b32.
#pragma once
#include <stdint.h>

#include "fe.h"

namespace vg {

VG_HD u32 rotr32(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
VG_HD u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
VG_HD u32 bswap32(u32 x) {
    return (x >> 24) | ((x >> 8) & 0x0000FF00u) | ((x << 8) & 0x00FF0000u) | (x << 24);
}

// Three-input boolean functions.  On the device they are spelled as v_bitop3_b32 (one full-rate
// instruction for ANY 3-input truth table on gfx950); hipcc finds the and/or/not forms by itself but
// leaves x^y^z of rotates as two v_xor, so the xor3 of the sigma functions is made explicit.  The truth
// table index is (a << 2 | b << 1 | c).  On the host the same functions are plain C.
#if defined(__HIP_DEVICE_COMPILE__)
#define VG_BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
#else
VG_HD u32 vg_bitop3_host(u32 a, u32 b, u32 c, u32 tt) {
    // the tables this file uses, as plain expressions (the generic evaluation below costs ~40 operations per call and made the
    // host's SHA-256 four times slower than it needs to be: every match is hashed four times on the host)
    switch (tt) {
    case 0x96: return a ^ b ^ c;                    // xor3
    case 0xCA: return (a & b) | (~a & c);           // SHA-256 Ch, RIPEMD F2
    case 0xE8: return (a & b) | (c & (a | b));      // SHA-256 Maj
    case 0x59: return (a | ~b) ^ c;                 // RIPEMD F3
    case 0xE4: return (a & c) | (b & ~c);           // RIPEMD F4
    case 0x2D: return a ^ (b | ~c);                 // RIPEMD F5
    case 0xD2: return a ^ (~b & c);                 // Keccak chi
    default: break;
    }
    u32 r = 0;
    for (int i = 0; i < 8; i++)
        if ((tt >> i) & 1) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
}
#define VG_BITOP3(a, b, c, tt) vg_bitop3_host((a), (b), (c), (tt))
#endif
#define VG_XOR3(a, b, c) VG_BITOP3(a, b, c, 0x96)
#define VG_CH(e, f, g) VG_BITOP3(e, f, g, 0xCA)     // (e & f) ^ (~e & g)
#define VG_MAJ(a, b, c) VG_BITOP3(a, b, c, 0xE8)    // majority

// ---- SHA-256 ---------------------------------------------------------------------------------

constexpr u32 SHA256_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                              0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

#define VG_SHA_ROUND(a, b, c, d, e, f, g, h, k, w)                                    \
    {                                                                                 \
        u32 t1_ = h + VG_XOR3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)) + VG_CH(e, f, g) + (k) + (w); \
        u32 t2_ = VG_XOR3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)) + VG_MAJ(a, b, c);                 \
        d += t1_;                                                                     \
        h = t1_ + t2_;                                                                \
    }
#define VG_SHA_S0(x) VG_XOR3(rotr32(x, 7), rotr32(x, 18), ((x) >> 3))
#define VG_SHA_S1(x) VG_XOR3(rotr32(x, 17), rotr32(x, 19), ((x) >> 10))
#define VG_SHA_SCHED(i) (w[(i) & 15] += VG_SHA_S1(w[((i) - 2) & 15]) + w[((i) - 7) & 15] + VG_SHA_S0(w[((i) - 15) & 15]))

#define VG_SHA_8ROUNDS(base, W)                                            \
    VG_SHA_ROUND(a, b, c, d, e, f, g, h, VG_SHA_K_(base, 0), W(base + 0)) \
    VG_SHA_ROUND(h, a, b, c, d, e, f, g, VG_SHA_K_(base, 1), W(base + 1)) \
    VG_SHA_ROUND(g, h, a, b, c, d, e, f, VG_SHA_K_(base, 2), W(base + 2)) \
    VG_SHA_ROUND(f, g, h, a, b, c, d, e, VG_SHA_K_(base, 3), W(base + 3)) \
    VG_SHA_ROUND(e, f, g, h, a, b, c, d, VG_SHA_K_(base, 4), W(base + 4)) \
    VG_SHA_ROUND(d, e, f, g, h, a, b, c, VG_SHA_K_(base, 5), W(base + 5)) \
    VG_SHA_ROUND(c, d, e, f, g, h, a, b, VG_SHA_K_(base, 6), W(base + 6)) \
    VG_SHA_ROUND(b, c, d, e, f, g, h, a, VG_SHA_K_(base, 7), W(base + 7))

constexpr u32 SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define VG_SHA_K_(base, i) SHA256_K[(base) + (i)]

// One SHA-256 compression.  st: chaining value (in/out), w: 16 big-endian message words (clobbered).
VG_HD void sha256_compress(u32 st[8], u32 w[16]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#define VG_W_DIRECT(i) w[(i) & 15]
#define VG_W_SCHED(i) VG_SHA_SCHED(i)
    VG_SHA_8ROUNDS(0, VG_W_DIRECT)
    VG_SHA_8ROUNDS(8, VG_W_DIRECT)
    VG_SHA_8ROUNDS(16, VG_W_SCHED)
    VG_SHA_8ROUNDS(24, VG_W_SCHED)
    VG_SHA_8ROUNDS(32, VG_W_SCHED)
    VG_SHA_8ROUNDS(40, VG_W_SCHED)
    VG_SHA_8ROUNDS(48, VG_W_SCHED)
    VG_SHA_8ROUNDS(56, VG_W_SCHED)
#undef VG_W_DIRECT
#undef VG_W_SCHED
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// SHA-256 of the 33-byte compressed SEC1 key: prefix (0x02/0x03) || X.  xw: X as eight 32-bit words,
// xw[0] least significant (fe_to_words).  out: the eight big-endian digest words.
VG_HD void sha256_pub33(u32 prefix, const u32 xw[8], u32 out[8]) {
    u32 w[16];
    w[0] = (prefix << 24) | (xw[7] >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (xw[8 - i] << 24) | (xw[7 - i] >> 8);
    w[8] = (xw[0] << 24) | 0x00800000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 33 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = SHA256_IV[i];
    sha256_compress(out, w);
}

// SHA-256 of the 65-byte uncompressed SEC1 key 0x04 || X || Y (two blocks).
VG_HD void sha256_pub65(const u32 xw[8], const u32 yw[8], u32 out[8]) {
    u32 w[16];
    w[0] = (0x04u << 24) | (xw[7] >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (xw[8 - i] << 24) | (xw[7 - i] >> 8);
    w[8] = (xw[0] << 24) | (yw[7] >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[8 + i] = (yw[8 - i] << 24) | (yw[7 - i] >> 8);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = SHA256_IV[i];
    sha256_compress(out, w);
    w[0] = (yw[0] << 24) | 0x00800000u;
#pragma unroll
    for (int i = 1; i < 15; i++) w[i] = 0;
    w[15] = 65 * 8;
    sha256_compress(out, w);
}

// SHA-256 of the 22-byte P2WPKH redeem script 0x00 0x14 || h160 (h160 as five LITTLE-endian words,
// i.e. RIPEMD-160's native output order).
VG_HD void sha256_script22(const u32 h[5], u32 out[8]) {
    u32 b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) b[i] = bswap32(h[i]);   // big-endian words of the 20 bytes
    u32 w[16];
    w[0] = 0x00140000u | (b[0] >> 16);
    w[1] = (b[0] << 16) | (b[1] >> 16);
    w[2] = (b[1] << 16) | (b[2] >> 16);
    w[3] = (b[2] << 16) | (b[3] >> 16);
    w[4] = (b[3] << 16) | (b[4] >> 16);
    w[5] = (b[4] << 16) | 0x00008000u;
#pragma unroll
    for (int i = 6; i < 15; i++) w[i] = 0;
    w[15] = 22 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = SHA256_IV[i];
    sha256_compress(out, w);
}

// BIP-341 TapTweak for a key-path-only output: t = SHA-256(SHA-256("TapTweak") || SHA-256("TapTweak") || x).
// The first 64-byte block is constant; its chaining value is the midstate the reference also carries
// (src/shaders/sha256.wgsl:180-183, cross-checked in tests/test_oracle_golden.py).  xw: x as eight words,
// xw[0] least significant; out: t as eight BIG-endian words (out[0] most significant).
constexpr u32 TAPTWEAK_MIDSTATE[8] = {0xd129a2f3u, 0x701c655du, 0x6583b6c3u, 0xb9419727u,
                                      0x95f4e232u, 0x94fd54f4u, 0xa2ae8d85u, 0x47ca590bu};

VG_HD void sha256_taptweak(const u32 xw[8], u32 out[8]) {
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = xw[7 - i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 96 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = TAPTWEAK_MIDSTATE[i];
    sha256_compress(out, w);
}

// ---- RIPEMD-160 --------------------------------------------------------------------------------

#define VG_RMD_F1(x, y, z) VG_XOR3(x, y, z)
#define VG_RMD_F2(x, y, z) VG_BITOP3(x, y, z, 0xCA)   // (x & y) | (~x & z)
#define VG_RMD_F3(x, y, z) VG_BITOP3(x, y, z, 0x59)   // (x | ~y) ^ z
#define VG_RMD_F4(x, y, z) VG_BITOP3(x, y, z, 0xE4)   // (x & z) | (y & ~z)
#define VG_RMD_F5(x, y, z) VG_BITOP3(x, y, z, 0x2D)   // x ^ (y | ~z)
#define VG_RMD_STEP(F, a, b, c, d, e, x, k, s)  \
    {                                           \
        a += F(b, c, d) + (x) + (k);            \
        a = rotl32(a, s) + e;                   \
        c = rotl32(c, 10);                      \
    }

// One RIPEMD-160 compression of the 16 little-endian words x into st (in/out).
VG_HD void ripemd160_compress(u32 st[5], const u32 x[16]) {
    u32 al = st[0], bl = st[1], cl = st[2], dl = st[3], el = st[4];
    u32 ar = al, br = bl, cr = cl, dr = dl, er = el;
#define L1(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F1, a, b, c, d, e, x[i], 0x00000000u, s)
#define L2(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F2, a, b, c, d, e, x[i], 0x5A827999u, s)
#define L3(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F3, a, b, c, d, e, x[i], 0x6ED9EBA1u, s)
#define L4(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F4, a, b, c, d, e, x[i], 0x8F1BBCDCu, s)
#define L5(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F5, a, b, c, d, e, x[i], 0xA953FD4Eu, s)
#define R1(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F5, a, b, c, d, e, x[i], 0x50A28BE6u, s)
#define R2(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F4, a, b, c, d, e, x[i], 0x5C4DD124u, s)
#define R3(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F3, a, b, c, d, e, x[i], 0x6D703EF3u, s)
#define R4(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F2, a, b, c, d, e, x[i], 0x7A6D76E9u, s)
#define R5(a, b, c, d, e, i, s) VG_RMD_STEP(VG_RMD_F1, a, b, c, d, e, x[i], 0x00000000u, s)
    // left line
    L1(al, bl, cl, dl, el, 0, 11) L1(el, al, bl, cl, dl, 1, 14) L1(dl, el, al, bl, cl, 2, 15) L1(cl, dl, el, al, bl, 3, 12)
    L1(bl, cl, dl, el, al, 4, 5)  L1(al, bl, cl, dl, el, 5, 8)  L1(el, al, bl, cl, dl, 6, 7)  L1(dl, el, al, bl, cl, 7, 9)
    L1(cl, dl, el, al, bl, 8, 11) L1(bl, cl, dl, el, al, 9, 13) L1(al, bl, cl, dl, el, 10, 14) L1(el, al, bl, cl, dl, 11, 15)
    L1(dl, el, al, bl, cl, 12, 6) L1(cl, dl, el, al, bl, 13, 7) L1(bl, cl, dl, el, al, 14, 9) L1(al, bl, cl, dl, el, 15, 8)
    L2(el, al, bl, cl, dl, 7, 7)  L2(dl, el, al, bl, cl, 4, 6)  L2(cl, dl, el, al, bl, 13, 8) L2(bl, cl, dl, el, al, 1, 13)
    L2(al, bl, cl, dl, el, 10, 11) L2(el, al, bl, cl, dl, 6, 9) L2(dl, el, al, bl, cl, 15, 7) L2(cl, dl, el, al, bl, 3, 15)
    L2(bl, cl, dl, el, al, 12, 7) L2(al, bl, cl, dl, el, 0, 12) L2(el, al, bl, cl, dl, 9, 15) L2(dl, el, al, bl, cl, 5, 9)
    L2(cl, dl, el, al, bl, 2, 11) L2(bl, cl, dl, el, al, 14, 7) L2(al, bl, cl, dl, el, 11, 13) L2(el, al, bl, cl, dl, 8, 12)
    L3(dl, el, al, bl, cl, 3, 11) L3(cl, dl, el, al, bl, 10, 13) L3(bl, cl, dl, el, al, 14, 6) L3(al, bl, cl, dl, el, 4, 7)
    L3(el, al, bl, cl, dl, 9, 14) L3(dl, el, al, bl, cl, 15, 9) L3(cl, dl, el, al, bl, 8, 13) L3(bl, cl, dl, el, al, 1, 15)
    L3(al, bl, cl, dl, el, 2, 14) L3(el, al, bl, cl, dl, 7, 8)  L3(dl, el, al, bl, cl, 0, 13) L3(cl, dl, el, al, bl, 6, 6)
    L3(bl, cl, dl, el, al, 13, 5) L3(al, bl, cl, dl, el, 11, 12) L3(el, al, bl, cl, dl, 5, 7) L3(dl, el, al, bl, cl, 12, 5)
    L4(cl, dl, el, al, bl, 1, 11) L4(bl, cl, dl, el, al, 9, 12) L4(al, bl, cl, dl, el, 11, 14) L4(el, al, bl, cl, dl, 10, 15)
    L4(dl, el, al, bl, cl, 0, 14) L4(cl, dl, el, al, bl, 8, 15) L4(bl, cl, dl, el, al, 12, 9) L4(al, bl, cl, dl, el, 4, 8)
    L4(el, al, bl, cl, dl, 13, 9) L4(dl, el, al, bl, cl, 3, 14) L4(cl, dl, el, al, bl, 7, 5)  L4(bl, cl, dl, el, al, 15, 6)
    L4(al, bl, cl, dl, el, 14, 8) L4(el, al, bl, cl, dl, 5, 6)  L4(dl, el, al, bl, cl, 6, 5)  L4(cl, dl, el, al, bl, 2, 12)
    L5(bl, cl, dl, el, al, 4, 9)  L5(al, bl, cl, dl, el, 0, 15) L5(el, al, bl, cl, dl, 5, 5)  L5(dl, el, al, bl, cl, 9, 11)
    L5(cl, dl, el, al, bl, 7, 6)  L5(bl, cl, dl, el, al, 12, 8) L5(al, bl, cl, dl, el, 2, 13) L5(el, al, bl, cl, dl, 10, 12)
    L5(dl, el, al, bl, cl, 14, 5) L5(cl, dl, el, al, bl, 1, 12) L5(bl, cl, dl, el, al, 3, 13) L5(al, bl, cl, dl, el, 8, 14)
    L5(el, al, bl, cl, dl, 11, 11) L5(dl, el, al, bl, cl, 6, 8) L5(cl, dl, el, al, bl, 15, 5) L5(bl, cl, dl, el, al, 13, 6)
    // right line
    R1(ar, br, cr, dr, er, 5, 8)  R1(er, ar, br, cr, dr, 14, 9) R1(dr, er, ar, br, cr, 7, 9)  R1(cr, dr, er, ar, br, 0, 11)
    R1(br, cr, dr, er, ar, 9, 13) R1(ar, br, cr, dr, er, 2, 15) R1(er, ar, br, cr, dr, 11, 15) R1(dr, er, ar, br, cr, 4, 5)
    R1(cr, dr, er, ar, br, 13, 7) R1(br, cr, dr, er, ar, 6, 7)  R1(ar, br, cr, dr, er, 15, 8) R1(er, ar, br, cr, dr, 8, 11)
    R1(dr, er, ar, br, cr, 1, 14) R1(cr, dr, er, ar, br, 10, 14) R1(br, cr, dr, er, ar, 3, 12) R1(ar, br, cr, dr, er, 12, 6)
    R2(er, ar, br, cr, dr, 6, 9)  R2(dr, er, ar, br, cr, 11, 13) R2(cr, dr, er, ar, br, 3, 15) R2(br, cr, dr, er, ar, 7, 7)
    R2(ar, br, cr, dr, er, 0, 12) R2(er, ar, br, cr, dr, 13, 8) R2(dr, er, ar, br, cr, 5, 9)  R2(cr, dr, er, ar, br, 10, 11)
    R2(br, cr, dr, er, ar, 14, 7) R2(ar, br, cr, dr, er, 15, 7) R2(er, ar, br, cr, dr, 8, 12) R2(dr, er, ar, br, cr, 12, 7)
    R2(cr, dr, er, ar, br, 4, 6)  R2(br, cr, dr, er, ar, 9, 15) R2(ar, br, cr, dr, er, 1, 13) R2(er, ar, br, cr, dr, 2, 11)
    R3(dr, er, ar, br, cr, 15, 9) R3(cr, dr, er, ar, br, 5, 7)  R3(br, cr, dr, er, ar, 1, 15) R3(ar, br, cr, dr, er, 3, 11)
    R3(er, ar, br, cr, dr, 7, 8)  R3(dr, er, ar, br, cr, 14, 6) R3(cr, dr, er, ar, br, 6, 6)  R3(br, cr, dr, er, ar, 9, 14)
    R3(ar, br, cr, dr, er, 11, 12) R3(er, ar, br, cr, dr, 8, 13) R3(dr, er, ar, br, cr, 12, 5) R3(cr, dr, er, ar, br, 2, 14)
    R3(br, cr, dr, er, ar, 10, 13) R3(ar, br, cr, dr, er, 0, 13) R3(er, ar, br, cr, dr, 4, 7) R3(dr, er, ar, br, cr, 13, 5)
    R4(cr, dr, er, ar, br, 8, 15) R4(br, cr, dr, er, ar, 6, 5)  R4(ar, br, cr, dr, er, 4, 8)  R4(er, ar, br, cr, dr, 1, 11)
    R4(dr, er, ar, br, cr, 3, 14) R4(cr, dr, er, ar, br, 11, 14) R4(br, cr, dr, er, ar, 15, 6) R4(ar, br, cr, dr, er, 0, 14)
    R4(er, ar, br, cr, dr, 5, 6)  R4(dr, er, ar, br, cr, 12, 9) R4(cr, dr, er, ar, br, 2, 12) R4(br, cr, dr, er, ar, 13, 9)
    R4(ar, br, cr, dr, er, 9, 12) R4(er, ar, br, cr, dr, 7, 5)  R4(dr, er, ar, br, cr, 10, 15) R4(cr, dr, er, ar, br, 14, 8)
    R5(br, cr, dr, er, ar, 12, 8) R5(ar, br, cr, dr, er, 15, 5) R5(er, ar, br, cr, dr, 10, 12) R5(dr, er, ar, br, cr, 4, 9)
    R5(cr, dr, er, ar, br, 1, 12) R5(br, cr, dr, er, ar, 5, 5)  R5(ar, br, cr, dr, er, 8, 14) R5(er, ar, br, cr, dr, 7, 6)
    R5(dr, er, ar, br, cr, 6, 8)  R5(cr, dr, er, ar, br, 2, 13) R5(br, cr, dr, er, ar, 13, 6) R5(ar, br, cr, dr, er, 14, 5)
    R5(er, ar, br, cr, dr, 0, 15) R5(dr, er, ar, br, cr, 3, 13) R5(cr, dr, er, ar, br, 9, 11) R5(br, cr, dr, er, ar, 11, 11)
#undef L1
#undef L2
#undef L3
#undef L4
#undef L5
#undef R1
#undef R2
#undef R3
#undef R4
#undef R5
    u32 t = st[1] + cl + dr;
    st[1] = st[2] + dl + er;
    st[2] = st[3] + el + ar;
    st[3] = st[4] + al + br;
    st[4] = st[0] + bl + cr;
    st[0] = t;
}

constexpr u32 RMD160_IV[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};

// RIPEMD-160 of a 32-byte SHA-256 digest given as eight big-endian words.  out: five little-endian
// words == the 20 digest bytes in memory order on a little-endian machine — the layout the
// reference's kernel stores (src/shaders/ripemd160.wgsl:93-99, src/gpu.rs:644-650).
VG_HD void ripemd160_of_sha(const u32 sha[8], u32 out[5]) {
    u32 x[16];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = bswap32(sha[i]);
    x[8] = 0x00000080u;
#pragma unroll
    for (int i = 9; i < 14; i++) x[i] = 0;
    x[14] = 32 * 8;
    x[15] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) out[i] = RMD160_IV[i];
    ripemd160_compress(out, x);
}

// ---- Keccak-256 of exactly 64 bytes (Ethereum: X || Y) ---------------------------------------------

// 64-bit rotate by a constant as two 32-bit funnel shifts (v_alignbit_b32 on the device; hipcc would build it
// from two 64-bit shifts and two ors).
VG_HD u32 funnel_r32(u32 hi, u32 lo, int s) {   // low word of {hi,lo} >> s, 0 < s < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (u32)s);
#else
    return (lo >> s) | (hi << (32 - s));
#endif
}
template <int N>
VG_HD u64 rotl64(u64 x) {
    static_assert(N > 0 && N < 64, "rotation");
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    if (N == 32) return ((u64)lo << 32) | hi;
    constexpr int M = N & 31;
    const u32 a = funnel_r32(hi, lo, 32 - M);   // (hi << M) | (lo >> (32 - M))
    const u32 b = funnel_r32(lo, hi, 32 - M);   // (lo << M) | (hi >> (32 - M))
    return N < 32 ? (((u64)a << 32) | b) : (((u64)b << 32) | a);
}

// 3-input boolean on 64-bit lanes = the 32-bit v_bitop3 on each half
template <u32 TT>
VG_HD u64 bitop3_64(u64 a, u64 b, u64 c) {
    const u32 lo = VG_BITOP3((u32)a, (u32)b, (u32)c, TT);
    const u32 hi = VG_BITOP3((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), TT);
    return ((u64)hi << 32) | lo;
}

// One round of Keccak-f[1600] (theta, rho, pi, chi, iota) on the 25 lanes.
VG_HD void keccak_round(u64 a[25], u64 rc) {
        // theta: column parities as two xor3 each; the "a ^ d" of every lane is one more xor3
        // (d_x = c_{x-1} ^ rotl(c_{x+1}, 1) is never materialised)
        const u64 c0 = bitop3_64<0x96>(bitop3_64<0x96>(a[0], a[5], a[10]), a[15], a[20]);
        const u64 c1 = bitop3_64<0x96>(bitop3_64<0x96>(a[1], a[6], a[11]), a[16], a[21]);
        const u64 c2 = bitop3_64<0x96>(bitop3_64<0x96>(a[2], a[7], a[12]), a[17], a[22]);
        const u64 c3 = bitop3_64<0x96>(bitop3_64<0x96>(a[3], a[8], a[13]), a[18], a[23]);
        const u64 c4 = bitop3_64<0x96>(bitop3_64<0x96>(a[4], a[9], a[14]), a[19], a[24]);
        const u64 r0 = rotl64<1>(c0), r1 = rotl64<1>(c1), r2 = rotl64<1>(c2), r3 = rotl64<1>(c3), r4 = rotl64<1>(c4);
#define VG_TH0(x) bitop3_64<0x96>(x, c4, r1)
#define VG_TH1(x) bitop3_64<0x96>(x, c0, r2)
#define VG_TH2(x) bitop3_64<0x96>(x, c1, r3)
#define VG_TH3(x) bitop3_64<0x96>(x, c2, r4)
#define VG_TH4(x) bitop3_64<0x96>(x, c3, r0)
        // theta + rho + pi into b
        u64 b0 = VG_TH0(a[0]);
        u64 b10 = rotl64<1>(VG_TH1(a[1])), b20 = rotl64<62>(VG_TH2(a[2])), b5 = rotl64<28>(VG_TH3(a[3])), b15 = rotl64<27>(VG_TH4(a[4]));
        u64 b16 = rotl64<36>(VG_TH0(a[5])), b1 = rotl64<44>(VG_TH1(a[6])), b11 = rotl64<6>(VG_TH2(a[7])), b21 = rotl64<55>(VG_TH3(a[8])), b6 = rotl64<20>(VG_TH4(a[9]));
        u64 b7 = rotl64<3>(VG_TH0(a[10])), b17 = rotl64<10>(VG_TH1(a[11])), b2 = rotl64<43>(VG_TH2(a[12])), b12 = rotl64<25>(VG_TH3(a[13])), b22 = rotl64<39>(VG_TH4(a[14]));
        u64 b23 = rotl64<41>(VG_TH0(a[15])), b8 = rotl64<45>(VG_TH1(a[16])), b18 = rotl64<15>(VG_TH2(a[17])), b3 = rotl64<21>(VG_TH3(a[18])), b13 = rotl64<8>(VG_TH4(a[19]));
        u64 b14 = rotl64<18>(VG_TH0(a[20])), b24 = rotl64<2>(VG_TH1(a[21])), b9 = rotl64<61>(VG_TH2(a[22])), b19 = rotl64<56>(VG_TH3(a[23])), b4 = rotl64<14>(VG_TH4(a[24]));
#undef VG_TH0
#undef VG_TH1
#undef VG_TH2
#undef VG_TH3
#undef VG_TH4
        // chi: x ^ (~y & z) is the 3-input truth table 0xD2
#define VG_CHI(x, y, z) bitop3_64<0xD2>(x, y, z)
        a[0] = VG_CHI(b0, b1, b2) ^ rc; a[1] = VG_CHI(b1, b2, b3); a[2] = VG_CHI(b2, b3, b4); a[3] = VG_CHI(b3, b4, b0); a[4] = VG_CHI(b4, b0, b1);
        a[5] = VG_CHI(b5, b6, b7); a[6] = VG_CHI(b6, b7, b8); a[7] = VG_CHI(b7, b8, b9); a[8] = VG_CHI(b8, b9, b5); a[9] = VG_CHI(b9, b5, b6);
        a[10] = VG_CHI(b10, b11, b12); a[11] = VG_CHI(b11, b12, b13); a[12] = VG_CHI(b12, b13, b14); a[13] = VG_CHI(b13, b14, b10); a[14] = VG_CHI(b14, b10, b11);
        a[15] = VG_CHI(b15, b16, b17); a[16] = VG_CHI(b16, b17, b18); a[17] = VG_CHI(b17, b18, b19); a[18] = VG_CHI(b18, b19, b15); a[19] = VG_CHI(b19, b15, b16);
        a[20] = VG_CHI(b20, b21, b22); a[21] = VG_CHI(b21, b22, b23); a[22] = VG_CHI(b22, b23, b24); a[23] = VG_CHI(b23, b24, b20); a[24] = VG_CHI(b24, b20, b21);
#undef VG_CHI
}

VG_HD void keccak_f1600(u64 a[25]) {
    constexpr u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // The first and the last round stand outside the loop so that the compiler specialises them: in round 0 most
    // lanes of the padded 64-byte message are zero or constants, and of round 23 only lanes 1..3 are read.
    keccak_round(a, RC[0]);
#pragma unroll 1
    for (int round = 1; round < 23; round++) keccak_round(a, RC[round]);
    keccak_round(a, RC[23]);
}

// Keccak-256(X || Y) for a 64-byte public key, returning the low 20 bytes of the digest (the
// Ethereum address) as five words in memory order (out[0] holds digest bytes 12..15, little-endian).
VG_HD void keccak256_pub64_addr(const u32 xw[8], const u32 yw[8], u32 out[5]) {
    u64 a[25];
    // lane i = message bytes 8i..8i+7 little-endian; message = X big-endian bytes then Y
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // bytes 8i..8i+7 of X (big-endian) are words xw[7-2i] then xw[6-2i], each byte-swapped
        a[i] = (u64)bswap32(xw[7 - 2 * i]) | ((u64)bswap32(xw[6 - 2 * i]) << 32);
        a[4 + i] = (u64)bswap32(yw[7 - 2 * i]) | ((u64)bswap32(yw[6 - 2 * i]) << 32);
    }
    a[8] = 0x01;   // Keccak (pre-SHA-3) domain padding
#pragma unroll
    for (int i = 9; i < 25; i++) a[i] = 0;
    a[16] = 0x8000000000000000ULL;   // last byte of the 136-byte rate block
    keccak_f1600(a);
    // digest bytes 12..31 = lane1 high half, lane2, lane3
    out[0] = (u32)(a[1] >> 32);
    out[1] = (u32)a[2];
    out[2] = (u32)(a[2] >> 32);
    out[3] = (u32)a[3];
    out[4] = (u32)(a[3] >> 32);
}

}  // namespace vg
