// ec.h — secp256k1 group arithmetic over fe (9x29), single source for host and device.
//
// Host use: the per-dispatch base point k0*G (what the reference computes with libsecp256k1 in
// key_to_affine, src/gpu.rs:901-910), the shared offset tables (the reference builds its table on
// the device in init.wgsl:3-10 with a 256-step double-and-add per entry) and match confirmation.
// Device use: the arbitrary-scalar (KEYS) kernel's Jacobian accumulation.
//
// Curve: y^2 = x^3 + 7 over F_p.  Jacobian (X, Y, Z) <-> (X/Z^2, Y/Z^3); infinity is an explicit flag.
// All fe results are magnitude 1 unless stated; inputs must be magnitude 1.
#pragma once
#include "fe.h"

namespace vg {

struct ge {          // affine point, canonical coordinates
    fe x, y;
};

struct gej {
    fe x, y, z;
    u32 inf;         // 1 = point at infinity
};

// r = a - b for magnitude-1 inputs; result weakly normalised (magnitude 1).
VG_HD void fe_sub_n(fe &r, const fe &a, const fe &b) {
    fe nb;
    fe_neg(nb, b, 1);
    fe_add(r, a, nb);       // magnitude 3
    fe_normalize_weak(r);
}

VG_HD bool fe_is_zero_any(const fe &a) {
    fe t = a;
    fe_normalize(t);
    return fe_is_zero_canonical(t);
}

VG_HD void gej_set_infinity(gej &r) {
    fe_set_zero(r.x);
    fe_set_zero(r.y);
    fe_set_zero(r.z);
    r.inf = 1;
}

VG_HD void gej_from_ge(gej &r, const ge &a) {
    r.x = a.x;
    r.y = a.y;
    fe_set_one(r.z);
    r.inf = 0;
}

// r = 2a   (dbl-2009-l, a = 0)
VG_HD void gej_double(gej &r, const gej &a) {
    if (a.inf) {
        gej_set_infinity(r);
        return;
    }
    fe A, B, C, D, E, F, t;
    fe_sqr(A, a.x);
    fe_sqr(B, a.y);
    fe_sqr(C, B);
    fe_add(t, a.x, B);            // m2
    fe_normalize_weak(t);
    fe_sqr(t, t);
    fe_sub_n(t, t, A);
    fe_sub_n(t, t, C);
    fe_add(D, t, t);              // D = 2((X+B)^2 - A - C), m2
    fe_normalize_weak(D);
    fe_add(E, A, A);
    fe_add(E, E, A);              // 3A, m3
    fe_normalize_weak(E);
    fe_sqr(F, E);
    fe x3, y3, z3;
    fe_sub_n(x3, F, D);
    fe_sub_n(x3, x3, D);
    fe_sub_n(t, D, x3);
    fe_mul(y3, E, t);
    fe c8;
    fe_add(c8, C, C);
    fe_add(c8, c8, c8);           // 4C, m4
    fe_normalize_weak(c8);
    fe_add(c8, c8, c8);           // 8C, m2
    fe_normalize_weak(c8);
    fe_sub_n(y3, y3, c8);
    fe_mul(z3, a.y, a.z);
    fe_add(z3, z3, z3);
    fe_normalize_weak(z3);
    r.x = x3;
    r.y = y3;
    r.z = z3;
    r.inf = 0;                    // y = 0 has no solution on this curve (order is odd), so 2a != inf
}

// r = a + b with b affine (madd-2007-bl shape), all special cases handled.
VG_HD void gej_add_ge(gej &r, const gej &a, const ge &b) {
    if (a.inf) {
        gej_from_ge(r, b);
        return;
    }
    fe z1z1, u2, s2, h, rr, hh, hhh, v, t;
    fe_sqr(z1z1, a.z);
    fe_mul(u2, b.x, z1z1);
    fe_mul(t, a.z, z1z1);
    fe_mul(s2, b.y, t);
    fe_sub_n(h, u2, a.x);
    fe_sub_n(rr, s2, a.y);
    if (fe_is_zero_any(h)) {
        if (fe_is_zero_any(rr)) {
            gej_double(r, a);
        } else {
            gej_set_infinity(r);
        }
        return;
    }
    fe_sqr(hh, h);
    fe_mul(hhh, hh, h);
    fe_mul(v, a.x, hh);
    fe x3, y3, z3;
    fe_sqr(x3, rr);
    fe_sub_n(x3, x3, hhh);
    fe_sub_n(x3, x3, v);
    fe_sub_n(x3, x3, v);
    fe_sub_n(t, v, x3);
    fe_mul(y3, rr, t);
    fe_mul(t, a.y, hhh);
    fe_sub_n(y3, y3, t);
    fe_mul(z3, a.z, h);
    r.x = x3;
    r.y = y3;
    r.z = z3;
    r.inf = 0;
}

// Affine result of a single Jacobian point (one field inversion).  Returns false for infinity.
VG_HD bool ge_from_gej(ge &r, const gej &a) {
    if (a.inf) return false;
    fe zi, zi2, zi3;
    fe_inv(zi, a.z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(r.x, a.x, zi2);
    fe_mul(r.y, a.y, zi3);
    fe_normalize(r.x);
    fe_normalize(r.y);
    return true;
}

VG_HD void ge_neg(ge &r, const ge &a) {
    r.x = a.x;
    fe_neg(r.y, a.y, 1);
    fe_normalize(r.y);
}

// Generator (cross-checked against src/shaders/field.wgsl:346-347 by the CPU tests)
VG_HD void ge_generator(ge &g) {
    const u32 gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    const u32 gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    fe_from_words(g.x, gx);
    fe_from_words(g.y, gy);
}

}  // namespace vg

namespace vg {

// r = a + b for a Jacobian point a (not infinity, coordinates magnitude 1) and an affine point b
// (canonical), assuming a != +/-b.  3 squarings + 8 multiplications, no branches: the device
// fixed-window accumulation never meets the exceptional cases (see keys_scan_kernel).
VG_HD void gej_add_ge_nz(gej &r, const gej &a, const ge &b) {
    fe z1z1, u2, s2, h, rr, hh, j, v, t, x3, y3, z3;
    fe_sqr(z1z1, a.z);
    fe_mul(u2, b.x, z1z1);
    fe_mul(t, a.z, z1z1);
    fe_mul(s2, b.y, t);
    fe_neg(t, a.x, 1);
    fe_add(h, u2, t);             // magnitude 3
    fe_normalize_weak(h);
    fe_neg(t, a.y, 1);
    fe_add(rr, s2, t);
    fe_normalize_weak(rr);
    fe_sqr(hh, h);
    fe_mul(j, h, hh);
    fe_mul(v, a.x, hh);
    fe_sqr(x3, rr);
    // x3 = rr^2 - j - 2v
    fe_neg(t, j, 1);              // m2
    fe_add(x3, x3, t);            // m3
    fe_neg(t, v, 1);              // m2
    fe_add(x3, x3, t);            // m5
    fe_add(x3, x3, t);            // m7
    fe_normalize_weak(x3);
    // y3 = rr*(v - x3) - y1*j
    fe_neg(t, x3, 1);
    fe_add(t, t, v);              // m3
    fe_mul(y3, rr, t);
    fe_mul(t, a.y, j);
    fe_neg(t, t, 1);
    fe_add(y3, y3, t);            // m3
    fe_normalize_weak(y3);
    fe_mul(z3, a.z, h);
    r.x = x3;
    r.y = y3;
    r.z = z3;
    r.inf = 0;
}

}  // namespace vg

namespace vg {

// acc = k * G by fixed windows of WB bits, least significant window first.  Table: [256/WB windows]
// [2^WB - 1 entries][ES words]: limbs 0..8 = x, 9..17 = y of d * 2^(WB*w) * G (entry stride ES >= 18 words; a
// stride that is a multiple of 4 lets the device fetch an entry with 16-byte loads).  k: eight little-endian
// words, 0 < k < n.  Unsigned digits accumulated low to high keep the running sum below the next addend's
// scalar, so the branch-free mixed addition never meets P = +/-Q; "still at infinity" is a select.
// Works on any memory the table lives in (global or LDS on the device, heap on the host).
struct alignas(16) ec_u4 {
    u32 v[4];
};

// One window's step: acc += t (the table point of a non-zero digit d); a zero digit leaves acc alone, the first
// non-zero digit replaces the point at infinity — all by selects, no branches.
VG_HD void ec_fixed_accumulate(gej &acc, const ge &t, u32 d) {
    gej sum;
    gej_add_ge_nz(sum, acc, t);      // garbage while acc is at infinity; replaced below
    const bool take_table = acc.inf != 0;
    const bool skip = d == 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 nx = take_table ? t.x.n[i] : sum.x.n[i];
        const u32 ny = take_table ? t.y.n[i] : sum.y.n[i];
        const u32 nzl = take_table ? (i == 0 ? 1u : 0u) : sum.z.n[i];
        acc.x.n[i] = skip ? acc.x.n[i] : nx;
        acc.y.n[i] = skip ? acc.y.n[i] : ny;
        acc.z.n[i] = skip ? acc.z.n[i] : nzl;
    }
    acc.inf = skip ? acc.inf : 0u;
}

template <int WB, int ES>
VG_HD void ec_mul_gen_fixed(gej &acc, const u32 k[8], const u32 *tab) {
    static_assert(32 % WB == 0 && ES >= 18, "window geometry");
    constexpr u32 NE = (1u << WB) - 1u;
    gej_set_infinity(acc);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int w = 0; w < 256 / WB; w++) {
        const u32 d = (k[(w * WB) >> 5] >> ((w * WB) & 31)) & NE;
        const u32 e = (d ? d : 1u) - 1u;
        const u32 *ent = tab + ((u32)w * NE + e) * (u32)ES;
        u32 raw[20];
        if (ES % 4 == 0) {
            const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(ent);
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const ec_u4 t4 = e4[q];
#pragma unroll
                for (int i = 0; i < 4; i++) raw[4 * q + i] = t4.v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 18; i++) raw[i] = ent[i];
        }
        ge t;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            t.x.n[i] = raw[i];
            t.y.n[i] = raw[9 + i];
        }
        ec_fixed_accumulate(acc, t, d);
    }
}

// 4-bit windows, packed entries: host_gen_table_limbs ([64][15][18]).
VG_HD void ec_mul_gen_windows(gej &acc, const u32 k[8], const u32 *tab) { ec_mul_gen_fixed<4, 18>(acc, k, tab); }

// 8-bit windows, 80-byte entries: host_gen_table8_limbs ([32][255][20], 652 800 B, global memory on the device:
// L2-resident, one entry = five 16-byte loads).  Half the additions of the 4-bit form.
constexpr u32 EC_TABLE8_WORDS = 32u * 255u * 20u;
VG_HD void ec_mul_gen_w8(gej &acc, const u32 k[8], const u32 *tab8) { ec_mul_gen_fixed<8, 20>(acc, k, tab8); }

// Wide windows: WB = 16 .. 24 bits, NW = ceil(256 / WB) windows x (2^WB - 1) entries x 64 bytes (x then y as eight
// little-endian 32-bit words each: one 64-byte sector per entry, four 16-byte loads), built on the device
// (kernels.hip: gen_table_wide_kernel) and gathered through the L2 / Infinity Cache / HBM.  NW - 1 additions instead
// of the 31 of the 8-bit form (16 bits: 15 additions, 67 MB; 20 bits: 12 additions, 872 MB; 22 bits: 11, 3.2 GB; 24 bits: 10, 11.8 GB; 26 bits: 9, 43 GB);
// the limb conversion of a table point (two fe_from_words) is 4 % of an addition.
VG_HD constexpr u32 ec_wide_windows(u32 wb) { return (256u + wb - 1u) / wb; }
VG_HD constexpr u64 ec_wide_entries(u32 wb) { return (u64)ec_wide_windows(wb) * ((1ull << wb) - 1ull); }
VG_HD constexpr u64 ec_wide_words(u32 wb) { return ec_wide_entries(wb) * 16ull; }

// digit `w` (WB bits, least significant first) of the 256-bit little-endian scalar k
VG_HD u32 ec_wide_digit(const u32 k[8], u32 w, u32 wb) {
    const u32 bit = w * wb, i = bit >> 5, sh = bit & 31u;
    const u64 lo = k[i], hi = i + 1 < 8 ? k[i + 1] : 0u;
    return (u32)(((lo | (hi << 32)) >> sh) & ((1ull << wb) - 1ull));
}

template <int WB>
VG_HD void ec_mul_gen_wide(gej &acc, const u32 k[8], const u32 *tab) {
    static_assert(WB >= 9 && WB <= 26, "window width");
    constexpr u32 NW = ec_wide_windows(WB);
    constexpr u64 NE = (1ull << WB) - 1ull;
    gej_set_infinity(acc);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (u32 w = 0; w < NW; w++) {
        const u32 d = ec_wide_digit(k, w, WB);
        const u32 e = (d ? d : 1u) - 1u;
        const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(tab + ((u64)w * NE + e) * 16ull);
        u32 raw[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ec_u4 t4 = e4[q];
#pragma unroll
            for (int i = 0; i < 4; i++) raw[4 * q + i] = t4.v[i];
        }
        ge t;
        fe_from_words(t.x, raw);
        fe_from_words(t.y, raw + 8);
        ec_fixed_accumulate(acc, t, d);
    }
}

// ---- signed windows: every entry serves a digit AND its negation -----------------------------------------------------------
// With 288 GB of HBM per device the table can be wider still.  Windows of ST bits (ST odd: 25 | 27 | 29); the raw ST-bit
// digit v (plus the carry of the window below) is recoded to d = v or d = v - 2^ST (carrying 1 up) so that |d| <= 2^(ST-1);
// the table holds m * 2^(ST w) * G for the MAGNITUDES m = 1 .. 2^(ST-1) and a negative digit takes (x, p - y).  One more
// bit (the last carry) makes it NW = ceil(257 / ST) windows:
//     29 bits:  9 windows =  8 additions, 138 GB        27 bits: 10 windows = 9 additions (the unsigned 26-bit table's count)
//     25 bits: 11 windows = 10 additions,  5.9 GB       in 21.5 GB instead of 43
// The partial sum of the windows below w is smaller in magnitude than any non-zero addend of window w, so the branch-free
// mixed addition still never meets P = +/-Q.  The top window is short (its raw digit has 256 - ST (NW - 1) bits) and its
// part of the table is sized for that; its largest magnitude, 2^(256 - ST (NW - 1)), stands for the scalar 2^256 itself,
// which the table builder takes mod n (kernels.hip: gen_small_signed_kernel).
VG_HD constexpr bool ec_table_signed(u32 bits) { return (bits & 1u) != 0u; }
VG_HD constexpr u32 ec_signed_windows(u32 st) { return (257u + st - 1u) / st; }
VG_HD constexpr u64 ec_signed_per(u32 st) { return 1ull << (st - 1u); }                                  // magnitudes 1 .. 2^(st-1) per window
VG_HD constexpr u32 ec_signed_top_bits(u32 st) { return 256u - st * (ec_signed_windows(st) - 1u); }      // raw bits of the top window
VG_HD constexpr u64 ec_signed_top_per(u32 st) { return 1ull << ec_signed_top_bits(st); }                 // its magnitudes: 1 .. 2^top_bits
VG_HD constexpr u64 ec_signed_entries(u32 st) { return (u64)(ec_signed_windows(st) - 1u) * ec_signed_per(st) + ec_signed_top_per(st); }
VG_HD constexpr u32 ec_signed_half(u32 st) { return (st - 1u) / 2u; }                                    // the builder's half-width
VG_HD constexpr u64 ec_signed_small_entries(u32 st) { return (u64)ec_signed_windows(st) * 2ull * (1ull << ec_signed_half(st)); }
// table sizes for either kind of width (words)
VG_HD constexpr u64 ec_table_words(u32 bits) { return ec_table_signed(bits) ? ec_signed_entries(bits) * 16ull : ec_wide_words(bits); }
VG_HD constexpr u64 ec_table_small_words(u32 bits) {
    return ec_table_signed(bits) ? ec_signed_small_entries(bits) * 16ull : ec_wide_words(bits / 2u);
}

template <int ST>
VG_HD void ec_mul_gen_signed(gej &acc, const u32 k[8], const u32 *tab) {
    static_assert(ST == 25 || ST == 27 || ST == 29, "signed window stride");
    constexpr u32 NW = ec_signed_windows(ST);
    constexpr u64 PER = ec_signed_per(ST);
    gej_set_infinity(acc);
    u32 carry = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (u32 w = 0; w < NW; w++) {
        const u32 v = ec_wide_digit(k, w, ST) + carry;           // <= 2^ST
        const bool neg = v > (1u << (ST - 1));
        const u32 m = neg ? (1u << ST) - v : v;                  // magnitude 0 .. 2^(ST-1)
        carry = neg ? 1u : 0u;
        const u32 e = (m ? m : 1u) - 1u;
        const ec_u4 *e4 = reinterpret_cast<const ec_u4 *>(tab + ((u64)w * PER + e) * 16ull);
        u32 raw[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ec_u4 t4 = e4[q];
#pragma unroll
            for (int i = 0; i < 4; i++) raw[4 * q + i] = t4.v[i];
        }
        ge t;
        fe ny;
        fe_from_words(t.x, raw);
        fe_from_words(t.y, raw + 8);
        fe_neg(ny, t.y, 1);
        fe_normalize_weak(ny);                                   // p - y, magnitude 1
#pragma unroll
        for (int i = 0; i < 9; i++) t.y.n[i] = neg ? ny.n[i] : t.y.n[i];
        ec_fixed_accumulate(acc, t, m);
    }
}

constexpr u32 EC_TABLE16_ENTRIES = 16u * 65535u;
constexpr u32 EC_TABLE16_WORDS = EC_TABLE16_ENTRIES * 16u;
VG_HD void ec_mul_gen_w16(gej &acc, const u32 k[8], const u32 *tab16) { ec_mul_gen_wide<16>(acc, k, tab16); }

// The generator tables a kernel may use: the 8-bit one always, a wide one (of `wide_bits` bits) when it has been built.
struct GenTables {
    const u32 *w8;
    const u32 *wide;
    u32 wide_bits;
};
VG_HD void ec_mul_gen_tables(gej &acc, const u32 k[8], const GenTables &g) {
    if (g.wide && g.wide_bits == 16) ec_mul_gen_wide<16>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 20) ec_mul_gen_wide<20>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 22) ec_mul_gen_wide<22>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 24) ec_mul_gen_wide<24>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 26) ec_mul_gen_wide<26>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 25) ec_mul_gen_signed<25>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 27) ec_mul_gen_signed<27>(acc, k, g.wide);
    else if (g.wide && g.wide_bits == 29) ec_mul_gen_signed<29>(acc, k, g.wide);
    else ec_mul_gen_w8(acc, k, g.w8);
}

}  // namespace vg
