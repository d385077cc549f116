// fe.h — arithmetic in F_p, p = 2^256 - 2^32 - 977 (secp256k1 base field), 9 limbs x 29 bits.
//
// Single source for the HIP kernels (hipcc, gfx950) and for the host side of libvgen_hip.so
// (base-point / table construction, match confirmation) and its CPU unit tests (g++).
//
// Why 9x29 and not the reference's 8x32 (src/shaders/field.wgsl:18-210): measured on MI355X
// (tools/ubench_valu.hip, profiles/r01_ubench_valu.jsonl) every carry-flag instruction
// (v_add_co/v_addc) issues at half rate AND needs a wait state before its consumer, while
// v_mad_u64_u32 (32x32+64 -> 64) and v_lshl_add_u64 run at the same half rate with no flags.
// A reduced radix with 3 spare bits per limb lets all 81 partial products of a multiplication
// be accumulated in 64-bit column sums by v_mad_u64_u32 with no carry handling at all (see "column form"
// below), and makes field add/sub/negate nine independent full-rate 32-bit adds.
//
// Representation: value = sum n[i] * 2^(29 i).  "magnitude m": a sum of m weakly normalised values, i.e.
// n[0] <= m*(2^29 + 2^23), n[1] <= m*(2^29 + 2^16), n[2..7] <= m*(2^29-1), n[8] <= m*2^24 (a product leaves its
// top fold un-rippled in limbs 0 and 1); m <= 7 always (7*(2^29 + 2^23) < 2^32).  Every function states the
// magnitudes it accepts and returns; products require m_a * m_b <= 6 so that a column sum of nine products
// of size m_a*m_b*2^58 (1.6 % more where limb 0 takes part) plus the fold terms stays below 2^64.
// Canonical = strictly normalised limbs (< 2^29, top < 2^24) and value < p (fe_normalize / fe_canonicalize).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VG_HD __host__ __device__ __forceinline__
#else
#define VG_HD inline
#endif

namespace vg {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe {
    u32 n[9];
};

constexpr u32 FE_M29 = 0x1FFFFFFFu;
constexpr u32 FE_M24 = 0x00FFFFFFu;
// p in 29-bit limbs
constexpr u32 FE_P0 = 0x1FFFFC2Fu, FE_P1 = 0x1FFFFFF7u, FE_PM = 0x1FFFFFFFu, FE_P8 = 0x00FFFFFFu;
// 2^261 mod p = 2^37 + 0x7A20  ->  R1 * 2^29 + R0
constexpr u32 FE_R0 = 0x7A20u, FE_R1 = 0x100u;

VG_HD void fe_set_zero(fe &r) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.n[i] = 0;
}

VG_HD void fe_set_one(fe &r) {
    fe_set_zero(r);
    r.n[0] = 1;
}

// r = a + b.  magnitude m_a + m_b.
VG_HD void fe_add(fe &r, const fe &a, const fe &b) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.n[i] = a.n[i] + b.n[i];
}

// r = -a for a of magnitude <= m.  Result magnitude m + 1.   ((m+1)*p - a, limb-wise, no borrows)
VG_HD void fe_neg(fe &r, const fe &a, u32 m) {
    const u32 k = m + 1;
    r.n[0] = k * FE_P0 - a.n[0];
    r.n[1] = k * FE_P1 - a.n[1];
#pragma unroll
    for (int i = 2; i < 8; i++) r.n[i] = k * FE_PM - a.n[i];
    r.n[8] = k * FE_P8 - a.n[8];
}

// ---- multiplication: column form -----------------------------------------------------------------------------
// The 17 column sums S_k = sum_{i+j=k} a_i b_j are accumulated INDEPENDENTLY in 64 bits (no carry chain:
// 9 * m_a*m_b * 2^58 < 2^63.76 for m_a*m_b <= 6), and the high columns are folded into the low ones as whole
// 64-bit quantities, split only into their two register halves (which costs nothing):
//     S_k * 2^(29k),  k >= 9,  2^261 = R1*2^29 + R0,  S_k = lo + 2^32 hi,  2^32 = 8 * 2^29
//       ->  D_{k-9} += lo*R0;   D_{k-8} += lo*R1 + hi*(8 R0);   D_{k-7} += hi*(8 R1)
// (k descending, so that column 16's spill into D_9 is folded when k reaches 9).  Column 8's high half has
// weight 2^264 = 2^8 (2^32 + 977):  D_0 += hi8 * (977*256),  D_1 += hi8 * 2^11.  Only then one carry pass and
// the short top fold.  Against a carry-chained column walk (the first version of this file: 219 instructions
// per multiplication) this trades 17 64-bit shifts, 17 masks and a shift-heavy fold for 34 more
// v_mad_u64_u32: 173 instructions, ~670 instead of ~790 issue cycles on gfx950 (tools/isa_census.py).
// Bounds: every D stays below 2^63.76 + 2^51 < 2^64; after folding hi8, D_8 < 2^32.

// A multiplier the compiler cannot see through: on gfx950 a 32x32+64 multiply-add (v_mad_u64_u32) is cheaper
// than the shift + zero-extension + 64-bit add that hipcc substitutes for a multiplication by a power of two.
template <u32 V>
VG_HD u32 fe_opaque_() {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 r;
    asm("s_mov_b32 %0, %1" : "=s"(r) : "i"(V));
    return r;
#else
    return V;
#endif
}

// One routine for a*b (+c) and a^2 (+c).  The high columns are produced first (k = 16 .. 9) and folded at
// once, so only ten accumulators D_0..D_9 plus the column in hand are alive (22 registers instead of 34).
template <bool SQR>
VG_HD void fe_mul_columns_(fe &r, const fe &a, const fe &b, const fe *c3) {
    const u32 R1 = fe_opaque_<FE_R1>(), R0x8 = FE_R0 * 8u, R1x8 = fe_opaque_<FE_R1 * 8u>();
    u32 d[9];   // SQR: doubled limbs for the cross terms (a of magnitude 1)
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = SQR ? (a.n[i] << 1) : 0u;
    u64 D[10];
#pragma unroll
    for (int k = 0; k < 10; k++) D[k] = 0;
#pragma unroll
    for (int k = 16; k >= 0; k--) {
        u64 s = k <= 9 ? D[k] : 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j >= 9) continue;
            if (!SQR) s += (u64)a.n[i] * b.n[j];
            else if (i < j) s += (u64)d[i] * a.n[j];
            else if (i == j) s += (u64)a.n[i] * a.n[i];
        }
        if (k >= 9) {
            const u32 lo = (u32)s, hi = (u32)(s >> 32);
            D[k - 9] += (u64)lo * FE_R0;
            D[k - 8] += (u64)lo * R1;
            D[k - 8] += (u64)hi * R0x8;
            D[k - 7] += (u64)hi * R1x8;   // k = 16 reaches D_9, which is column 9's starting value
        } else {
            D[k] = s;
        }
    }
    if (c3) {   // 2^31 per limb against 2^63.76: no effect on the bounds; a mad, not mov + 64-bit add
        const u32 one = fe_opaque_<1>();
#pragma unroll
        for (int k = 0; k < 9; k++) D[k] += (u64)c3->n[k] * one;
    }
    {
        const u32 hi8 = (u32)(D[8] >> 32);
        D[8] = (u32)D[8];
        D[0] += (u64)hi8 * (977u * 256u);
        D[1] += (u64)hi8 * R1x8;
    }
    u32 f[8];
    u64 c = D[0];
    f[0] = (u32)c & FE_M29; c >>= 29;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        c += D[k];
        f[k] = (u32)c & FE_M29; c >>= 29;
    }
    c += D[8];                                   // < 2^36 + 2^32
    const u32 n8 = (u32)c & FE_M24;
    const u32 ov = (u32)(c >> 24);               // < 2^13, weight 2^256 == 2^32 + 977 = 8*2^29 + 977
    // The top fold lands in limbs 0 and 1 and is NOT rippled further: a product is "weakly normalised" with
    // n[0] < 2^29 + 2^23, n[1] < 2^29 + 2^16, n[2..7] < 2^29, n[8] < 2^24 (value < 2^256 + 2^53).  Every consumer
    // (products: 1.6 % on one limb against 19 % of slack; fe_neg; the carry chains of fe_canonicalize /
    // fe_parity_weak / fe_normalize_weak) takes that; the ripple would be 22 instructions per multiplication.
    r.n[0] = f[0] + ov * 977u;
    r.n[1] = f[1] + ov * 8u;
#pragma unroll
    for (int k = 2; k < 8; k++) r.n[k] = f[k];
    r.n[8] = n8;
}

// r = a * b.  Requires m_a * m_b <= 6.  Result magnitude 1.
VG_HD void fe_mul(fe &r, const fe &a, const fe &b) {
    fe_mul_columns_<false>(r, a, b, nullptr);
}

// r = a^2.  Requires m_a <= 1 (the doubled cross terms use 2*a_i).  Result magnitude 1.
VG_HD void fe_sqr(fe &r, const fe &a) {
    fe_mul_columns_<true>(r, a, a, nullptr);
}

// r = a * b + c, c of magnitude <= 3: c's limbs join the low product digits before the fold, so the
// sum costs nine 32-bit adds and no extra carry pass.  Result magnitude 1 (as fe_mul).
VG_HD void fe_mul_add(fe &r, const fe &a, const fe &b, const fe &c3) {
    fe_mul_columns_<false>(r, a, b, &c3);
}

// r = a^2 + c, c of magnitude <= 3 (see fe_mul_add); a of magnitude 1.
VG_HD void fe_sqr_add(fe &r, const fe &a, const fe &c3) {
    fe_mul_columns_<true>(r, a, a, &c3);
}

// Canonical representative of a WEAKLY NORMALISED value (magnitude 1: output of fe_mul / fe_sqr / fe_*_add /
// fe_normalize_weak; limbs 0 and 1 may carry an un-rippled top fold, n[8] <= 2^24 + 1).  One carry pass computing both v = value with the
// bits above 2^256 folded back and u = v + C (C = 2^32 + 977 = 2^256 - p); u reaching 2^256 means
// v >= p and the answer is u - 2^256, otherwise it is v.
VG_HD void fe_canonicalize(fe &r) {
    const u32 ov = r.n[8] >> 24;                      // 0 or 1 (times 2^256)
    u32 v[9], u[9];
    u32 cv = r.n[0] + ov * 977u;                      // v = r - ov*2^256 + ov*C
    u32 cu = cv + 977u;
    v[0] = cv & FE_M29; cv >>= 29;
    u[0] = cu & FE_M29; cu >>= 29;
    cv += r.n[1] + ov * 8u;
    cu += r.n[1] + ov * 8u + 8u;
    v[1] = cv & FE_M29; cv >>= 29;
    u[1] = cu & FE_M29; cu >>= 29;
#pragma unroll
    for (int k = 2; k < 8; k++) {
        cv += r.n[k];
        cu += r.n[k];
        v[k] = cv & FE_M29; cv >>= 29;
        u[k] = cu & FE_M29; cu >>= 29;
    }
    const u32 top = r.n[8] & FE_M24;
    cv += top;                                        // < 2^24 + 1: if the fold made v reach 2^256 ...
    cu += top;                                        // ... then u did as well and u - 2^256 is the answer
    v[8] = cv & FE_M24;
    u[8] = cu & FE_M24;
    const bool ge = (cu >> 24) != 0;
#pragma unroll
    for (int k = 0; k < 9; k++) r.n[k] = ge ? u[k] : v[k];
}

// "some lane of the wave" on the device (a wave-uniform branch the compiler cannot turn into selects), the
// plain condition on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define VG_ANY_LANE(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
#else
#define VG_ANY_LANE(cond) (cond)
#endif

// fe_canonicalize for products, which are almost always canonical already apart from the un-rippled carries of
// limbs 0 and 1: the value differs from its representative only if it reaches 2^256 (probability ~2^-23 after
// the fold) or lies in [p, 2^256) (2^-224).  One carry pass produces the strictly normalised limbs and, from the
// chain of v + C, whether that rare case is present; the full routine then runs behind a wave-uniform branch
// (it is the identity on lanes that did not need it).  ~40 instructions instead of 61.
VG_HD void fe_canonicalize_product(fe &r) {
    u32 v[9];
    u32 cv = r.n[0];
    v[0] = cv & FE_M29; cv >>= 29;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        cv += r.n[k];
        v[k] = cv & FE_M29; cv >>= 29;
    }
    cv += r.n[8];
    v[8] = cv;
    // v >= p needs limbs 2..7 all ones and the top limb 2^24 - 1 (p = 2^256 - 2^32 - 977); v >= 2^256 shows in
    // bit 24 of the top limb.  A superset test: the slow path decides exactly.
    const u32 ones = v[2] & v[3] & v[4] & v[5] & v[6] & v[7];
    const bool fix = cv >= FE_M24 && (cv > FE_M24 || ones == FE_M29);
    if (VG_ANY_LANE(fix)) {
        fe_canonicalize(r);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) r.n[k] = v[k];
    }
}

// Parity (bit 0) of the canonical representative of a WEAKLY NORMALISED value, without producing the
// representative: the carry chain of u = v + C alone decides v >= p, and subtracting the odd p flips the parity.
// 25 instructions instead of fe_canonicalize's 61 — all a compressed public key needs of y.
VG_HD u32 fe_parity_weak(const fe &r) {
    const u32 ov = r.n[8] >> 24;
    const u32 v0 = r.n[0] + ov * 977u;
    u32 cu = (v0 + 977u) >> 29;
    cu += r.n[1] + ov * 8u + 8u;
    cu >>= 29;
#pragma unroll
    for (int k = 2; k < 8; k++) {
        cu += r.n[k];
        cu >>= 29;
    }
    cu += r.n[8] & FE_M24;
    return (v0 ^ (cu >> 24)) & 1u;
}

// Weak normalisation: any magnitude <= 7 in, magnitude 1 out (value unchanged mod p, < 2^256 + 2^233).
VG_HD void fe_normalize_weak(fe &r) {
    u32 c = r.n[0];
    u32 t[9];
    t[0] = c & FE_M29; c >>= 29;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        c += r.n[k];
        t[k] = c & FE_M29; c >>= 29;
    }
    c += r.n[8];
    u32 ov = c >> 24;                 // < 2^8
    t[8] = c & FE_M24;
    c = t[0] + ov * 977u;
    r.n[0] = c & FE_M29; c >>= 29;
    c += t[1] + ov * 8u;
    r.n[1] = c & FE_M29; c >>= 29;
#pragma unroll
    for (int k = 2; k < 8; k++) {
        c += t[k];
        r.n[k] = c & FE_M29; c >>= 29;
    }
    r.n[8] = t[8] + c;
}

// Full normalisation to the canonical representative in [0, p).  Any magnitude <= 7 in.
VG_HD void fe_normalize(fe &r) {
    fe_normalize_weak(r);             // value < 2^256 + 2^233, limbs in range
    // one more top fold: n[8] may be exactly 2^24 (bit 24 set)
    u32 ov = r.n[8] >> 24;
    r.n[8] &= FE_M24;
    // now value < 2^256; add ov*C and decide whether value >= p.  Adding C = 2^32+977 and checking
    // bit 256 of (value + C) tells value >= p, because p + C = 2^256.
    u32 t[9];
    u32 c = r.n[0] + ov * 977u;
    t[0] = c & FE_M29; c >>= 29;
    c += r.n[1] + ov * 8u;
    t[1] = c & FE_M29; c >>= 29;
#pragma unroll
    for (int k = 2; k < 8; k++) {
        c += r.n[k];
        t[k] = c & FE_M29; c >>= 29;
    }
    t[8] = r.n[8] + c;                // < 2^24 + 1; cannot reach 2^25
    // t < 2^256 + small and t == value (mod p).  If the ov fold pushed bit 24 of t[8] up again it
    // is still < p + C, handled by the same test below.
    // u = t + C; if u >= 2^256 then t >= p and the canonical value is u - 2^256.
    u32 u[9];
    c = t[0] + 977u;
    u[0] = c & FE_M29; c >>= 29;
    c += t[1] + 8u;
    u[1] = c & FE_M29; c >>= 29;
#pragma unroll
    for (int k = 2; k < 8; k++) {
        c += t[k];
        u[k] = c & FE_M29; c >>= 29;
    }
    c += t[8];
    u[8] = c & FE_M24;
    const bool ge = (c >> 24) != 0;
#pragma unroll
    for (int k = 0; k < 9; k++) r.n[k] = ge ? u[k] : t[k];
}

VG_HD bool fe_is_zero_canonical(const fe &a) {
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) z |= a.n[i];
    return z == 0;
}

VG_HD bool fe_equal_canonical(const fe &a, const fe &b) {
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) z |= a.n[i] ^ b.n[i];
    return z == 0;
}

// Canonical fe -> eight 32-bit words, w[0] least significant (the reference's limb order,
// src/gpu.rs:891-899).
VG_HD void fe_to_words(const fe &a, u32 w[8]) {
    w[0] = a.n[0] | (a.n[1] << 29);
    w[1] = (a.n[1] >> 3) | (a.n[2] << 26);
    w[2] = (a.n[2] >> 6) | (a.n[3] << 23);
    w[3] = (a.n[3] >> 9) | (a.n[4] << 20);
    w[4] = (a.n[4] >> 12) | (a.n[5] << 17);
    w[5] = (a.n[5] >> 15) | (a.n[6] << 14);
    w[6] = (a.n[6] >> 18) | (a.n[7] << 11);
    w[7] = (a.n[7] >> 21) | (a.n[8] << 8);
}

VG_HD void fe_from_words(fe &r, const u32 w[8]) {
    r.n[0] = w[0] & FE_M29;
    r.n[1] = ((w[0] >> 29) | (w[1] << 3)) & FE_M29;
    r.n[2] = ((w[1] >> 26) | (w[2] << 6)) & FE_M29;
    r.n[3] = ((w[2] >> 23) | (w[3] << 9)) & FE_M29;
    r.n[4] = ((w[3] >> 20) | (w[4] << 12)) & FE_M29;
    r.n[5] = ((w[4] >> 17) | (w[5] << 15)) & FE_M29;
    r.n[6] = ((w[5] >> 14) | (w[6] << 18)) & FE_M29;
    r.n[7] = ((w[6] >> 11) | (w[7] << 21)) & FE_M29;
    r.n[8] = w[7] >> 8;
}

// r = a^-1 (a != 0, magnitude 1).  Fermat, a^(p-2), with the standard secp256k1 addition chain
// (255 squarings + 15 multiplications).  inv(0) = 0.  Kept as the independent cross-check of fe_inv (tests) — the
// product inverts with divsteps below: ~30 000 instructions here against ~16 000 there, and the root inversions of a
// dispatch are one lone wave per SIMD whose latency is its instruction count.
VG_HD void fe_sqr_n_(fe &r, int n) {
    for (int i = 0; i < n; i++) fe_sqr(r, r);
}

VG_HD void fe_inv_fermat(fe &r, const fe &a) {
    fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t1;
    fe_sqr(x2, a);        fe_mul(x2, x2, a);
    fe_sqr(x3, x2);       fe_mul(x3, x3, a);
    x6 = x3;   fe_sqr_n_(x6, 3);    fe_mul(x6, x6, x3);
    x9 = x6;   fe_sqr_n_(x9, 3);    fe_mul(x9, x9, x3);
    x11 = x9;  fe_sqr_n_(x11, 2);   fe_mul(x11, x11, x2);
    x22 = x11; fe_sqr_n_(x22, 11);  fe_mul(x22, x22, x11);
    x44 = x22; fe_sqr_n_(x44, 22);  fe_mul(x44, x44, x22);
    x88 = x44; fe_sqr_n_(x88, 44);  fe_mul(x88, x88, x44);
    x176 = x88; fe_sqr_n_(x176, 88); fe_mul(x176, x176, x88);
    x220 = x176; fe_sqr_n_(x220, 44); fe_mul(x220, x220, x44);
    x223 = x220; fe_sqr_n_(x223, 3);  fe_mul(x223, x223, x3);
    t1 = x223; fe_sqr_n_(t1, 23);   fe_mul(t1, t1, x22);
    fe_sqr_n_(t1, 5);    fe_mul(t1, t1, a);
    fe_sqr_n_(t1, 3);    fe_mul(t1, t1, x2);
    fe_sqr_n_(t1, 2);    fe_mul(r, t1, a);
}

// ---- inversion by divsteps (Bernstein-Yang "safegcd", the delta = 1/2 variant) --------------------------------------
// The reference inverts with a bit-serial Fermat ladder per key (src/shaders/field.wgsl:195-210).  Here: the divstep map
//     (delta, f, g) -> (1 - delta, g, (g - f)/2)   if delta > 0 and g odd
//                      (1 + delta, f, (g + f)/2)   if g odd
//                      (1 + delta, f, g/2)         otherwise
// started at (1/2, p, a) reaches g = 0, f = +-gcd = +-1 within 590 steps for 256-bit inputs (the published bound for this
// variant); tracking the same linear steps on (d, e) = (0, 1) modulo p keeps f == d a, g == e a (mod p), so a^-1 = f d.
// Steps run in batches of 29 (the limb width): a batch works on the low 32 bits of f and g only and yields a 2x2 integer
// matrix t with [f', g'] = t [f, g] / 2^29, which is then applied once to the 9-limb f, g (exact division) and to d, e
// (division made exact by adding the multiple of p that clears the low limb; p = 2^256 - 2^32 - 977 makes that multiple
// three multiply-adds).  21 batches = 609 >= 590 steps; a wave stops early once every lane's g is zero (further steps
// would change nothing but d's representative).  Branch-free per lane: masks, no divergence.  inv(0) = 0.
//
// Signed 9 x 29: limbs 0..7 in [0, 2^29), limb 8 carries the sign.
struct fe_sgn {
    int32_t n[9];
};

constexpr u32 FE_PINV29 = 0x0DDACACFu;   // p^-1 mod 2^29
constexpr int FE_DIVSTEP_BATCHES = 21;

// a * b + c on the signed 32 x 32 + 64 multiplier (v_mad_i64_i32).  Spelled out for the device: hipcc knows the limbs
// below are non-negative, turns their sign extension into a zero extension and then emulates the mixed-sign product
// with two unsigned multiply-adds and a correction per term (102 v_mad_u64_u32 + 56 v_mul_lo_u32 per batch).
VG_HD int64_t fe_smad_(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t d;
    u64 carry_out;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_out) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (int64_t)a * b + c;
#endif
}

// 29 divsteps on the low words; zeta = -(delta + 1/2).  t = (u, v, q, r), entries of magnitude <= 2^29.
VG_HD int32_t fe_divsteps29_(int32_t zeta, u32 f, u32 g, int32_t t[4]) {
    u32 u = 1, v = 0, q = 0, r = 1;
#pragma unroll
    for (int i = 0; i < 29; i++) {   // 21 instructions per step; unrolled: no loop counter on the critical path
        const u32 odd = 0u - (g & 1u);                                 // all ones when g is odd
        const u32 sw = (u32)((int32_t)((u32)zeta & odd) >> 31);        // ... and delta > 0: f and g change places
        const u32 one = sw & 1u;
        const u32 nf = (f ^ sw) + one, nu = (u ^ sw) + one, nv = (v ^ sw) + one;   // -f, -u, -v when swapping
        g += nf & odd;
        q += nu & odd;
        r += nv & odd;
        zeta = (int32_t)(((u32)zeta ^ sw) - 1u);                       // delta -> 1 - delta | delta + 1
        f += g & sw;                                                   // swapping: f + (g - f) = the old g
        u = (u + (q & sw)) << 1;
        v = (v + (r & sw)) << 1;
        g >>= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return zeta;
}

// [f, g] <- t [f, g] / 2^29 (exact)
VG_HD void fe_divsteps_apply_fg_(fe_sgn &f, fe_sgn &g, const int32_t t[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = fe_smad_(u, f.n[0], fe_smad_(v, g.n[0], 0));
    int64_t cg = fe_smad_(q, f.n[0], fe_smad_(r, g.n[0], 0));
    cf >>= 29;   // the low 29 bits are zero by construction
    cg >>= 29;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf = fe_smad_(u, f.n[i], fe_smad_(v, g.n[i], cf));
        cg = fe_smad_(q, f.n[i], fe_smad_(r, g.n[i], cg));
        f.n[i - 1] = (int32_t)((u32)cf & FE_M29);
        g.n[i - 1] = (int32_t)((u32)cg & FE_M29);
        cf >>= 29;
        cg >>= 29;
    }
    f.n[8] = (int32_t)cf;
    g.n[8] = (int32_t)cg;
}

// [d, e] <- t [d, e] / 2^29 mod p, with d, e kept in (-2p, p): a negative input is lifted by p first (as a multiple
// m of p riding along: u p + ... ), then m is lowered by the amount in [0, 2^29) that makes the low limb vanish.
VG_HD void fe_divsteps_apply_de_(fe_sgn &d, fe_sgn &e, const int32_t t[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.n[8] >> 31, se = e.n[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = fe_smad_(u, d.n[0], fe_smad_(v, e.n[0], 0));
    int64_t ce = fe_smad_(q, d.n[0], fe_smad_(r, e.n[0], 0));
    md -= (int32_t)((FE_PINV29 * (u32)cd + (u32)md) & FE_M29);
    me -= (int32_t)((FE_PINV29 * (u32)ce + (u32)me) & FE_M29);
    // + m p,  p = 2^256 - 2^32 - 977:  -977 m in limb 0, -8 m in limb 1 (2^32 = 8 * 2^29), +2^24 m in limb 8
    cd = fe_smad_(md, -977, cd);
    ce = fe_smad_(me, -977, ce);
    cd >>= 29;
    ce >>= 29;
    cd = fe_smad_(md, -8, cd);
    ce = fe_smad_(me, -8, ce);
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd = fe_smad_(u, d.n[i], fe_smad_(v, e.n[i], cd));
        ce = fe_smad_(q, d.n[i], fe_smad_(r, e.n[i], ce));
        if (i == 8) {
            cd = fe_smad_(md, 1 << 24, cd);
            ce = fe_smad_(me, 1 << 24, ce);
        }
        d.n[i - 1] = (int32_t)((u32)cd & FE_M29);
        e.n[i - 1] = (int32_t)((u32)ce & FE_M29);
        cd >>= 29;
        ce >>= 29;
    }
    d.n[8] = (int32_t)cd;
    e.n[8] = (int32_t)ce;
}

// The last step of the inversion: a^-1 = sign(f) d with d anywhere in (-2p, p) — signed limbs: 0..7 in [0, 2^29), limb 8
// from -2^25 (d near -2p) to 2^24.  Lifted by 3p LIMB-WISE, which keeps every limb non-negative for both signs:
//   f > 0:  3 P_8 + d_8 >= 3 (2^24 - 1) - 2^25 > 0, the low limbs only grow;
//   f < 0:  3 P_i - d_i >= 3 (2^29 - 977) - 2^29 > 0 for i < 8, 3 P_8 - d_8 >= 3 (2^24 - 1) - 2^24 > 0.
// (Round 3 lifted by 2p: for f > 0 and d within ~2^233 of -2p limb 8 came out negative and wrapped as u32 — a silent wrong
// inverse, however improbable; tests/test_core_field.py drives this function with exactly those d.)  All limbs stay below
// 4 * 2^29: magnitude 4 into fe_normalize.
VG_HD void fe_divsteps_lift_(fe &r, const fe_sgn &d, int32_t f_top) {
    const u32 neg = (u32)(f_top >> 31);
    r.n[0] = 3u * FE_P0 + (((u32)d.n[0] ^ neg) - neg);
    r.n[1] = 3u * FE_P1 + (((u32)d.n[1] ^ neg) - neg);
#pragma unroll
    for (int i = 2; i < 8; i++) r.n[i] = 3u * FE_PM + (((u32)d.n[i] ^ neg) - neg);
    r.n[8] = 3u * FE_P8 + (((u32)d.n[8] ^ neg) - neg);
    fe_normalize(r);
}

// r = a^-1 (any magnitude <= 7 in; canonical out; inv(0) = 0).
VG_HD void fe_inv(fe &r, const fe &a) {
    fe x = a;
    fe_normalize(x);
    fe_sgn f, g, d, e;
    f.n[0] = (int32_t)FE_P0; f.n[1] = (int32_t)FE_P1; f.n[8] = (int32_t)FE_P8;
#pragma unroll
    for (int i = 2; i < 8; i++) f.n[i] = (int32_t)FE_PM;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        g.n[i] = (int32_t)x.n[i];
        d.n[i] = 0;
        e.n[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int b = 0; b < FE_DIVSTEP_BATCHES; b++) {
        u32 nz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nz |= (u32)g.n[i];
        if (!VG_ANY_LANE(nz != 0)) break;
        int32_t t[4];
        // the low 32 bits of f and g: limb 0 and the low bits of limb 1 (limb 8 never takes part: 9 limbs)
        zeta = fe_divsteps29_(zeta, (u32)f.n[0] | ((u32)f.n[1] << 29), (u32)g.n[0] | ((u32)g.n[1] << 29), t);
        fe_divsteps_apply_fg_(f, g, t);
        fe_divsteps_apply_de_(d, e, t);
    }
    fe_divsteps_lift_(r, d, f.n[8]);   // a^-1 = sign(f) d, canonical
}

}  // namespace vg
