// scalar.h — 256-bit private-key scalars on the host (big-endian bytes at the ABI, eight
// little-endian 32-bit words inside; the reference's bytes_be_to_u32_le, src/gpu.rs:891-899).
#pragma once
#include <stdint.h>
#include <string.h>

#include "../core/rnd.h"

namespace vg {

struct Scalar {
    uint32_t w[8];   // w[0] least significant
};

// group order n
static const uint32_t SCALAR_N[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                     0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

inline void scalar_from_be(Scalar &s, const uint8_t be[32]) {
    for (int i = 0; i < 8; i++) {
        const uint8_t *p = be + 28 - 4 * i;
        s.w[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
}

inline void scalar_to_be(const Scalar &s, uint8_t be[32]) {
    for (int i = 0; i < 8; i++) {
        uint8_t *p = be + 28 - 4 * i;
        p[0] = (uint8_t)(s.w[i] >> 24);
        p[1] = (uint8_t)(s.w[i] >> 16);
        p[2] = (uint8_t)(s.w[i] >> 8);
        p[3] = (uint8_t)s.w[i];
    }
}

inline int scalar_cmp_words(const uint32_t a[8], const uint32_t b[8]) {
    for (int i = 7; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}

inline int scalar_cmp(const Scalar &a, const Scalar &b) { return scalar_cmp_words(a.w, b.w); }

inline bool scalar_is_zero(const Scalar &a) {
    uint32_t z = 0;
    for (int i = 0; i < 8; i++) z |= a.w[i];
    return z == 0;
}

// SecretKey::from_slice validity: 0 < k < n (src/address.rs:93)
inline bool scalar_is_valid(const Scalar &a) { return !scalar_is_zero(a) && scalar_cmp_words(a.w, SCALAR_N) < 0; }

// r = a + amount; returns the carry out of bit 255 (plain 256-bit add, src/gpu.rs:951-961)
inline uint32_t scalar_add_u64(Scalar &r, const Scalar &a, uint64_t amount) {
    uint64_t c = amount & 0xFFFFFFFFu;
    uint64_t hi = amount >> 32;
    for (int i = 0; i < 8; i++) {
        c += a.w[i];
        if (i == 1) c += hi;
        r.w[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}

// r = n - a  (a in [1, n-1])
inline void scalar_negate(Scalar &r, const Scalar &a) {
    int64_t b = 0;
    for (int i = 0; i < 8; i++) {
        int64_t d = (int64_t)SCALAR_N[i] - a.w[i] + b;
        r.w[i] = (uint32_t)d;
        b = d >> 32;   // 0 or -1
    }
}

// r = n - 1 - x style helpers are not needed; distance to n as a saturated u64: min(n - a, UINT64_MAX)
inline uint64_t scalar_distance_to_n(const Scalar &a) {
    Scalar d;
    scalar_negate(d, a);   // n - a
    for (int i = 2; i < 8; i++)
        if (d.w[i]) return UINT64_MAX;
    return ((uint64_t)d.w[1] << 32) | d.w[0];
}

// r = a * b mod n.  Schoolbook 256 x 256 -> 512 bits, then 2^256 = c (mod n), c = 2^256 - n (129 bits), folded until
// nothing is left above bit 255 (at most five times) and finished by conditional subtractions.  Host-side, rare (one per reported match of an endomorphism scan).
inline void scalar_mul_mod_n(Scalar &r, const Scalar &a, const Scalar &b) {
    static const uint32_t C[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u};   // 2^256 - n
    uint32_t t[24] = {0};
    for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 8; j++) {
            carry += (uint64_t)a.w[i] * b.w[j] + t[i + j];
            t[i + j] = (uint32_t)carry;
            carry >>= 32;
        }
        t[i + 8] = (uint32_t)carry;
    }
    // value = lo(8 words) + hi * 2^256  ==  lo + hi * C; hi shrinks from 8 to 5 to 2 to 0..1 words
    int hi_words = 8;
    for (int round = 0; round < 6 && hi_words > 0; round++) {
        uint32_t acc[24] = {0};
        for (int i = 0; i < 8; i++) acc[i] = t[i];
        for (int i = 0; i < hi_words; i++) {
            uint64_t carry = 0;
            for (int j = 0; j < 5; j++) {
                carry += (uint64_t)t[8 + i] * C[j] + acc[i + j];
                acc[i + j] = (uint32_t)carry;
                carry >>= 32;
            }
            for (int k = i + 5; carry; k++) {
                carry += acc[k];
                acc[k] = (uint32_t)carry;
                carry >>= 32;
            }
        }
        memcpy(t, acc, sizeof t);
        hi_words = 0;
        for (int i = 23; i >= 8; i--)
            if (t[i]) {
                hi_words = i - 7;
                break;
            }
    }
    for (int i = 0; i < 8; i++) r.w[i] = t[i];
    while (scalar_cmp_words(r.w, SCALAR_N) >= 0) {
        int64_t bw = 0;
        for (int i = 0; i < 8; i++) {
            int64_t d = (int64_t)r.w[i] - SCALAR_N[i] + bw;
            r.w[i] = (uint32_t)d;
            bw = d >> 32;
        }
    }
}

// The secp256k1 endomorphism on scalars: lambda * (x, y) = (beta * x, y), lambda^3 = 1 (mod n).
static const uint32_t SCALAR_LAMBDA[8] = {0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu,
                                          0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu};

// Key variant v of k (0 < k < n) as the endomorphism kernels enumerate them: v % 3 = power of lambda, v >= 3 = negated:
// k, lambda k, lambda^2 k, -k, -lambda k, -lambda^2 k (mod n).  Public keys: (x,y), (bx,y), (b^2 x,y), (x,-y), ...
inline void scalar_variant(Scalar &r, const Scalar &k, uint32_t v) {
    Scalar lam;
    memcpy(lam.w, SCALAR_LAMBDA, sizeof lam.w);
    r = k;
    for (uint32_t e = 0; e < v % 3; e++) {
        Scalar t;
        scalar_mul_mod_n(t, r, lam);
        r = t;
    }
    if (v >= 3) {
        Scalar t;
        scalar_negate(t, r);
        r = t;
    }
}

// Candidate `index` of stream `stream` under `seed` of the counter-based scalar stream (core/rnd.h) as 32 big-endian
// bytes; false when the draw is not a valid scalar (such candidates yield no key, as in the reference's CPU loop).
inline RndSeed rnd_seed_from_bytes(const uint8_t b[24]) {
    RndSeed s;
    for (int i = 0; i < 6; i++) s.w[i] = (uint32_t)b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
    return s;
}
inline void rnd_seed_to_bytes(const RndSeed &s, uint8_t b[24]) {
    for (int i = 0; i < 24; i++) b[i] = (uint8_t)(s.w[i / 4] >> (8 * (i % 4)));
}
inline bool random_key_be(const RndSeed &seed, uint32_t stream, uint64_t index, uint8_t be[32]) {
    Scalar k;
    rnd_scalar(seed, stream, (uint32_t)index, (uint32_t)(index >> 32), k.w);
    scalar_to_be(k, be);
    return scalar_is_valid(k);
}

}  // namespace vg
