// scalar.h — 256-bit private-key scalars on the host (big-endian bytes at the ABI, eight
// little-endian 32-bit words inside; the reference's bytes_be_to_u32_le, src/gpu.rs:891-899).
#pragma once
#include <stdint.h>
#include <string.h>

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

}  // namespace vg
