// filter_eval.h — evaluation of the device prefilter program on one payload; single source for the
// kernels and for the host (tests run it against the DFA on encoded addresses).
//
// In the reference every hash160 crosses PCIe and the host encodes + regex-matches all of them
// (src/gpu.rs:1030-1093).  Here the device applies a necessary condition derived from the pattern
// and only candidates are reported; the host confirms them with the exact DFA (filter.cpp).
//
// NW = payload words: 5 for the 20-byte payloads (hash160 / Ethereum address), 8 for the 32-byte
// x-only output key of P2TR.
#pragma once
#include "../device/device_types.h"
#include "hash.h"

namespace vg {

// -1 / 0 / +1 for 160-bit big-endian word arrays
VG_HD int cmp160(const u32 a[5], const u32 b[5]) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        int d = (a[i] > b[i]) - (a[i] < b[i]);
        c = (c == 0) ? d : c;
    }
    return c;
}

VG_HD u32 bech32_polymod_step(u32 c, u32 v) {
    const u32 b = c >> 25;
    c = ((c & 0x1FFFFFFu) << 5) ^ v;
    c ^= (0u - ((b >> 0) & 1u)) & 0x3b6a57b2u;
    c ^= (0u - ((b >> 1) & 1u)) & 0x26508e6du;
    c ^= (0u - ((b >> 2) & 1u)) & 0x1ea119fau;
    c ^= (0u - ((b >> 3) & 1u)) & 0x3d4233ddu;
    c ^= (0u - ((b >> 4) & 1u)) & 0x2a1462b3u;
    return c;
}

// k-th 5-bit symbol of an NW-word big-endian bit string (zero padded at the end)
template <int NW>
VG_HD u32 bits5(const u32 *H, int k) {
    const int bit = 5 * k, w = bit >> 5, o = bit & 31;
    if (o <= 27) return (H[w] >> (27 - o)) & 31u;
    const u32 next = (w + 1 < NW) ? H[w + 1] : 0u;
    return ((H[w] << (o - 27)) | (next >> (59 - o))) & 31u;
}

// Bech32 / Bech32m checksum (30 bits; first checksum symbol in bits 29..25) of a witness program given as
// NW big-endian words, hrp "bc": NW = 5 -> 32 data symbols, NW = 8 -> 52 data symbols (4 pad bits).
// `witver` is the witness-version symbol; for witver != 0 the Bech32m constant applies (BIP-350).
template <int NW>
VG_HD u32 bech32_checksum_bc(const u32 *H, u32 witver) {
    constexpr int NSYM = (NW * 32 + 4) / 5;
    u32 c = 1;   // polymod over hrp-expand("bc") = {3, 3, 0, 2, 3}
    c = bech32_polymod_step(c, 3);
    c = bech32_polymod_step(c, 3);
    c = bech32_polymod_step(c, 0);
    c = bech32_polymod_step(c, 2);
    c = bech32_polymod_step(c, 3);
    c = bech32_polymod_step(c, witver);
#pragma unroll
    for (int s = 0; s < NSYM; s++) c = bech32_polymod_step(c, bits5<NW>(H, s));
#pragma unroll
    for (int i = 0; i < 6; i++) c = bech32_polymod_step(c, 0);
    return c ^ (witver == 0 ? 1u : 0x2bc830a3u);
}

VG_HD u32 bech32_checksum_bc20(const u32 H[5], u32 witver) { return bech32_checksum_bc<5>(H, witver); }

// The checksum through the filter's byte tables when present (it is affine in the payload bytes).
template <int NW>
VG_HD u32 filter_bech32_checksum(const DevFilter *f, const u32 *H) {
    if (f->chk_lut) {
        const u32 *lut = f->chk_lut;
        u32 chk = f->chk_base;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            chk ^= lut[(4 * i + 0) * 256 + (H[i] >> 24)];
            chk ^= lut[(4 * i + 1) * 256 + ((H[i] >> 16) & 255u)];
            chk ^= lut[(4 * i + 2) * 256 + ((H[i] >> 8) & 255u)];
            chk ^= lut[(4 * i + 3) * 256 + (H[i] & 255u)];
        }
        return chk;
    }
    return bech32_checksum_bc<NW>(H, f->witver);
}

// payload: NW words in memory order (little-endian words of the byte string).
template <int NW>
VG_HD bool filter_eval_n(const DevFilter *f, const u32 *payload) {
    const u32 kind = f->kind;
    if (kind == DEVF_ALL || kind == DEVF_HOST_ALL || kind == DEVF_DFA) return true;   // DFA: see dfa_eval.h
    u32 H[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) H[i] = bswap32(payload[i]);
    const u32 n = f->count;
    bool hit = false;
    if (kind == DEVF_RANGES) {   // Base58 prefixes: 20-byte payloads only
        // The leading word decides almost every key (a range of a k-character prefix spans ~58^-k of the
        // word space): two compares per range; the five-word comparison runs only for keys whose leading
        // word falls inside [lo_0, hi_0].
        for (u32 t = 0; t < n; t++) {
            const DevFilterTest &T = f->tests[t];
            const bool near = H[0] >= T.a[0] && H[0] <= T.b[0];
            if (VG_ANY_LANE(near)) hit = hit || (near && cmp160(H, T.a) >= 0 && cmp160(H, T.b) <= 0);
        }
        return hit;
    }
    // DEVF_MASKED
    const bool need_chk = (f->flags & DEVF_FLAG_BECH32_CHK) != 0;
    bool any_data = false;
    for (u32 t = 0; t < n; t++) {
        const DevFilterTest &T = f->tests[t];
        u32 diff = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) diff |= (H[i] & T.a[i]) ^ T.b[i];
        any_data = any_data || (diff == 0);
    }
    if (!need_chk) return any_data;
    if (!any_data) return false;         // the checksum only for data-part survivors
    const u32 chk = filter_bech32_checksum<NW>(f, H);
    for (u32 t = 0; t < n; t++) {
        const DevFilterTest &T = f->tests[t];
        u32 diff = (chk & T.chk_mask) ^ T.chk_value;
#pragma unroll
        for (int i = 0; i < NW; i++) diff |= (H[i] & T.a[i]) ^ T.b[i];
        hit = hit || (diff == 0);
    }
    return hit;
}

VG_HD bool filter_eval(const DevFilter *f, const u32 payload[5]) { return filter_eval_n<5>(f, payload); }

}  // namespace vg
