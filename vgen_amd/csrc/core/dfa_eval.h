// dfa_eval.h — full on-device matching: encode the address of a payload symbol by symbol and walk the
// pattern's DFA over it.  Single source for the kernels and the host (the CPU tests run exactly this
// code against Dfa::is_match on the host-encoded string).
//
// Used for patterns that have no cheap necessary condition on the payload (unanchored literals such as
// the reference's own "1[Oo]ri" example, Base58 suffixes, ...).  The reference matches every key of
// every batch on the host (src/gpu.rs:1030-1093); here the Base58Check conversion (double SHA-256
// checksum + 200-bit base conversion), the Bech32 symbols + checksum and the hex nibbles are produced
// in registers and fed to the DFA whose tables are staged in LDS, so that again only candidates leave
// the device.  Ethereum addresses are matched case-folded (EIP-55 casing needs a second Keccak): the
// device result is then a superset and the host confirms, as for every other candidate.
#pragma once
#include "../device/device_types.h"
#include "filter_eval.h"
#include "hash.h"

namespace vg {

// Blob layout (32-bit words):
//   [0] n_states  [1] n_cls  [2] start state (after the literal head "bc1q" / "0x", 0 for Base58)
//   [3] flags offset (bytes from blob start)  [4] trans offset (bytes)  [5] total bytes  [6..7] reserved
//   [8..23] class of each alphabet symbol, one byte per symbol (58 / 32 / 16 used)
//   flags: one byte per state (bit0 match_now, bit1 match_at_end, bit2 dead);  trans: u16[n_states][n_cls]
constexpr u32 DFA_HDR_WORDS = 24;
constexpr u32 DFA_MAX_BYTES = 48 * 1024;

struct DfaView {
    const u32 *blob;
    const uint8_t *sym_cls;
    const uint8_t *flags;
    const uint16_t *trans;
    u32 n_cls;
};

VG_HD DfaView dfa_view(const u32 *blob) {
    DfaView v;
    v.blob = blob;
    v.n_cls = blob[1];
    const uint8_t *b = reinterpret_cast<const uint8_t *>(blob);
    v.sym_cls = b + 8 * 4;
    v.flags = b + blob[3];
    v.trans = reinterpret_cast<const uint16_t *>(b + blob[4]);
    return v;
}

// state machine state: s = DFA state; once the state is absorbing (match_now / dead) it stays there
// because the tables are built that way, so no early exit is needed for correctness.
VG_HD u32 dfa_step(const DfaView &v, u32 s, u32 symbol) { return v.trans[s * v.n_cls + v.sym_cls[symbol]]; }
VG_HD bool dfa_accept(const DfaView &v, u32 s) { return (v.flags[s] & 3u) != 0; }

// ---- Base58Check ----------------------------------------------------------------------------------------

// checksum = first 4 bytes of SHA-256(SHA-256(version || h160)); H = h160 as five big-endian words
#if defined(__HIP_DEVICE_COMPILE__) && defined(VG_HASH_BLOCKS)
// in the kernels: the two compressions as one scheduled instruction block (device/hashgen.py -> device/hash_blocks.inc)
__device__ __forceinline__ void base58_check_block(u32 version, const u32 H[5], u32 out[1]);
#endif
VG_HD u32 base58_checksum(u32 version, const u32 H[5]) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(VG_HASH_BLOCKS)
    u32 chk[1];
    base58_check_block(version, H, chk);
    return chk[0];
#else
    u32 w[16], st[8];
    w[0] = (version << 24) | (H[0] >> 8);
    w[1] = (H[0] << 24) | (H[1] >> 8);
    w[2] = (H[1] << 24) | (H[2] >> 8);
    w[3] = (H[2] << 24) | (H[3] >> 8);
    w[4] = (H[3] << 24) | (H[4] >> 8);
    w[5] = (H[4] << 24) | 0x00800000u;
#pragma unroll
    for (int i = 6; i < 15; i++) w[i] = 0;
    w[15] = 21 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = SHA256_IV[i];
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = st[i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 32 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = SHA256_IV[i];
    sha256_compress(st, w);
    return st[0];
#endif
}

// (hi:lo) / 58^5 for hi < 58^5: quotient (< 2^32) and remainder.  One double-precision reciprocal
// multiplication (the 62-bit numerator rounds to 53 bits: error in the quotient << 1) and a fix-up.
constexpr u32 B58_D5 = 656356768u;   // 58^5

VG_HD void divmod_d5(u32 hi, u32 lo, u32 &q, u32 &r) {
    const u64 cur = ((u64)hi << 32) | lo;
    const double inv = 1.0 / 656356768.0 * (1.0 - 1e-15);   // never over-estimates
    u32 qe = (u32)((double)cur * inv);
    u64 rem = cur - (u64)qe * B58_D5;
    if (rem >= B58_D5) {
        qe += 1;
        rem -= B58_D5;
    }
    if (rem >= B58_D5) {   // second fix-up: cannot trigger for in-range inputs, kept for robustness
        qe += 1;
        rem -= B58_D5;
    }
    q = qe;
    r = (u32)rem;
}

// Walks the DFA over the Base58Check string of (version, H) without materialising it.
VG_HD bool dfa_match_base58(const u32 *blob, u32 version, const u32 H[5]) {
    const DfaView v = dfa_view(blob);
    u32 W[7];
    W[0] = version;
#pragma unroll
    for (int i = 0; i < 5; i++) W[1 + i] = H[i];
    W[6] = base58_checksum(version, H);

    // leading zero BYTES of the 25-byte payload become leading '1' characters
    u32 z = 0;
    {
        bool run = true;
        if (version != 0) run = false;
        else z = 1;
#pragma unroll
        for (int i = 1; i < 7; i++) {
#pragma unroll
            for (int b = 3; b >= 0; b--) {
                const u32 byte = (W[i] >> (8 * b)) & 255u;
                run = run && byte == 0;
                z += run ? 1u : 0u;
            }
        }
    }
    // 200-bit -> seven chunks of five base-58 digits, least significant chunk first
    // (each division by 58^5 ~ 2^29.3 clears one more leading word of the < 2^200 number, so chunk c only
    // has to visit words c..6, and after six divisions what is left in W[6] is the top chunk)
    u32 chunk[7];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        u32 r = 0;
#pragma unroll
        for (int i = c; i < 7; i++) {
            u32 q;
            divmod_d5(r, W[i], q, r);
            W[i] = q;
        }
        chunk[c] = r;
    }
    chunk[6] = W[6];
    u32 s = blob[2];
    for (u32 t = 0; t < z; t++) s = dfa_step(v, s, 0);   // '1' is digit 0
    bool started = false;
#pragma unroll
    for (int c = 6; c >= 0; c--) {
        u32 r = chunk[c];
        u32 d[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            d[k] = r % 58u;
            r /= 58u;
        }
#pragma unroll
        for (int k = 4; k >= 0; k--) {
            started = started || d[k] != 0;
            const u32 ns = dfa_step(v, s, d[k]);
            s = started ? ns : s;
        }
    }
    return dfa_accept(v, s);
}

// ---- fixed-length symbol strings ------------------------------------------------------------------------------

// Bech32(m) witness programs: NW = 5 -> "bc1q" + 32 data symbols, NW = 8 -> "bc1p" + 52 data symbols (P2TR),
// then 6 checksum symbols.  The literal head is already consumed (blob[2]).
template <int NW>
VG_HD bool dfa_match_bech32(const u32 *blob, const u32 *H, u32 witver) {
    constexpr int NSYM = (NW * 32 + 4) / 5;
    const DfaView v = dfa_view(blob);
    u32 s = blob[2];
#pragma unroll
    for (int k = 0; k < NSYM; k++) s = dfa_step(v, s, bits5<NW>(H, k));
    const u32 chk = bech32_checksum_bc<NW>(H, witver);
#pragma unroll
    for (int k = 0; k < 6; k++) s = dfa_step(v, s, (chk >> (25 - 5 * k)) & 31u);
    return dfa_accept(v, s);
}

// Ethereum: 40 lowercase hex digits after "0x" (DFA compiled case-insensitively for the device).
VG_HD bool dfa_match_hex40(const u32 *blob, const u32 H[5]) {
    const DfaView v = dfa_view(blob);
    u32 s = blob[2];
#pragma unroll
    for (int k = 0; k < 40; k++) s = dfa_step(v, s, (H[k >> 3] >> (28 - 4 * (k & 7))) & 15u);
    return dfa_accept(v, s);
}

// payload: NW words in memory order (5, or 8 for P2TR).  fmt: VGF_* (run time).  KFMT: what the caller knows about fmt at
// compile time — -1 nothing; a kernel instantiated for one payload kind passes its own format, so that it carries only the
// encoders it can need (VGF_P2PKH: hash160 of the compressed key, i.e. P2PKH or P2WPKH, told apart by fmt; every other value: exactly that format).
template <int NW, int KFMT = -1>
VG_HD bool dfa_match_payload_n(const u32 *blob, int fmt, const u32 *payload) {
    u32 H[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) H[i] = bswap32(payload[i]);
    if (NW == 8) return dfa_match_bech32<NW>(blob, H, 1);
    if (KFMT == VGF_ETHEREUM) return dfa_match_hex40(blob, H);
    if (KFMT == VGF_P2WPKH) return dfa_match_bech32<NW>(blob, H, 0);
    if (KFMT == VGF_P2SH_P2WPKH) return dfa_match_base58(blob, 5u, H);
    if (KFMT == VGF_P2PKH_UNCOMPRESSED) return dfa_match_base58(blob, 0u, H);
    if (fmt == VGF_P2WPKH) return dfa_match_bech32<NW>(blob, H, 0);
    if (KFMT == VGF_P2PKH) return dfa_match_base58(blob, 0u, H);
    if (fmt == VGF_ETHEREUM) return dfa_match_hex40(blob, H);
    return dfa_match_base58(blob, fmt == VGF_P2SH_P2WPKH ? 5u : 0u, H);
}

VG_HD bool dfa_match_payload(const u32 *blob, int fmt, const u32 payload[5]) {
    return dfa_match_payload_n<5>(blob, fmt, payload);
}

}  // namespace vg
