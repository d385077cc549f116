// encode.cpp — see encode.h.
#include "encode.h"

#include <string.h>

#include <mutex>
#include <vector>

#include "../core/hash.h"
#include "../core/taproot.h"
#include "../device/device_types.h"
#include "host_ec.h"

namespace vg {

// ---- general-length hashing on top of the single-source block functions ------------------------------

// One SHA-256 block with the x86 SHA extensions (every EPYC and recent Xeon host has them): ~4x the portable block function.
// The host hashes every confirmed match four times (two Base58Check checksums), which is what bounds permissive patterns and
// the reference's own mode (host filtering of full dumps, src/gpu.rs:1030-1093).  Selected at run time from CPUID; the
// portable path (the single-source core/hash.h the kernels compile) stays the fallback and the test's second opinion.
#if defined(__x86_64__)
}  // namespace vg
#include <cpuid.h>
#include <immintrin.h>
namespace vg {
namespace {
bool cpu_has_sha_ni() {
    static const bool have = []() {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
        const bool sha = (b >> 29) & 1u;
        if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
        return sha && ((c >> 19) & 1u) /* SSE4.1 */ && ((c >> 9) & 1u) /* SSSE3 */;
    }();
    return have;
}

__attribute__((target("sha,sse4.1,ssse3"))) void sha256_block_sha_ni(u32 st[8], const uint8_t *p) {
    alignas(16) static const u32 K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    // the instruction wants the state as (A B E F) and (C D G H), most significant lane first
    __m128i tmp = _mm_loadu_si128((const __m128i *)&st[0]);      // D C B A (lanes 3..0)
    __m128i s1 = _mm_loadu_si128((const __m128i *)&st[4]);       // H G F E
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                          // C D A B
    s1 = _mm_shuffle_epi32(s1, 0x1B);                            // E F G H
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                    // A B E F
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                         // C D G H
    const __m128i save0 = s0, save1 = s1;
    __m128i m[4];
    for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * i)), bswap);
    for (int g = 0; g < 16; g++) {
        // four rounds with message words m[g & 3] (already scheduled), then schedule the words four groups ahead
        __m128i wk = _mm_add_epi32(m[g & 3], _mm_load_si128((const __m128i *)&K[4 * g]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
        wk = _mm_shuffle_epi32(wk, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, wk);
        if (g < 12) {
            // W[16 + 4g .. 19 + 4g] from groups g, g+1, g+2, g+3:  msg1(g, g+1) + W[t-7 lanes] then msg2(.., g+3)
            __m128i t = _mm_sha256msg1_epu32(m[g & 3], m[(g + 1) & 3]);
            t = _mm_add_epi32(t, _mm_alignr_epi8(m[(g + 3) & 3], m[(g + 2) & 3], 4));
            m[g & 3] = _mm_sha256msg2_epu32(t, m[(g + 3) & 3]);
        }
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    tmp = _mm_shuffle_epi32(s0, 0x1B);                           // F E B A
    s1 = _mm_shuffle_epi32(s1, 0xB1);                            // D C H G
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                         // D C B A
    s1 = _mm_alignr_epi8(s1, tmp, 8);                            // H G F E
    _mm_storeu_si128((__m128i *)&st[0], s0);
    _mm_storeu_si128((__m128i *)&st[4], s1);
}
}  // namespace
#else
namespace {
bool cpu_has_sha_ni() { return false; }
void sha256_block_sha_ni(u32 *, const uint8_t *) {}
}  // namespace
#endif

void host_sha256_with(const uint8_t *msg, size_t len, uint8_t out[32], bool allow_sha_ni) {
    u32 st[8];
    for (int i = 0; i < 8; i++) st[i] = SHA256_IV[i];
    const bool ni = allow_sha_ni && cpu_has_sha_ni();
    // Whole blocks straight from the message, the tail (with its padding: one or two blocks) from a buffer on the stack: every
    // match is hashed four times on the host (two Base58Check checksums), and a heap buffer per call was a third of that cost.
    auto block = [&](const uint8_t *p) {
        if (ni) {
            sha256_block_sha_ni(st, p);
            return;
        }
        u32 w[16];
        for (int i = 0; i < 16; i++) w[i] = ((u32)p[4 * i] << 24) | ((u32)p[4 * i + 1] << 16) | ((u32)p[4 * i + 2] << 8) | p[4 * i + 3];
        sha256_compress(st, w);
    };
    size_t off = 0;
    for (; off + 64 <= len; off += 64) block(msg + off);
    uint8_t tail[128];
    const size_t rest = len - off, padded = rest < 56 ? 64 : 128;
    memset(tail, 0, sizeof tail);
    if (rest) memcpy(tail, msg + off, rest);
    tail[rest] = 0x80;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[padded - 1 - i] = (uint8_t)(bits >> (8 * i));
    block(tail);
    if (padded == 128) block(tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(st[i] >> 24);
        out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8);
        out[4 * i + 3] = (uint8_t)st[i];
    }
}

void host_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) { host_sha256_with(msg, len, out, true); }

void host_ripemd160(const uint8_t *msg, size_t len, uint8_t out[20]) {
    u32 st[5];
    for (int i = 0; i < 5; i++) st[i] = RMD160_IV[i];
    std::vector<uint8_t> buf(msg, msg + len);
    buf.push_back(0x80);
    while (buf.size() % 64 != 56) buf.push_back(0);
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) buf.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < buf.size(); off += 64) {
        u32 x[16];
        for (int i = 0; i < 16; i++) {
            const uint8_t *p = &buf[off + 4 * i];
            x[i] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
        }
        ripemd160_compress(st, x);
    }
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(st[i] >> (8 * j));
}

void host_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    const size_t RATE = 136;
    u64 a[25];
    memset(a, 0, sizeof a);
    std::vector<uint8_t> buf(msg, msg + len);
    buf.push_back(0x01);
    while (buf.size() % RATE != 0) buf.push_back(0);
    buf.back() |= 0x80;
    for (size_t off = 0; off < buf.size(); off += RATE) {
        for (size_t i = 0; i < RATE / 8; i++) {
            u64 w = 0;
            for (int j = 7; j >= 0; j--) w = (w << 8) | buf[off + 8 * i + j];
            a[i] ^= w;
        }
        keccak_f1600(a);
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(a[i] >> (8 * j));
}

void host_hash160(const uint8_t *msg, size_t len, uint8_t out[20]) {
    uint8_t d[32];
    host_sha256(msg, len, d);
    host_ripemd160(d, 32, out);
}

// ---- encoders --------------------------------------------------------------------------------------------

static const char B58[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
static const char BECH32[] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";

// Base-58 digits by repeated division of the big-endian number by 58^5 (five digits per pass over the words)
// instead of one pass per input byte: a match is encoded twice (address, WIF) on the scan thread.
std::string base58_encode(const uint8_t *data, size_t len) {
    size_t zeros = 0;
    while (zeros < len && data[zeros] == 0) zeros++;
    const size_t n = len - zeros;
    std::string out(zeros, '1');
    if (n == 0) return out;
    if (n > 128) return std::string();   // not an address-sized payload
    uint32_t w[32];
    const size_t nw = (n + 3) / 4;
    for (size_t i = 0; i < nw; i++) w[i] = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t pos = nw * 4 - n + i;   // right-aligned
        w[pos / 4] |= (uint32_t)data[zeros + i] << (8 * (3 - pos % 4));
    }
    const uint64_t D5 = 656356768ull;   // 58^5
    uint8_t digits[256];                // least significant first
    size_t nd = 0, first = 0;
    while (first < nw) {
        uint64_t rem = 0;
        for (size_t i = first; i < nw; i++) {
            const uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / D5);
            rem = cur % D5;
        }
        while (first < nw && w[first] == 0) first++;
        uint32_t r = (uint32_t)rem;
        for (int k = 0; k < 5; k++) {
            digits[nd++] = (uint8_t)(r % 58);
            r /= 58;
        }
    }
    while (nd > 0 && digits[nd - 1] == 0) nd--;   // leading zero digits of the top chunk
    for (size_t i = nd; i-- > 0;) out.push_back(B58[digits[i]]);
    return out;
}

std::string base58check(const uint8_t *payload, size_t len) {
    if (len > 64) return std::string();
    uint8_t buf[68], d1[32], d2[32];
    memcpy(buf, payload, len);
    host_sha256(payload, len, d1);
    host_sha256(d1, 32, d2);
    memcpy(buf + len, d2, 4);
    return base58_encode(buf, len + 4);
}

static uint32_t polymod_step(uint32_t c, uint32_t v) {
    static const uint32_t GEN[5] = {0x3b6a57b2u, 0x26508e6du, 0x1ea119fau, 0x3d4233ddu, 0x2a1462b3u};
    uint32_t b = c >> 25;
    c = ((c & 0x1FFFFFFu) << 5) ^ v;
    for (int i = 0; i < 5; i++)
        if ((b >> i) & 1) c ^= GEN[i];
    return c;
}

std::string segwit_address(const char *hrp, int witver, const uint8_t *prog, size_t len) {
    std::vector<uint8_t> data;
    data.push_back((uint8_t)witver);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < len; i++) {
        acc = (acc << 8) | prog[i];
        bits += 8;
        while (bits >= 5) {
            bits -= 5;
            data.push_back((acc >> bits) & 31);
        }
    }
    if (bits) data.push_back((acc << (5 - bits)) & 31);
    uint32_t c = 1;
    std::string h(hrp);
    for (char ch : h) c = polymod_step(c, (uint8_t)ch >> 5);
    c = polymod_step(c, 0);
    for (char ch : h) c = polymod_step(c, ch & 31);
    for (uint8_t d : data) c = polymod_step(c, d);
    for (int i = 0; i < 6; i++) c = polymod_step(c, 0);
    c ^= witver == 0 ? 1u : 0x2bc830a3u;
    std::string out = h + "1";
    for (uint8_t d : data) out.push_back(BECH32[d]);
    for (int i = 0; i < 6; i++) out.push_back(BECH32[(c >> (5 * (5 - i))) & 31]);
    return out;
}

std::string hex_lower(const uint8_t *data, size_t len) {
    static const char HX[] = "0123456789abcdef";
    std::string s;
    s.reserve(2 * len);
    for (size_t i = 0; i < len; i++) {
        s.push_back(HX[data[i] >> 4]);
        s.push_back(HX[data[i] & 15]);
    }
    return s;
}

std::string eip55_address(const uint8_t addr20[20]) {
    std::string lower = hex_lower(addr20, 20);
    uint8_t h[32];
    host_keccak256((const uint8_t *)lower.data(), 40, h);
    std::string out = "0x";
    for (int i = 0; i < 40; i++) {
        int nib = (i & 1) ? (h[i / 2] & 15) : (h[i / 2] >> 4);
        char c = lower[i];
        if (nib >= 8 && c >= 'a' && c <= 'f') c = (char)(c - 'a' + 'A');
        out.push_back(c);
    }
    return out;
}

std::string address_from_payload(uint32_t format, const uint8_t *payload) {
    uint8_t buf[21];
    switch (format) {
    case VGF_P2PKH:
    case VGF_P2PKH_UNCOMPRESSED:
        buf[0] = 0x00;
        memcpy(buf + 1, payload, 20);
        return base58check(buf, 21);
    case VGF_P2SH_P2WPKH:
        buf[0] = 0x05;
        memcpy(buf + 1, payload, 20);
        return base58check(buf, 21);
    case VGF_P2WPKH:
        return segwit_address("bc", 0, payload, 20);
    case VGF_P2TR:
        return segwit_address("bc", 1, payload, 32);
    case VGF_ETHEREUM:
        return eip55_address(payload);
    default:
        return std::string();
    }
}

std::string key_to_wif(uint32_t format, const uint8_t key_be[32]) {
    if (format == VGF_ETHEREUM) return hex_lower(key_be, 32);
    uint8_t buf[34];
    buf[0] = 0x80;
    memcpy(buf + 1, key_be, 32);
    size_t n = 33;
    if (format != VGF_P2PKH_UNCOMPRESSED) buf[n++] = 0x01;
    return base58check(buf, n);
}

static void fe_to_be32(const fe &a, uint8_t out[32]) {
    u32 w[8];
    fe_to_words(a, w);
    for (int i = 0; i < 8; i++) {
        out[4 * (7 - i)] = (uint8_t)(w[i] >> 24);
        out[4 * (7 - i) + 1] = (uint8_t)(w[i] >> 16);
        out[4 * (7 - i) + 2] = (uint8_t)(w[i] >> 8);
        out[4 * (7 - i) + 3] = (uint8_t)w[i];
    }
}

int payload_from_key(uint32_t format, const uint8_t key_be[32], uint8_t out[32]) {
    Scalar k;
    scalar_from_be(k, key_be);
    if (!scalar_is_valid(k)) return 0;
    ge p;
    if (!host_ec_mul_gen(k, p)) return 0;
    uint8_t pub65[65], pub33[33], h[20], script[22], kk[32];
    pub65[0] = 0x04;
    fe_to_be32(p.x, pub65 + 1);
    fe_to_be32(p.y, pub65 + 33);
    pub33[0] = (pub65[64] & 1) ? 0x03 : 0x02;
    memcpy(pub33 + 1, pub65 + 1, 32);
    switch (format) {
    case VGF_P2PKH:
    case VGF_P2WPKH:
        host_hash160(pub33, 33, out);
        return 20;
    case VGF_P2PKH_UNCOMPRESSED:
        host_hash160(pub65, 65, out);
        return 20;
    case VGF_P2SH_P2WPKH:
        host_hash160(pub33, 33, h);
        script[0] = 0x00;
        script[1] = 0x14;
        memcpy(script + 2, h, 20);
        host_hash160(script, 22, out);
        return 20;
    case VGF_ETHEREUM:
        host_keccak256(pub65 + 1, 64, kk);
        memcpy(out, kk + 12, 20);
        return 20;
    case VGF_P2TR: {
        // BIP-341 key path, no script tree (address.rs:136-140): the single-source device algorithm
        static std::vector<uint32_t> tab;
        static std::once_flag once;
        std::call_once(once, [] { host_gen_table8_limbs(tab); });
        u32 xw[8];
        if (!taproot_output_x(p.x, p.y, tab.data(), xw)) return 0;
        for (int i = 0; i < 8; i++) {
            out[4 * (7 - i)] = (uint8_t)(xw[i] >> 24);
            out[4 * (7 - i) + 1] = (uint8_t)(xw[i] >> 16);
            out[4 * (7 - i) + 2] = (uint8_t)(xw[i] >> 8);
            out[4 * (7 - i) + 3] = (uint8_t)xw[i];
        }
        return 32;
    }
    default:
        return 0;
    }
}

}  // namespace vg
