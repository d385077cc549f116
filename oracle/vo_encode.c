/*
 * vo_encode.c — Base58Check, Bech32/Bech32m, WIF, EIP-55 and AddressGenerator::generate for
 * the parity oracle (TEST INFRASTRUCTURE ONLY, see vgen_oracle.h).
 *
 * Follows the per-format recipe of the reference, src/address.rs:92-151:
 *   P2PKH (+uncompressed) :121-124  -> Base58Check(0x00 || hash160(pubkey))
 *   P2WPKH               :125-129  -> bech32("bc", v0, hash160(pub33))
 *   P2SH-P2WPKH          :130-135  -> Base58Check(0x05 || hash160(0x0014 || hash160(pub33)))
 *   P2TR                 :136-140  -> bech32m("bc", v1, x(P_even + TapTweak(x)*G))
 *   Ethereum             :95-113   -> "0x" + EIP-55(keccak256(X||Y)[12..]),  wif := hex
 *   WIF                  :117-118,146 -> Base58Check(0x80 || k || [0x01 if compressed])
 * and to_checksum_address, src/address.rs:176-198.  Encodings restate base58ck 0.1.0 and
 * bech32 0.11.1 (BIP-173 / BIP-350).
 */
#include "vgen_oracle.h"
#include "vo_internal.h"

#include <stdio.h>
#include <string.h>

static const char B58_ALPHABET[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
static const char BECH32_CHARSET[] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";

/* ---- Base58 ----------------------------------------------------------------------------- */

static int base58_encode(const uint8_t *in, size_t len, char *out, size_t cap) {
    uint8_t digits[160]; /* little-endian base-58 digits */
    size_t nd = 0;
    if (len > 100) return -1;
    size_t zeros = 0;
    while (zeros < len && in[zeros] == 0) zeros++;
    for (size_t i = zeros; i < len; i++) {
        unsigned carry = in[i];
        for (size_t j = 0; j < nd; j++) {
            carry += (unsigned)digits[j] << 8;
            digits[j] = (uint8_t)(carry % 58);
            carry /= 58;
        }
        while (carry) {
            digits[nd++] = (uint8_t)(carry % 58);
            carry /= 58;
        }
    }
    if (zeros + nd + 1 > cap) return -1;
    size_t o = 0;
    for (size_t i = 0; i < zeros; i++) out[o++] = '1';
    for (size_t i = 0; i < nd; i++) out[o++] = B58_ALPHABET[digits[nd - 1 - i]];
    out[o] = 0;
    return (int)o;
}

int vo_base58check(const uint8_t *payload, size_t len, char *out, size_t cap) {
    uint8_t buf[128], d1[32], d2[32];
    if (len > 96) return -1;
    memcpy(buf, payload, len);
    vo_sha256(payload, len, d1);
    vo_sha256(d1, 32, d2);
    memcpy(buf + len, d2, 4);
    return base58_encode(buf, len + 4, out, cap);
}

/* ---- Bech32 / Bech32m ---------------------------------------------------------------------- */

static uint32_t bech32_polymod_step(uint32_t pre) {
    uint32_t b = pre >> 25;
    return ((pre & 0x1FFFFFF) << 5) ^ (-((b >> 0) & 1) & 0x3b6a57b2UL) ^
           (-((b >> 1) & 1) & 0x26508e6dUL) ^ (-((b >> 2) & 1) & 0x1ea119faUL) ^
           (-((b >> 3) & 1) & 0x3d4233ddUL) ^ (-((b >> 4) & 1) & 0x2a1462b3UL);
}

int vo_segwit_addr(const char *hrp, int witver, const uint8_t *prog, size_t prog_len, char *out,
                   size_t cap) {
    uint8_t data[80];
    size_t nd = 0;
    if (witver < 0 || witver > 16 || prog_len < 2 || prog_len > 40) return -1;
    data[nd++] = (uint8_t)witver;
    /* 8 -> 5 bit regrouping with zero padding */
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < prog_len; i++) {
        acc = (acc << 8) | prog[i];
        bits += 8;
        while (bits >= 5) {
            bits -= 5;
            data[nd++] = (acc >> bits) & 31;
        }
    }
    if (bits) data[nd++] = (acc << (5 - bits)) & 31;

    size_t hl = strlen(hrp);
    uint32_t chk = 1;
    for (size_t i = 0; i < hl; i++) chk = bech32_polymod_step(chk) ^ ((uint8_t)hrp[i] >> 5);
    chk = bech32_polymod_step(chk);
    for (size_t i = 0; i < hl; i++) chk = bech32_polymod_step(chk) ^ (hrp[i] & 31);
    for (size_t i = 0; i < nd; i++) chk = bech32_polymod_step(chk) ^ data[i];
    for (int i = 0; i < 6; i++) chk = bech32_polymod_step(chk);
    chk ^= (witver == 0) ? 1u : 0x2bc830a3u; /* bech32 for v0, bech32m for v1+ */

    if (hl + 1 + nd + 6 + 1 > cap) return -1;
    size_t o = 0;
    memcpy(out, hrp, hl);
    o = hl;
    out[o++] = '1';
    for (size_t i = 0; i < nd; i++) out[o++] = BECH32_CHARSET[data[i]];
    for (int i = 0; i < 6; i++) out[o++] = BECH32_CHARSET[(chk >> (5 * (5 - i))) & 31];
    out[o] = 0;
    return (int)o;
}

/* ---- WIF / hex / EIP-55 --------------------------------------------------------------------- */

int vo_wif(const uint8_t key_be[32], int compressed, char out[64]) {
    uint8_t buf[34];
    buf[0] = 0x80;
    memcpy(buf + 1, key_be, 32);
    size_t n = 33;
    if (compressed) buf[n++] = 0x01;
    return vo_base58check(buf, n, out, 64);
}

static void hex_lower(const uint8_t *in, size_t len, char *out) {
    static const char HX[] = "0123456789abcdef";
    for (size_t i = 0; i < len; i++) {
        out[2 * i] = HX[in[i] >> 4];
        out[2 * i + 1] = HX[in[i] & 15];
    }
    out[2 * len] = 0;
}

void vo_eip55(const uint8_t addr20[20], char out[43]) {
    char lower[41];
    uint8_t h[32];
    hex_lower(addr20, 20, lower);
    vo_keccak256((const uint8_t *)lower, 40, h);
    out[0] = '0';
    out[1] = 'x';
    for (int i = 0; i < 40; i++) {
        int nib = (i & 1) ? (h[i / 2] & 15) : (h[i / 2] >> 4);
        char c = lower[i];
        if (nib >= 8 && c >= 'a' && c <= 'f') c = (char)(c - 'a' + 'A');
        out[2 + i] = c;
    }
    out[42] = 0;
}

/* ---- payloads and addresses ------------------------------------------------------------------ */

static void compress_pub(const uint8_t pub65[65], uint8_t pub33[33]) {
    pub33[0] = (pub65[64] & 1) ? 0x03 : 0x02;
    memcpy(pub33 + 1, pub65 + 1, 32);
}

static int payload_from_pub(int fmt, const uint8_t pub65[65], uint8_t out[32]) {
    uint8_t pub33[33], h[20], script[22], kk[32];
    switch (fmt) {
    case VO_FMT_P2PKH:
    case VO_FMT_P2WPKH:
        compress_pub(pub65, pub33);
        vo_hash160(pub33, 33, out);
        return 20;
    case VO_FMT_P2PKH_UNCOMPRESSED:
        vo_hash160(pub65, 65, out);
        return 20;
    case VO_FMT_P2SH_P2WPKH:
        compress_pub(pub65, pub33);
        vo_hash160(pub33, 33, h);
        script[0] = 0x00;
        script[1] = 0x14;
        memcpy(script + 2, h, 20);
        vo_hash160(script, 22, out);
        return 20;
    case VO_FMT_ETHEREUM:
        vo_keccak256(pub65 + 1, 64, kk);
        memcpy(out, kk + 12, 20);
        return 20;
    case VO_FMT_P2TR:
        if (!vo_taproot_output_key(pub65, out)) return 0;
        return 32;
    default:
        return 0;
    }
}

int vo_payload(int fmt, const uint8_t key_be[32], uint8_t out[32]) {
    uint8_t pub65[65];
    if (!vo_pubkey(key_be, pub65)) return 0;
    return payload_from_pub(fmt, pub65, out);
}

int vo_address_from_hash160(int fmt, const uint8_t h160[20], char out[96]) {
    uint8_t buf[21];
    switch (fmt) {
    case VO_FMT_P2PKH:
    case VO_FMT_P2PKH_UNCOMPRESSED:
        buf[0] = 0x00;
        memcpy(buf + 1, h160, 20);
        return vo_base58check(buf, 21, out, 96);
    case VO_FMT_P2SH_P2WPKH:
        buf[0] = 0x05;
        memcpy(buf + 1, h160, 20);
        return vo_base58check(buf, 21, out, 96);
    case VO_FMT_P2WPKH:
        return vo_segwit_addr("bc", 0, h160, 20, out, 96);
    case VO_FMT_ETHEREUM:
        vo_eip55(h160, out);
        return 42;
    default:
        return -1;
    }
}

int vo_generate(int fmt, const uint8_t key_be[32], vo_generated *out) {
    uint8_t payload[32];
    int n = vo_payload(fmt, key_be, payload);
    if (n == 0) return 0;
    memset(out, 0, sizeof *out);
    out->format = fmt;
    hex_lower(key_be, 32, out->hex);
    if (fmt == VO_FMT_P2TR) {
        if (vo_segwit_addr("bc", 1, payload, 32, out->address, sizeof out->address) < 0) return 0;
    } else {
        if (vo_address_from_hash160(fmt, payload, out->address) < 0) return 0;
    }
    if (fmt == VO_FMT_ETHEREUM)
        strcpy(out->wif, out->hex);
    else
        vo_wif(key_be, fmt != VO_FMT_P2PKH_UNCOMPRESSED, out->wif);
    return 1;
}
