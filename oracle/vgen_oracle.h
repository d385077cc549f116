/*
 * vgen_oracle.h — CPU restatement of the vgen hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X scan engine: a plain-C restatement of the
 * reference's CPU path  scalar -> secp256k1 point -> SHA-256+RIPEMD-160 / Keccak-256 ->
 * Base58Check / Bech32(m) / EIP-55 hex -> regex match   (reference: src/address.rs:92-151,
 * src/address.rs:176-198, src/pattern.rs:21-45, src/scanner.rs:81-330).
 *
 * The arithmetic the reference calls lives in un-vendored crates (Cargo.lock: bitcoin 0.32.8,
 * secp256k1 0.29.1 / secp256k1-sys 0.10.1, bitcoin_hashes 0.14.1, base58ck 0.1.0,
 * bech32 0.11.1, sha3 0.10.8 / keccak 0.1.5, regex 1.12.2); their published algorithms
 * (SEC2 secp256k1, FIPS 180-4, RIPEMD-160, Keccak[c=512] with 0x01 padding, Base58Check,
 * BIP-173/350, BIP-341, EIP-55) are restated here.
 *
 * Parity pinning: the oracle is checked against the reference's own known-answer test
 * (src/address.rs:232-238), its documented vectors (README.md:102-108,121-127;
 * src/provider.rs:75-87), public standard vectors, and bulk fixtures generated with OpenSSL
 * libcrypto as an independent second source (tests/golden/, script committed).
 *
 * ONLY tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The shipped library (libvgen_hip.so) never does.
 */
#ifndef VGEN_ORACLE_H
#define VGEN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AddressFormat — reference src/address.rs:11-24 (same order). */
enum {
    VO_FMT_P2PKH = 0,
    VO_FMT_P2WPKH = 1,
    VO_FMT_P2SH_P2WPKH = 2,
    VO_FMT_P2TR = 3,
    VO_FMT_P2PKH_UNCOMPRESSED = 4,
    VO_FMT_ETHEREUM = 5
};

/* ---- hashes -------------------------------------------------------------------------- */
void vo_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void vo_ripemd160(const uint8_t *msg, size_t len, uint8_t out[20]);
void vo_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);
void vo_hash160(const uint8_t *msg, size_t len, uint8_t out[20]);

/* ---- secp256k1 ----------------------------------------------------------------------- */
/* 1 if 0 < key < n (SecretKey::from_slice, address.rs:93), else 0. */
int vo_key_valid(const uint8_t key_be[32]);
/* pub65 = 04 || X || Y. Returns 0 for an invalid key.  Uses the windowed fixed-base table. */
int vo_pubkey(const uint8_t key_be[32], uint8_t pub65[65]);
/* Same result by plain MSB-first double-and-add (self-check of the table path). */
int vo_pubkey_naive(const uint8_t key_be[32], uint8_t pub65[65]);
/* out = a + b*? helpers used by tests: 256-bit big-endian add of a u64; returns carry-out. */
int vo_key_add_u64(const uint8_t key_be[32], uint64_t amount, uint8_t out_be[32]);
/* BIP-341 key-path output key for an internal pubkey with no script tree
 * (Address::p2tr(secp, internal_key, None, ..), address.rs:136-140). Returns 0 on failure. */
int vo_taproot_output_key(const uint8_t pub65[65], uint8_t out_x[32]);
/* BIP-340 lift_x (even-Y point for an x-only key); 0 if x is not on the curve. */
int vo_lift_x(const uint8_t x_be[32], uint8_t pub65[65]);
/* SHA-256 chaining state after compressing ONE 64-byte block from the IV (for the TapTweak
 * midstate constant the reference carries in src/shaders/sha256.wgsl:180-183). */
void vo_sha256_midstate(const uint8_t block[64], uint32_t state[8]);

/* ---- encoders ------------------------------------------------------------------------ */
/* All return the string length (no NUL counted); out must hold the stated capacity. */
int vo_base58check(const uint8_t *payload, size_t len, char *out, size_t cap);
int vo_segwit_addr(const char *hrp, int witver, const uint8_t *prog, size_t prog_len, char *out, size_t cap);
int vo_wif(const uint8_t key_be[32], int compressed, char out[64]);
void vo_eip55(const uint8_t addr20[20], char out[43]);

/* hash160/20-byte payload -> address string, as the reference's GPU host loop does
 * (src/gpu.rs:1034-1067).  fmt in {P2PKH, P2PKH_UNCOMPRESSED, P2WPKH, P2SH_P2WPKH, ETHEREUM}. */
int vo_address_from_hash160(int fmt, const uint8_t h160[20], char out[96]);

/* AddressGenerator::generate (address.rs:92-151). Returns 1 and fills the strings, or 0 for
 * an invalid key.  wif = hex for Ethereum (address.rs:110). */
typedef struct {
    char address[96];
    char wif[72];
    char hex[72];
    int format;
} vo_generated;
int vo_generate(int fmt, const uint8_t key_be[32], vo_generated *out);

/* The 20-byte (or 32-byte for P2TR) payload the device kernels emit for a key:
 * hash160(pub33) for P2PKH/P2WPKH, hash160(pub65) for P2PKH_UNCOMPRESSED,
 * hash160(0x0014||hash160(pub33)) for P2SH-P2WPKH, keccak(X||Y)[12..] for Ethereum,
 * the tweaked x-only output key for P2TR.  Returns payload length or 0. */
int vo_payload(int fmt, const uint8_t key_be[32], uint8_t out[32]);

/* ---- pattern (pattern.rs:21-45) ------------------------------------------------------ */
typedef struct vo_regex vo_regex;
/* NULL on empty pattern or unsupported/invalid syntax; err (optional) gets a message. */
vo_regex *vo_regex_new(const char *pattern, int case_insensitive, char *err, size_t errcap);
int vo_regex_is_match(const vo_regex *re, const char *text);
void vo_regex_free(vo_regex *re);

/* ---- scan loops (scanner.rs:81-330) -------------------------------------------------- */
typedef struct {
    uint8_t key[32];
    vo_generated gen;
} vo_match;

typedef struct {
    vo_match *matches;    /* malloc'd; vo_scan_free */
    size_t n_matches;
    uint64_t operations;
    double elapsed_secs;
} vo_scan_result;

/* scan_range_cpu (scanner.rs:211-330): keys start..=end, 10 000 per batch, full scalar
 * multiplication per key, every match kept (not truncated to count), invalid keys skipped
 * and not counted.  threads<=0 -> all cores.  Matches are returned sorted by key. */
int vo_scan_range(int fmt, const char *pattern, int ci, const uint8_t start_be[32],
                  const uint8_t end_be[32], size_t count, int threads, vo_scan_result *out);

/* scan_with_progress (scanner.rs:81-208) with a build-side deterministic key stream
 * (the reference seeds from OS entropy and is not reproducible):
 *   key(seed, thread, i) = SHA-256("vgen-mi355x-rand" || u64le(seed) || u32le(thread) || u64le(i)).
 * Stops once `count` matches are found or `max_keys` keys were tried (0 = no limit). */
int vo_scan_random(int fmt, const char *pattern, int ci, uint64_t seed, size_t count,
                   uint64_t max_keys, int threads, vo_scan_result *out);
void vo_scan_free(vo_scan_result *r);

/* Seeded base scalar frozen in BASELINE.md §4:
 *   k0(seed, shard) = SHA-256("vgen-mi355x" || u64le(seed) || u32le(shard)) mod n, re-drawn if 0 */
void vo_seed_key(uint64_t seed, uint32_t shard, uint8_t out_be[32]);

/* Bulk helper for parity tests: payload of keys start+0 .. start+n-1 (20 or 32 bytes each,
 * zeroed for invalid keys), `threads` worker threads. */
int vo_payload_seq(int fmt, const uint8_t start_be[32], uint64_t n, int threads, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
