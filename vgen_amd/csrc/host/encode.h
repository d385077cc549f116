// encode.h — host-side hashing and address/WIF encoders of libvgen_hip.so.
//
// What the Rust host of the reference gets from rust-bitcoin / bitcoin_hashes / base58ck / bech32 /
// sha3 when it turns a device hash160 back into an address string and a WIF
// (src/gpu.rs:1034-1088, src/address.rs:92-151,168-198).  The block functions are the single-source
// ones in core/hash.h (the same code the kernels run).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace vg {

void host_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
// the same with the choice of block function exposed (tests: the portable path against the SHA-extension path)
void host_sha256_with(const uint8_t *msg, size_t len, uint8_t out[32], bool allow_sha_ni);
void host_ripemd160(const uint8_t *msg, size_t len, uint8_t out[20]);
void host_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);
void host_hash160(const uint8_t *msg, size_t len, uint8_t out[20]);

std::string base58_encode(const uint8_t *data, size_t len);
std::string base58check(const uint8_t *payload, size_t len);
// hrp + '1' + data(5-bit regrouped, version first) + checksum; bech32 for v0, bech32m for v1+
std::string segwit_address(const char *hrp, int witver, const uint8_t *prog, size_t len);
std::string eip55_address(const uint8_t addr20[20]);
std::string hex_lower(const uint8_t *data, size_t len);

// Address string from a device payload; empty string for an unsupported format.
std::string address_from_payload(uint32_t format, const uint8_t *payload);
// WIF (compressed unless P2PKH_UNCOMPRESSED); hex for Ethereum (src/address.rs:110)
std::string key_to_wif(uint32_t format, const uint8_t key_be[32]);
// Host derivation of the device payload for one key (single-key / verify-style use, tests).
// Returns the payload length (20 / 32) or 0 for an invalid key.
int payload_from_key(uint32_t format, const uint8_t key_be[32], uint8_t out[32]);

}  // namespace vg
