// provider.h — "provider:path" patterns: a named puzzle stands for its target address (which becomes an
// exact or prefix pattern) and its key range.  Mirrors src/provider.rs:6-62 (ProviderResult, resolve,
// build_pattern, build_exact_pattern).  The reference reads the puzzles from the `boha` crate, an
// un-vendored dependency whose data is not available here; this build carries a static table instead
// (SURVEY.md §8(f)-4): every b1000 puzzle's key range (2^(N-1) .. 2^N - 1 by construction) and the
// target addresses the reference itself pins in its tests / README, extensible from a table file.
#pragma once
#include <stdint.h>

#include <string>

namespace vg {

struct ProviderResult {
    std::string address;
    unsigned format = 0;       // VGF_*
    bool has_range = false;
    uint8_t start[32] = {0};   // big-endian, inclusive
    uint8_t end[32] = {0};
};

// 1 = resolved, 0 = `pattern` is not a provider pattern (no ':' or an unknown provider name, treated
// as a regex exactly as provider.rs:12-20 does), -1 = error (unknown puzzle, bad table; see err).
// table_path: optional CSV file "collection/id,address,kind,start_hex,end_hex" (kind p2pkh | p2wpkh |
// p2tr | p2sh; start/end may be empty); its rows take precedence over the built-in ones.
int provider_resolve(const std::string &pattern, const char *table_path, ProviderResult &out, std::string &err);

// regex::escape: backslash before every regex metacharacter.
std::string regex_escape(const std::string &s);
// "^" + escape(first prefix_length characters)  (provider.rs:54-58; the length is clamped)
std::string provider_build_pattern(const std::string &address, size_t prefix_length);
// "^" + escape(address) + "$"  (provider.rs:60-62)
std::string provider_build_exact_pattern(const std::string &address);

}  // namespace vg
