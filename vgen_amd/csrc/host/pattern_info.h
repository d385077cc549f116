// pattern_info.h — the pattern front-end around the matcher: which characters of a pattern can never
// occur in an address of the chosen format, and the "1 in N" difficulty heuristic that feeds warnings
// and time-to-first-match expectations.  Same answers as Pattern::validate_charset
// (src/pattern.rs:49-177) and Pattern::estimate_difficulty (src/pattern.rs:183-253), including their
// documented quirks (digits of a {n} quantifier count as pattern characters, escapes are skipped, ...).
#pragma once
#include <stdint.h>

#include <string>

namespace vg {

// AddressFormat::charset_name (src/address.rs:39-45): "Base58" | "Bech32" | "Hex"; nullptr for an unknown format.
const char *format_charset_name(unsigned format);

// Characters (in order of first appearance) that make the pattern unsatisfiable for `format`:
// literals outside classes that are not in the format's alphabet, and the members of every
// non-negated class none of whose members is.  Patterns are ASCII (regex_dfa.cpp rejects the rest).
std::string pattern_invalid_chars(const std::string &pattern, bool case_insensitive, unsigned format);

// Number of alphanumeric pattern characters outside classes and escapes (count_fixed_chars, pattern.rs:269-293).
unsigned pattern_fixed_chars(const std::string &pattern);

// alphabet ^ (fixed characters - characters of the format's constant prefix when the pattern is
// anchored on it), saturating at 2^64-1; 1 when nothing is fixed.
uint64_t pattern_difficulty(const std::string &pattern, bool case_insensitive, unsigned format);

}  // namespace vg
