// regex_dfa.h — pattern front end of libvgen_hip.so: regex subset -> NFA -> DFA.
//
// Stands in for regex::Regex as the reference uses it (src/pattern.rs:21-45): Pattern::new compiles
// "(?i)"+pattern when case-insensitive, Pattern::matches is an UNANCHORED is_match over the address
// string.  Address strings are ASCII, so the DFA works on bytes.  The parser covers the regex crate's syntax as
// far as it can matter on ASCII (flags i m s x U u, word boundaries, Unicode / POSIX classes by their ASCII
// members, nested classes and set operators — see regex_dfa.cpp); what it does not cover (the CRLF flag, Unicode
// class names outside its table) is rejected with an error instead of being approximated.  Zero-width assertions
// are part of the DFA (states carry the kind of the previous symbol), so is_match stays a table walk.
// The DFA is used (a) on the host to confirm every device candidate exactly, (b) by filter.cpp to derive the
// device prefilter (accepted prefixes / fixed suffixes) and (c) on the device itself (core/dfa_eval.h).
#pragma once
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

namespace vg {

struct LazyNfa;   // regex_dfa.cpp

struct Dfa {
    // byte -> symbol class (bytes >= 128 share one class that only '.' and negated classes accept)
    uint8_t cls[256];
    uint32_t n_cls = 0;
    // trans[state * n_cls + class]; state 0 = start
    std::vector<uint32_t> trans;
    // flags per state
    std::vector<uint8_t> match_now;     // pattern already matched (absorbing)
    std::vector<uint8_t> match_at_end;  // matches if the haystack ends here
    std::vector<uint8_t> dead;          // no match reachable any more
    uint32_t n_states = 0;
    // Patterns whose DFA would need more states than regex_compile builds (an end anchor behind a counted wildcard —
    // "a.{20}$" — remembers 2^20 sets of offsets): the tables above stay empty and is_match simulates the NFA, as the regex
    // crate's own fallback engines do (n_states == 0; filter.cpp then has no device test to derive: every key goes to the host).
    std::shared_ptr<const LazyNfa> lazy;

    bool is_match(const char *text) const;
    bool is_match(const std::string &s) const { return is_match(s.c_str()); }
};

// Compiles `pattern`; returns false and sets err on empty/invalid/unsupported patterns
// ("Pattern cannot be empty" / "Invalid regex pattern: ..." as in src/pattern.rs:22-33).
bool regex_compile(const std::string &pattern, bool case_insensitive, Dfa &out, std::string &err);

}  // namespace vg
