// pattern_info.cpp — see pattern_info.h.
#include "pattern_info.h"

#include <ctype.h>
#include <string.h>

#include "../device/device_types.h"

namespace vg {
namespace {

enum class Charset { Base58, Bech32, Hex, None };

Charset charset_of(unsigned format) {
    switch ((int)format) {
    case VGF_P2PKH:
    case VGF_P2PKH_UNCOMPRESSED:
    case VGF_P2SH_P2WPKH: return Charset::Base58;
    case VGF_P2WPKH:
    case VGF_P2TR: return Charset::Bech32;
    case VGF_ETHEREUM: return Charset::Hex;
    default: return Charset::None;
    }
}

// The reference's three alphabets (pattern.rs:50-59).  Hex includes both cases and the 'x' of "0x".
const char *alphabet_of(Charset cs) {
    switch (cs) {
    case Charset::Base58: return "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
    case Charset::Bech32: return "023456789acdefghjklmnpqrstuvwxyz";
    case Charset::Hex: return "0123456789abcdefABCDEFx";
    default: return "";
    }
}

// Insertion-ordered set of ASCII characters.
struct OrderedChars {
    bool seen[256] = {};
    std::string order;
    void add(unsigned char c) {
        if (!seen[c]) {
            seen[c] = true;
            order.push_back((char)c);
        }
    }
    void clear() {
        memset(seen, 0, sizeof seen);
        order.clear();
    }
};

struct Alphabet {
    bool in[256] = {};
    bool fold;
    Alphabet(const char *chars, bool case_insensitive) : fold(case_insensitive) {
        for (const char *p = chars; *p; p++) in[(unsigned char)*p] = true;
    }
    bool allows(unsigned char c) const {
        if (!fold) return in[c];
        return in[(unsigned char)tolower(c)] || in[(unsigned char)toupper(c)];   // pattern.rs:71-78
    }
};

bool is_alnum(unsigned char c) { return c < 128 && isalnum(c); }

// Walks one bracket expression whose '[' has just been consumed; returns the index after its ']'
// (or the end of the pattern).  Members: literals, escaped characters, expanded a-z ranges; a '-'
// is never a member; a '^' negates only as the very first character; a nested '[' restarts the
// expression (pattern.rs:95-102,103-121,122-125,133-170).
size_t walk_class(const std::string &p, size_t i, const Alphabet &alpha, OrderedChars &bad) {
    OrderedChars members;
    bool negated = false, at_start = true, have_prev = false, range_open = false;
    unsigned char prev = 0;
    while (i < p.size()) {
        const unsigned char c = (unsigned char)p[i++];
        if (c == '\\') {
            if (i < p.size()) members.add((unsigned char)p[i++]);
            at_start = false;
            continue;
        }
        if (c == '[') {
            members.clear();
            negated = false, at_start = true, have_prev = false, range_open = false;
            continue;
        }
        if (c == ']') {
            if (!negated) {   // a negated class is almost always satisfiable: never flagged
                bool satisfiable = false;
                for (char m : members.order) satisfiable = satisfiable || alpha.allows((unsigned char)m);
                if (!satisfiable)
                    for (char m : members.order) bad.add((unsigned char)m);
            }
            return i;
        }
        if (c == '^' && at_start) {
            negated = true;
            at_start = false;
            continue;
        }
        at_start = false;
        if (c == '-') {
            range_open = have_prev;   // a leading hyphen is a literal that no alphabet contains
            continue;
        }
        if (is_alnum(c)) {
            if (range_open) {
                const unsigned char lo = prev < c ? prev : c, hi = prev < c ? c : prev;
                for (unsigned v = lo; v <= hi; v++) members.add((unsigned char)v);
                range_open = false;
            } else {
                members.add(c);
            }
            prev = c;
            have_prev = true;
        } else {
            members.add(c);   // '.', '_', '^' after the start, ...: literals inside a class
        }
    }
    return i;   // unterminated: nothing to report (the regex compiler rejects it anyway)
}

}  // namespace

const char *format_charset_name(unsigned format) {
    switch (charset_of(format)) {
    case Charset::Base58: return "Base58";
    case Charset::Bech32: return "Bech32";
    case Charset::Hex: return "Hex";
    default: return nullptr;
    }
}

std::string pattern_invalid_chars(const std::string &p, bool case_insensitive, unsigned format) {
    const Alphabet alpha(alphabet_of(charset_of(format)), case_insensitive);
    OrderedChars bad;
    size_t i = 0;
    while (i < p.size()) {
        const unsigned char c = (unsigned char)p[i++];
        if (c == '\\') {
            i++;   // an escaped character outside a class is not examined (pattern.rs:81-90)
        } else if (c == '[') {
            i = walk_class(p, i, alpha, bad);
        } else if (is_alnum(c) && !alpha.allows(c)) {
            bad.add(c);   // metacharacters and punctuation are not alphanumeric: skipped
        }
    }
    return bad.order;
}

unsigned pattern_fixed_chars(const std::string &p) {
    unsigned n = 0;
    bool in_class = false;
    for (size_t i = 0; i < p.size(); i++) {
        const unsigned char c = (unsigned char)p[i];
        if (c == '\\') {
            i++;
        } else if (c == '[' || c == ']') {
            in_class = c == '[';
        } else if (!in_class && is_alnum(c)) {
            n++;
        }
    }
    return n;
}

uint64_t pattern_difficulty(const std::string &p, bool case_insensitive, unsigned format) {
    const Charset cs = charset_of(format);
    const uint64_t alphabet = cs == Charset::Base58 ? (case_insensitive ? 34 : 58) : cs == Charset::Bech32 ? 32 : 16;
    unsigned fixed = pattern_fixed_chars(p);

    // characters of the format's constant prefix that an anchored pattern spells out (pattern.rs:205-244)
    unsigned shared = 0;
    if (!p.empty() && p[0] == '^') {
        const char *rest = p.c_str() + 1;
        auto starts = [&](const char *lit) { return strncmp(rest, lit, strlen(lit)) == 0; };
        switch ((int)format) {
        case VGF_P2PKH:
        case VGF_P2PKH_UNCOMPRESSED: shared = starts("1"); break;
        case VGF_P2SH_P2WPKH: shared = starts("3"); break;
        case VGF_P2WPKH: shared = starts("bc1q") ? 4 : starts("bc1") ? 3 : starts("bc") ? 2 : starts("b"); break;
        case VGF_P2TR: shared = starts("bc1p") ? 4 : starts("bc1") ? 3 : starts("bc") ? 2 : starts("b"); break;
        case VGF_ETHEREUM: shared = (starts("0x") || starts("0X")) ? 2 : starts("0"); break;
        default: break;
        }
    }
    fixed = fixed > shared ? fixed - shared : 0;
    if (fixed == 0) return 1;
    uint64_t d = 1;
    for (unsigned k = 0; k < fixed; k++) {
        if (d > UINT64_MAX / alphabet) return UINT64_MAX;   // saturating_pow
        d *= alphabet;
    }
    return d;
}

}  // namespace vg
