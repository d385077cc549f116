// filter.cpp — see filter.h.
#include "filter.h"

#include "../core/dfa_eval.h"
#include "../core/filter_eval.h"

#include <string.h>

#include <algorithm>
#include <set>
#include <vector>

namespace vg {

namespace {

// ---- tiny 256-bit unsigned integer (address payloads are 200 bits) ------------------------------------

struct U256 {
    uint64_t w[4] = {0, 0, 0, 0};
    static U256 from_u64(uint64_t v) {
        U256 r;
        r.w[0] = v;
        return r;
    }
    static U256 pow2(unsigned e) {
        U256 r;
        r.w[e >> 6] = 1ull << (e & 63);
        return r;
    }
};

int cmp(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; i--) {
        if (a.w[i] < b.w[i]) return -1;
        if (a.w[i] > b.w[i]) return 1;
    }
    return 0;
}

U256 add(const U256 &a, const U256 &b) {
    U256 r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)a.w[i] + b.w[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return r;
}

U256 sub(const U256 &a, const U256 &b) {   // a >= b
    U256 r;
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - (uint64_t)bw;
        r.w[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    return r;
}

U256 mul_small(const U256 &a, uint32_t m) {
    U256 r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)a.w[i] * m;
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return r;
}

U256 shr32(const U256 &a) {
    U256 r;
    for (int i = 0; i < 4; i++) r.w[i] = (a.w[i] >> 32) | (i < 3 ? a.w[i + 1] << 32 : 0);
    return r;
}

U256 umax(const U256 &a, const U256 &b) { return cmp(a, b) >= 0 ? a : b; }
U256 umin(const U256 &a, const U256 &b) { return cmp(a, b) <= 0 ? a : b; }

void to_be_words160(const U256 &a, uint32_t out[5]) {
    // a < 2^160
    out[4] = (uint32_t)a.w[0];
    out[3] = (uint32_t)(a.w[0] >> 32);
    out[2] = (uint32_t)a.w[1];
    out[1] = (uint32_t)(a.w[1] >> 32);
    out[0] = (uint32_t)a.w[2];
}

double to_double(const U256 &a) {
    double r = 0;
    for (int i = 3; i >= 0; i--) r = r * 18446744073709551616.0 + (double)a.w[i];
    return r;
}

const char B58[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
const char BECH32[] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";
const char HEXL[] = "0123456789abcdef";

int b58_digit(char c) {
    const char *p = strchr(B58, c);
    return p && c ? (int)(p - B58) : -1;
}

uint32_t step(const Dfa &d, uint32_t s, unsigned char c) { return d.trans[(size_t)s * d.n_cls + d.cls[c]]; }

// Enumerates the address prefixes (over `alphabet`) from DFA state `from` that can still lead to a
// match: a branch ends when the state is absorbing (match_now: every extension matches), dead, or
// `max_depth` symbols long.  Returns false when the search exceeds its budget (pattern does not
// constrain prefixes usefully).
struct Prefix {
    std::string s;
    bool absorbing;
};

bool enumerate_prefixes(const Dfa &d, uint32_t from, const char *alphabet, size_t max_depth, size_t budget,
                        std::vector<Prefix> &out) {
    out.clear();
    struct Item {
        uint32_t state;
        std::string s;
    };
    std::vector<Item> stack;
    stack.push_back({from, ""});
    size_t visited = 0;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (++visited > budget) return false;
        if (d.dead[it.state]) continue;
        if (d.match_now[it.state]) {
            out.push_back({it.s, true});
            continue;
        }
        if (it.s.size() >= max_depth) {
            out.push_back({it.s, false});
            continue;
        }
        for (const char *a = alphabet; *a; a++) stack.push_back({step(d, it.state, (unsigned char)*a), it.s + *a});
    }
    return true;
}

// ---- Base58 (P2PKH / P2SH) ----------------------------------------------------------------------------------

struct Range {
    U256 lo, hi;   // inclusive, as 160-bit hash160 values
};

// Ranges of the 25-byte payload integer V whose Base58Check string starts with `p`.
void prefix_to_ranges(const std::string &p, uint8_t version, std::vector<Range> &out) {
    // a = leading '1's of the prefix, q = the rest
    size_t a = 0;
    while (a < p.size() && p[a] == '1') a++;
    const std::string q = p.substr(a);
    U256 vlo, vhi;   // payload-integer window the version byte allows
    if (version == 0x00) {
        if (a == 0 && !q.empty()) return;   // P2PKH addresses start with '1'
    } else {
        if (a > 0) return;                   // a P2SH payload has no leading zero byte
    }
    std::vector<std::pair<U256, U256>> v;    // payload integer ranges
    const U256 one = U256::from_u64(1);
    if (version == 0x00) {
        if (a > 25) return;
        if (q.empty()) {
            // at least a leading zero bytes: V < 256^(25-a)
            if (a == 0) v.push_back({U256(), sub(U256::pow2(192), one)});
            else v.push_back({U256(), sub(U256::pow2(8 * (25 - (unsigned)a)), one)});
        } else {
            if (a > 24) return;
            // exactly a leading zero bytes
            vlo = U256::pow2(8 * (24 - (unsigned)a));
            vhi = sub(U256::pow2(8 * (25 - (unsigned)a)), one);
        }
    } else {
        vlo = mul_small(U256::pow2(192), version);
        vhi = sub(mul_small(U256::pow2(192), (uint32_t)version + 1), one);
    }
    if (!q.empty()) {
        U256 qv;
        for (char c : q) {
            int dg = b58_digit(c);
            if (dg < 0) return;
            qv = add(mul_small(qv, 58), U256::from_u64((uint64_t)dg));
        }
        U256 lo = qv, hi = qv;   // m = |q| digits: exactly the value q
        // grow the digit count; 35 digits cover 2^200
        for (size_t m = q.size(); m <= 35; m++) {
            U256 l = umax(lo, vlo), h = umin(hi, vhi);
            if (cmp(l, h) <= 0) v.push_back({l, h});
            if (cmp(lo, vhi) > 0) break;
            // next m: [lo*58, hi*58 + 57]; stop before overflowing 256 bits
            if (lo.w[3] >> 56) break;
            lo = mul_small(lo, 58);
            hi = add(mul_small(hi, 58), U256::from_u64(57));
        }
    }
    for (auto &r : v) {
        U256 l = r.first, h = r.second;
        if (version != 0x00) {
            const U256 base = mul_small(U256::pow2(192), version);
            l = sub(l, base);
            h = sub(h, base);
        }
        out.push_back({shr32(l), shr32(h)});   // drop the 4 checksum bytes: superset on the edges
    }
}

void merge_ranges(std::vector<Range> &r) {
    std::sort(r.begin(), r.end(), [](const Range &a, const Range &b) { return cmp(a.lo, b.lo) < 0; });
    std::vector<Range> out;
    const U256 one = U256::from_u64(1);
    for (auto &x : r) {
        if (!out.empty() && cmp(x.lo, add(out.back().hi, one)) <= 0) {
            out.back().hi = umax(out.back().hi, x.hi);
        } else {
            out.push_back(x);
        }
    }
    r.swap(out);
}

void derive_base58(const Dfa &d, uint8_t version, DevFilter &dev, double &sel) {
    dev.kind = DEVF_HOST_ALL;
    dev.count = 0;
    sel = 1.0;
    if (d.match_now[0]) {
        dev.kind = DEVF_ALL;
        return;
    }
    for (size_t depth = 12; depth >= 1; depth--) {
        std::vector<Prefix> prefixes;
        if (!enumerate_prefixes(d, 0, B58, depth, 200000, prefixes)) continue;
        std::vector<Range> ranges;
        for (auto &p : prefixes) prefix_to_ranges(p.s, version, ranges);
        merge_ranges(ranges);
        if (ranges.size() > DEVF_MAX_TESTS) continue;
        double total = 0;
        for (auto &r : ranges) total += to_double(sub(r.hi, r.lo)) + 1.0;
        sel = total / 1.4615016373309029e48;   // 2^160
        if (sel > 0.999999) {
            // everything is a candidate: exact "match all" only if no branch was cut at the depth limit
            bool exact = true;
            for (auto &p : prefixes) exact = exact && p.absorbing;
            dev.kind = exact ? DEVF_ALL : DEVF_HOST_ALL;
            return;
        }
        if (sel > 0.2) {   // ranges that cover most of the space are no prefilter
            bool exact = true;
            for (auto &p : prefixes) exact = exact && p.absorbing;
            if (!exact) {
                dev.kind = DEVF_HOST_ALL;
                sel = 1.0;
                return;
            }
        }
        dev.kind = DEVF_RANGES;
        dev.count = (uint32_t)ranges.size();
        for (size_t i = 0; i < ranges.size(); i++) {
            to_be_words160(ranges[i].lo, dev.tests[i].a);
            to_be_words160(ranges[i].hi, dev.tests[i].b);
        }
        return;
    }
}

// ---- fixed-length symbol strings (Bech32 P2WPKH, hex Ethereum) ---------------------------------------------

struct SymSpec {
    const char *header;     // literal address head fed to the DFA first
    const char *alphabet;   // symbol -> character
    unsigned bits;          // bits per symbol
    unsigned n_data;        // symbols that come from the payload
    unsigned n_chk;         // trailing checksum symbols (Bech32)
    unsigned payload_bits;  // 160, or 256 for P2TR (the last data symbol is then zero-padded)
};

int sym_index(const SymSpec &sp, char c) {
    const char *p = strchr(sp.alphabet, c);
    return p && c ? (int)(p - sp.alphabet) : -1;
}

// Writes symbol `v` at position `pos` (0 = first data symbol) into a test's mask/value.  Returns false
// when the symbol cannot occur there (a set bit in the zero padding of the last data symbol).
bool set_symbol(const SymSpec &sp, DevFilterTest &t, unsigned pos, unsigned v) {
    if (pos < sp.n_data) {
        const unsigned bit = pos * sp.bits;   // from the most significant bit of the payload
        for (unsigned k = 0; k < sp.bits; k++) {
            const unsigned b = bit + k, w = b >> 5, o = 31 - (b & 31);
            const bool one = (v >> (sp.bits - 1 - k)) & 1;
            if (b >= sp.payload_bits) {
                if (one) return false;
                continue;
            }
            t.a[w] |= 1u << o;
            if (one) t.b[w] |= 1u << o;
        }
    } else {
        const unsigned c = pos - sp.n_data;   // checksum symbol c in bits (29 - 5c) .. (25 - 5c)
        const unsigned sh = 25 - 5 * c;
        t.chk_mask |= 31u << sh;
        t.chk_value |= (v & 31u) << sh;
    }
    return true;
}

void derive_symbols(const Dfa &d, const SymSpec &sp, DevFilter &dev, double &sel) {
    dev.kind = DEVF_HOST_ALL;
    dev.count = 0;
    dev.flags = 0;
    sel = 1.0;
    uint32_t s = 0;
    for (const char *h = sp.header; *h; h++) {
        if (d.match_now[s]) break;
        s = step(d, s, (unsigned char)*h);
    }
    if (d.match_now[s]) {
        dev.kind = DEVF_ALL;
        return;
    }
    const unsigned n_total = sp.n_data + sp.n_chk;
    const size_t A = strlen(sp.alphabet);

    // (1) accepted prefixes over the data symbols
    // the shallowest depth that reaches the best selectivity: deeper enumerations of the same language
    // (e.g. "q" + any symbol for ^bc1qq.*) only multiply the tests
    std::vector<Prefix> prefixes;
    size_t pdepth = 0;
    double prefix_sel = 1.0;
    for (size_t depth = 1; depth <= std::min<size_t>(12, sp.n_data); depth++) {
        std::vector<Prefix> p;
        if (!enumerate_prefixes(d, s, sp.alphabet, depth, 100000, p) || p.size() > DEVF_MAX_TESTS) break;
        double sel_d = 0;
        for (auto &q : p) {
            double f = 1.0;
            for (size_t i = 0; i < q.s.size(); i++) f /= (double)A;
            sel_d += f;
        }
        if (pdepth == 0 || sel_d < prefix_sel * (1.0 - 1e-9)) {
            prefixes.swap(p);
            pdepth = depth;
            prefix_sel = sel_d;
        }
    }
    // prefixes that together cover every string (e.g. the 32 one-symbol prefixes of an unanchored or
    // suffix-only pattern) constrain nothing: crossing them with the suffixes would only multiply tests
    bool prefix_constrains = pdepth > 0 && prefix_sel < 0.999999;

    // (2) required suffixes: states reachable after t symbols, then suffixes that can accept
    std::vector<std::set<uint32_t>> reach(n_total + 1);
    reach[0].insert(s);
    for (unsigned t = 0; t < n_total; t++)
        for (uint32_t st : reach[t]) {
            if (d.dead[st]) continue;
            for (size_t a = 0; a < A; a++) reach[t + 1].insert(d.match_now[st] ? st : step(d, st, (unsigned char)sp.alphabet[a]));
        }
    std::vector<std::string> best_suffixes;
    double suffix_sel = 1.0;
    const unsigned kmax = std::min<unsigned>(8, n_total);
    // viable[r][s]: from DFA state s some string of exactly r more symbols ends in acceptance — prunes
    // the suffix enumeration to branches that can still succeed
    std::vector<std::vector<uint8_t>> viable(kmax + 1, std::vector<uint8_t>(d.n_states, 0));
    for (uint32_t st = 0; st < d.n_states; st++) viable[0][st] = d.match_now[st] || d.match_at_end[st];
    for (unsigned r = 1; r <= kmax; r++)
        for (uint32_t st = 0; st < d.n_states; st++) {
            if (d.match_now[st]) {
                viable[r][st] = 1;
                continue;
            }
            for (size_t a = 0; a < A && !viable[r][st]; a++)
                viable[r][st] = viable[r - 1][step(d, st, (unsigned char)sp.alphabet[a])];
        }
    for (unsigned k = 1; k <= kmax; k++) {
        // DFS over suffix strings with state sets
        struct Item {
            std::vector<uint32_t> states;
            std::string s;
        };
        std::vector<std::string> acc;
        std::vector<Item> stack;
        stack.push_back({std::vector<uint32_t>(reach[n_total - k].begin(), reach[n_total - k].end()), ""});
        size_t visited = 0;
        bool overflow = false;
        while (!stack.empty() && !overflow) {
            Item it = std::move(stack.back());
            stack.pop_back();
            if (++visited > 400000) {
                overflow = true;
                break;
            }
            if (it.s.size() == k) {
                bool ok = false;
                for (uint32_t st : it.states) ok = ok || d.match_now[st] || d.match_at_end[st];
                if (ok) {
                    acc.push_back(it.s);
                    if (acc.size() > DEVF_MAX_TESTS) overflow = true;
                }
                continue;
            }
            for (size_t a = 0; a < A; a++) {
                std::vector<uint32_t> nxt;
                const unsigned remaining = k - (unsigned)it.s.size() - 1;
                for (uint32_t st : it.states) {
                    uint32_t n = d.match_now[st] ? st : step(d, st, (unsigned char)sp.alphabet[a]);
                    if (!d.dead[n] && viable[remaining][n]) nxt.push_back(n);
                }
                if (nxt.empty()) continue;
                std::sort(nxt.begin(), nxt.end());
                nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
                stack.push_back({std::move(nxt), it.s + sp.alphabet[a]});
            }
        }
        if (overflow) break;
        double f = (double)acc.size();
        for (unsigned i = 0; i < k; i++) f /= (double)A;
        if (f < suffix_sel) {
            suffix_sel = f;
            best_suffixes = acc;
        }
    }
    const bool suffix_constrains = suffix_sel < 0.999999;

    if (!prefix_constrains && !suffix_constrains) return;   // DEVF_HOST_ALL
    std::vector<Prefix> use_p;
    std::vector<std::string> use_s;
    if (prefix_constrains && suffix_constrains && prefixes.size() * best_suffixes.size() <= DEVF_MAX_TESTS) {
        use_p = prefixes;
        use_s = best_suffixes;
        sel = prefix_sel * suffix_sel;
    } else if (prefix_constrains && (!suffix_constrains || prefix_sel <= suffix_sel)) {
        use_p = prefixes;
        use_s.push_back("");
        sel = prefix_sel;
    } else {
        use_p.push_back({"", false});
        use_s = best_suffixes;
        sel = suffix_sel;
    }
    if (sel > 0.2) {   // not a useful necessary condition: leave it to the full match (DEVF_DFA) / host
        sel = 1.0;
        return;
    }
    dev.kind = DEVF_MASKED;
    dev.count = 0;
    for (auto &p : use_p)
        for (auto &sx : use_s) {
            DevFilterTest t;
            memset(&t, 0, sizeof t);
            bool possible = true;
            for (size_t i = 0; i < p.s.size(); i++)
                possible = set_symbol(sp, t, (unsigned)i, (unsigned)sym_index(sp, p.s[i])) && possible;
            for (size_t i = 0; i < sx.size(); i++)
                possible = set_symbol(sp, t, n_total - (unsigned)sx.size() + (unsigned)i, (unsigned)sym_index(sp, sx[i])) && possible;
            if (!possible) continue;
            if (t.chk_mask) dev.flags |= DEVF_FLAG_BECH32_CHK;
            dev.tests[dev.count++] = t;
        }
    // count == 0 means nothing can match: the kernel then reports no candidates
}

// Packs `d` into the device layout of core/dfa_eval.h.  `head` is the literal every address of the format
// starts with and that the device does not re-walk; `alphabet` maps symbol index -> character.
// Returns false when the automaton does not fit the LDS budget (or 16-bit state numbers).
bool build_dfa_blob(const Dfa &d, const char *head, const char *alphabet, std::vector<uint32_t> &blob) {
    if (d.n_states > 65535) return false;
    uint32_t s = 0;
    for (const char *h = head; *h; h++) s = d.match_now[s] ? s : step(d, s, (unsigned char)*h);
    const size_t flags_off = DFA_HDR_WORDS * 4;
    const size_t trans_off = (flags_off + d.n_states + 3) & ~(size_t)3;
    const size_t total = (trans_off + (size_t)d.n_states * d.n_cls * 2 + 3) & ~(size_t)3;
    if (total > DFA_MAX_BYTES) return false;
    blob.assign(total / 4, 0);
    blob[0] = d.n_states;
    blob[1] = d.n_cls;
    blob[2] = s;
    blob[3] = (uint32_t)flags_off;
    blob[4] = (uint32_t)trans_off;
    blob[5] = (uint32_t)total;
    uint8_t *b = reinterpret_cast<uint8_t *>(blob.data());
    for (size_t i = 0; alphabet[i]; i++) b[8 * 4 + i] = d.cls[(unsigned char)alphabet[i]];
    for (uint32_t st = 0; st < d.n_states; st++)
        b[flags_off + st] = (uint8_t)((d.match_now[st] ? 1 : 0) | (d.match_at_end[st] ? 2 : 0) | (d.dead[st] ? 4 : 0));
    uint16_t *t = reinterpret_cast<uint16_t *>(b + trans_off);
    for (size_t i = 0; i < (size_t)d.n_states * d.n_cls; i++) t[i] = (uint16_t)d.trans[i];
    return true;
}

}  // namespace

bool filter_compile(const std::string &pattern, bool case_insensitive, uint32_t format, vgen_filter &out,
                    std::string &err) {
    out.pattern = pattern;
    out.case_insensitive = case_insensitive;
    out.format = format;
    memset(&out.dev, 0, sizeof out.dev);
    if (!regex_compile(pattern, case_insensitive, out.dfa, err)) return false;
    if (out.dfa.lazy) {
        // no DFA tables (regex_dfa.h): nothing to derive a device test from — the reference's mode, every key to the host
        out.dev.kind = DEVF_HOST_ALL;
        out.selectivity = 1.0;
        return true;
    }
    switch (format) {
    case VGF_P2PKH:
    case VGF_P2PKH_UNCOMPRESSED:
        derive_base58(out.dfa, 0x00, out.dev, out.selectivity);
        if (out.dev.kind == DEVF_HOST_ALL && build_dfa_blob(out.dfa, "", B58, out.dfa_blob)) out.dev.kind = DEVF_DFA;
        break;
    case VGF_P2SH_P2WPKH:
        derive_base58(out.dfa, 0x05, out.dev, out.selectivity);
        if (out.dev.kind == DEVF_HOST_ALL && build_dfa_blob(out.dfa, "", B58, out.dfa_blob)) out.dev.kind = DEVF_DFA;
        break;
    case VGF_P2WPKH: {
        const SymSpec sp = {"bc1q", BECH32, 5, 32, 6, 160};
        derive_symbols(out.dfa, sp, out.dev, out.selectivity);
        out.dev.witver = 0;
        if (out.dev.flags & DEVF_FLAG_BECH32_CHK) {
            // chk(H) = chk(0) ^ XOR_i (chk(only byte i set to b) ^ chk(0)): the polymod is linear over GF(2)
            const u32 zero[5] = {0, 0, 0, 0, 0};
            const u32 base = bech32_checksum_bc20(zero, 0);
            out.chk_lut.assign(20 * 256, 0);
            for (int i = 0; i < 20; i++)
                for (u32 b = 0; b < 256; b++) {
                    u32 H[5] = {0, 0, 0, 0, 0};
                    H[i / 4] = b << (24 - 8 * (i % 4));
                    out.chk_lut[(size_t)i * 256 + b] = bech32_checksum_bc20(H, 0) ^ base;
                }
            out.dev.chk_base = base;
            out.dev.chk_lut = out.chk_lut.data();   // host pointer; the runtime swaps in the device copy
        }
        if (out.dev.kind == DEVF_HOST_ALL && build_dfa_blob(out.dfa, "bc1q", BECH32, out.dfa_blob)) out.dev.kind = DEVF_DFA;
        break;
    }
    case VGF_ETHEREUM: {
        // EIP-55 casing depends on a second Keccak the kernel does not compute: derive the device
        // test from the case-insensitive language (a superset); the host confirms with the exact DFA.
        Dfa folded;
        std::string e2;
        if (!regex_compile(pattern, true, folded, e2)) {
            err = e2;
            return false;
        }
        if (folded.lazy) {
            out.dev.kind = DEVF_HOST_ALL;
            out.selectivity = 1.0;
            return true;
        }
        const SymSpec sp = {"0x", HEXL, 4, 40, 0, 160};
        derive_symbols(folded, sp, out.dev, out.selectivity);
        if (out.dev.kind == DEVF_HOST_ALL && build_dfa_blob(folded, "0x", HEXL, out.dfa_blob)) out.dev.kind = DEVF_DFA;
        break;
    }
    case VGF_P2TR: {
        // bc1p + 52 data symbols (256 bits + 4 pad bits) + 6 Bech32m checksum symbols
        const SymSpec sp = {"bc1p", BECH32, 5, 52, 6, 256};
        derive_symbols(out.dfa, sp, out.dev, out.selectivity);
        out.dev.witver = 1;
        if (out.dev.flags & DEVF_FLAG_BECH32_CHK) {
            const u32 zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const u32 base = bech32_checksum_bc<8>(zero, 1);
            out.chk_lut.assign(32 * 256, 0);
            for (int i = 0; i < 32; i++)
                for (u32 b = 0; b < 256; b++) {
                    u32 H[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    H[i / 4] = b << (24 - 8 * (i % 4));
                    out.chk_lut[(size_t)i * 256 + b] = bech32_checksum_bc<8>(H, 1) ^ base;
                }
            out.dev.chk_base = base;
            out.dev.chk_lut = out.chk_lut.data();
        }
        if (out.dev.kind == DEVF_HOST_ALL && build_dfa_blob(out.dfa, "bc1p", BECH32, out.dfa_blob)) out.dev.kind = DEVF_DFA;
        break;
    }
    default:
        out.dev.kind = DEVF_HOST_ALL;
        out.selectivity = 1.0;
        break;
    }
    if (out.dev.kind == DEVF_DFA) {
        out.dev.dfa_blob = out.dfa_blob.data();   // host pointer; the runtime swaps in the device copy
        out.dev.dfa_bytes = (uint32_t)(out.dfa_blob.size() * 4);
        out.selectivity = -1.0;                   // unknown: the scanner adapts (ring overflow -> host filtering)
    }
    return true;
}

}  // namespace vg
