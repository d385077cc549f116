// regex_dfa.cpp — see regex_dfa.h.
#include "regex_dfa.h"

#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <stdexcept>

namespace vg {

namespace {

// A set of haystack symbols: the 128 ASCII codes + one bit for "any byte >= 128" (address strings are ASCII; the
// bit only keeps negated classes and '.' well defined on arbitrary bytes).
struct CharSet {
    uint64_t bits[2] = {0, 0};
    bool other = false;
    void add(int c) { bits[c >> 6] |= 1ull << (c & 63); }
    void add_range(int lo, int hi) {
        for (int c = lo; c <= hi; c++) add(c);
    }
    bool has(int c) const { return c < 128 ? (bits[c >> 6] >> (c & 63)) & 1 : other; }
    void merge(const CharSet &o) {
        bits[0] |= o.bits[0];
        bits[1] |= o.bits[1];
        other = other || o.other;
    }
    void intersect(const CharSet &o) {
        bits[0] &= o.bits[0];
        bits[1] &= o.bits[1];
        other = other && o.other;
    }
    void subtract(const CharSet &o) {
        bits[0] &= ~o.bits[0];
        bits[1] &= ~o.bits[1];
        other = other && !o.other;
    }
    void symdiff(const CharSet &o) {
        bits[0] ^= o.bits[0];
        bits[1] ^= o.bits[1];
        other = other != o.other;
    }
    void negate() {
        bits[0] = ~bits[0];
        bits[1] = ~bits[1];
        other = !other;
    }
    // simple case folding restricted to what can matter on ASCII text
    void fold_case() {
        for (int c = 'a'; c <= 'z'; c++) {
            int u = c - 'a' + 'A';
            if (has(c) || has(u)) {
                add(c);
                add(u);
            }
        }
    }
};

// Zero-width assertions.  prev / next are the symbol kinds either side of the position.
enum Assert { A_TEXT_START, A_TEXT_END, A_LINE_START, A_LINE_END, A_WORD, A_NOT_WORD, A_WORD_START, A_WORD_END,
              A_WORD_START_HALF, A_WORD_END_HALF };
enum Kind { K_EDGE = 0, K_NEWLINE = 1, K_WORD = 2, K_OTHER = 3, K_UNKNOWN = 4 };   // K_EDGE: start / end of the haystack

bool is_word_byte(int c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
Kind kind_of(int c) { return c == '\n' ? K_NEWLINE : (c < 128 && is_word_byte(c)) ? K_WORD : K_OTHER; }

bool needs_next(Assert a) { return !(a == A_TEXT_START || a == A_LINE_START || a == A_WORD_START_HALF); }

bool assert_holds(Assert a, Kind prev, Kind next) {
    const bool pw = prev == K_WORD, nw = next == K_WORD;
    switch (a) {
    case A_TEXT_START: return prev == K_EDGE;
    case A_TEXT_END: return next == K_EDGE;
    case A_LINE_START: return prev == K_EDGE || prev == K_NEWLINE;
    case A_LINE_END: return next == K_EDGE || next == K_NEWLINE;
    case A_WORD: return pw != nw;
    case A_NOT_WORD: return pw == nw;
    case A_WORD_START: return !pw && nw;
    case A_WORD_END: return pw && !nw;
    case A_WORD_START_HALF: return !pw;
    case A_WORD_END_HALF: return !nw;
    }
    return false;
}

struct Node;
using NodeP = std::shared_ptr<Node>;
struct Node {
    enum Type { EMPTY, SET, CAT, ALT, STAR, PLUS, QUEST, ASSERT, REPEAT } type = EMPTY;
    CharSet set;
    Assert as = A_TEXT_START;
    NodeP a, b;
    int lo = 0, hi = 0;   // REPEAT; hi < 0 = unbounded
};

NodeP mk(Node::Type t, NodeP a = nullptr, NodeP b = nullptr) {
    auto n = std::make_shared<Node>();
    n->type = t;
    n->a = std::move(a);
    n->b = std::move(b);
    return n;
}

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- parser: the syntax of the regex crate (regex-syntax 0.8) as far as it can matter on ASCII haystacks ------
//
// flags i m s x U u (inline, scoped and negated), groups (capturing, named, non-capturing), alternation,
// repetition (* + ? {n} {n,} {n,m}, lazy forms: laziness cannot change is_match), '.', anchors ^ $ \A \z,
// word boundaries \b \B \< \> \b{start} \b{end} \b{start-half} \b{end-half}, escapes (\n \t ... \xHH \x{..}
// \uHHHH \u{..} \U........), Perl classes \d \s \w, Unicode classes \pX \p{Name} \P{..} \p{^..} \p{gc=..}
// restricted to their ASCII members, bracket classes with ranges, negation, nesting, POSIX names [[:alpha:]] and
// the set operators && -- ~~.  Non-ASCII literals are accepted: they match no ASCII symbol.
// Not supported (an error, never a guess): the CRLF flag R, Unicode class names outside the table below,
// \p{..} value forms other than gc= / sc= / scx=.
struct Flags {
    bool i = false, m = false, s = false, x = false, U = false;
};

struct PropName {
    const char *name;
    const char *members;   // ASCII members as ranges "az" pairs; "" = no ASCII member
};

// Unicode general categories, scripts and binary properties -> their ASCII members (UCD 15; names compared
// loosely: case, '_', '-' and spaces ignored, as UAX44-LM3 / the regex crate do).
const PropName PROPS[] = {
    {"any", "\x01\x7f\x00\x00"}, {"ascii", "\x01\x7f\x00\x00"}, {"assigned", "\x01\x7f\x00\x00"},
    {"l", "AZaz"}, {"letter", "AZaz"}, {"lc", "AZaz"}, {"casedletter", "AZaz"}, {"alphabetic", "AZaz"}, {"alpha", "AZaz"},
    {"lu", "AZ"}, {"uppercaseletter", "AZ"}, {"uppercase", "AZ"}, {"upper", "AZ"},
    {"ll", "az"}, {"lowercaseletter", "az"}, {"lowercase", "az"}, {"lower", "az"},
    {"cased", "AZaz"}, {"caseignorable", "''..::^^``"},
    {"lt", ""}, {"titlecaseletter", ""}, {"lm", ""}, {"modifierletter", ""}, {"lo", ""}, {"otherletter", ""},
    {"m", ""}, {"mark", ""}, {"mn", ""}, {"nonspacingmark", ""}, {"mc", ""}, {"spacingmark", ""}, {"me", ""}, {"enclosingmark", ""},
    {"n", "09"}, {"number", "09"}, {"nd", "09"}, {"decimalnumber", "09"}, {"digit", "09"},
    {"nl", ""}, {"letternumber", ""}, {"no", ""}, {"othernumber", ""},
    {"p", "!#%*,/:;?@[]__{{}}"}, {"punctuation", "!#%*,/:;?@[]__{{}}"}, {"punct", "!#%*,/:;?@[]__{{}}"},
    {"pc", "__"}, {"connectorpunctuation", "__"}, {"pd", "--"}, {"dashpunctuation", "--"},
    {"ps", "(([[{{"}, {"openpunctuation", "(([[{{"}, {"pe", "))]]}}"}, {"closepunctuation", "))]]}}"},
    {"pi", ""}, {"initialpunctuation", ""}, {"pf", ""}, {"finalpunctuation", ""},
    {"po", "!#%'**,,./:;?@\\\\"}, {"otherpunctuation", "!#%'**,,./:;?@\\\\"},
    {"s", "$$++<>^^``||~~"}, {"symbol", "$$++<>^^``||~~"}, {"sm", "++<>||~~"}, {"mathsymbol", "++<>||~~"},
    {"sc", "$$"}, {"currencysymbol", "$$"}, {"sk", "^^``"}, {"modifiersymbol", "^^``"}, {"so", ""}, {"othersymbol", ""},
    {"z", "  "}, {"separator", "  "}, {"zs", "  "}, {"spaceseparator", "  "}, {"zl", ""}, {"lineseparator", ""},
    {"zp", ""}, {"paragraphseparator", ""},
    {"c", "\x01\x1f\x7f\x7f"}, {"other", "\x01\x1f\x7f\x7f"}, {"cc", "\x01\x1f\x7f\x7f"}, {"control", "\x01\x1f\x7f\x7f"}, {"cntrl", "\x01\x1f\x7f\x7f"},
    {"cf", ""}, {"format", ""}, {"cs", ""}, {"surrogate", ""}, {"co", ""}, {"privateuse", ""}, {"cn", ""}, {"unassigned", ""},
    {"whitespace", "\x09\x0d  "}, {"space", "\x09\x0d  "}, {"wspace", "\x09\x0d  "},
    {"hexdigit", "09AFaf"}, {"hex", "09AFaf"}, {"asciihexdigit", "09AFaf"}, {"ahex", "09AFaf"},
    {"latin", "AZaz"}, {"latn", "AZaz"},
    {"common", "\x01@[`{\x7f"}, {"zyyy", "\x01@[`{\x7f"},
    {"greek", ""}, {"grek", ""}, {"cyrillic", ""}, {"cyrl", ""}, {"han", ""}, {"hani", ""}, {"arabic", ""}, {"arab", ""},
    {"hebrew", ""}, {"hebr", ""}, {"hiragana", ""}, {"hira", ""}, {"katakana", ""}, {"kana", ""}, {"hangul", ""}, {"hang", ""},
    {"thai", ""}, {"devanagari", ""}, {"deva", ""}, {"armenian", ""}, {"armn", ""}, {"georgian", ""}, {"geor", ""},
    {"inherited", ""}, {"zinh", ""}, {"unknown", ""}, {"zzzz", ""}, {"emoji", "##**09"}, {"idstart", "AZaz"}, {"idcontinue", "09AZ__az"},
    {"xidstart", "AZaz"}, {"xidcontinue", "09AZ__az"}, {"math", "++<>^^||~~"}, {"dash", "--"}, {"quotationmark", "\"\"''"},
    {"patternwhitespace", "\x09\x0d  "}, {"patternsyntax", "!/:@[^``{~"}, {"terminalpunctuation", "!!,,..:;??"},
};

// Scripts without any ASCII member (Unicode 16 long names; the frequent ones with their four-letter codes are in PROPS):
// \p{Tamil} matches no address character, \P{Tamil} every one.
const char *const SCRIPTS_WITHOUT_ASCII[] = {
    "adlam", "ahom", "anatolianhieroglyphs", "avestan", "balinese", "bamum", "bassavah", "batak", "bengali",
    "bhaiksuki", "bopomofo", "brahmi", "braille", "buginese", "buhid", "canadianaboriginal", "carian",
    "caucasianalbanian", "chakma", "cham", "cherokee", "chorasmian", "coptic", "cuneiform", "cypriot",
    "cyprominoan", "deseret", "divesakuru", "dogra", "duployan", "egyptianhieroglyphs", "elbasan", "elymaic",
    "ethiopic", "garay", "glagolitic", "gothic", "grantha", "gujarati", "gunjalagondi", "gurmukhi", "gurungkhema",
    "hanifirohingya", "hanunoo", "hatran", "imperialaramaic", "inscriptionalpahlavi", "inscriptionalparthian",
    "javanese", "kaithi", "kannada", "kawi", "kayahli", "kharoshthi", "khitansmallscript", "khmer", "khojki",
    "khudawadi", "kiratrai", "lao", "lepcha", "limbu", "lineara", "linearb", "lisu", "lycian", "lydian", "mahajani",
    "makasar", "malayalam", "mandaic", "manichaean", "marchen", "masaramgondi", "medefaidrin", "meeteimayek",
    "mendekikakui", "meroiticcursive", "meroitichieroglyphs", "miao", "modi", "mongolian", "mro", "multani",
    "myanmar", "nabataean", "nagmundari", "nandinagari", "newa", "newtailue", "nko", "nushu",
    "nyiakengpuachuehmong", "ogham", "olchiki", "oldhungarian", "olditalic", "oldnortharabian", "oldpermic",
    "oldpersian", "oldsogdian", "oldsoutharabian", "oldturkic", "olduyghur", "olonal", "oriya", "osage", "osmanya",
    "pahawhhmong", "palmyrene", "paucinhau", "phagspa", "phoenician", "psalterpahlavi", "rejang", "runic",
    "samaritan", "saurashtra", "sharada", "shavian", "siddham", "signwriting", "sinhala", "sogdian", "sorasompeng",
    "soyombo", "sundanese", "sunuwar", "sylotinagri", "syriac", "tagalog", "tagbanwa", "taile", "taitham",
    "taiviet", "takri", "tamil", "tangsa", "tangut", "telugu", "thaana", "tibetan", "tifinagh", "tirhuta", "todhri",
    "toto", "tulutigalari", "ugaritic", "vai", "vithkuqi", "wancho", "warangciti", "yezidi", "yi", "zanabazarsquare"};

// NUL cannot be written in the member strings above: U+0000 is added for the classes that contain it.
bool prop_has_nul(const std::string &n) {
    return n == "any" || n == "ascii" || n == "assigned" || n == "c" || n == "other" || n == "cc" || n == "control" || n == "cntrl" ||
           n == "common" || n == "zyyy";
}

class Parser {
  public:
    Parser(const std::string &p, bool ci) : s_(p) { top_.i = ci; }

    NodeP parse() {
        Flags f = top_;
        NodeP n = alternation(f);
        if (pos_ < s_.size()) throw ParseError(s_[pos_] == ')' ? "unopened group" : "unexpected character");
        return n;
    }

  private:
    const std::string &s_;
    size_t pos_ = 0;
    Flags top_;
    int depth_ = 0;

    bool more() const { return pos_ < s_.size(); }
    int peek(size_t k = 0) const { return pos_ + k < s_.size() ? (unsigned char)s_[pos_ + k] : -1; }
    int take() { return (unsigned char)s_[pos_++]; }
    bool looking_at(const char *lit) const { return s_.compare(pos_, strlen(lit), lit) == 0; }

    // verbose mode: whitespace and # comments are not part of the pattern
    void skip_space(const Flags &f) {
        if (!f.x) return;
        while (more()) {
            int c = peek();
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') {
                pos_++;
            } else if (c == '#') {
                while (more() && peek() != '\n') pos_++;
            } else {
                break;
            }
        }
    }

    // one scalar value of the UTF-8 pattern text
    int take_scalar() {
        int c = take();
        if (c < 0x80) return c;
        int n = c >= 0xF0 ? 3 : c >= 0xE0 ? 2 : c >= 0xC0 ? 1 : -1;
        if (n < 0) throw ParseError("pattern is not valid UTF-8");
        int cp = c & (0x3F >> n);
        for (int k = 0; k < n; k++) {
            if ((peek() & 0xC0) != 0x80) throw ParseError("pattern is not valid UTF-8");
            cp = (cp << 6) | (take() & 0x3F);
        }
        return cp;
    }

    // adds scalar value cp (and, case-insensitively, what folds to it) to cs
    static void add_scalar(CharSet &cs, int cp, bool ci) {
        if (cp < 128) {
            cs.add(cp);
        } else {
            cs.other = true;
            if (ci && cp == 0x17F) cs.add('s');   // LATIN SMALL LETTER LONG S folds to s
            if (ci && cp == 0x212A) cs.add('k');  // KELVIN SIGN folds to k
        }
        if (ci) cs.fold_case();
    }

    static void perl_class(CharSet &out, int kind) {
        CharSet t;
        switch (kind | 0x20) {
        case 'd': t.add_range('0', '9'); break;
        case 'w':
            t.add_range('0', '9');
            t.add_range('a', 'z');
            t.add_range('A', 'Z');
            t.add('_');
            break;
        default:   // 's' (White_Space)
            t.add_range('\t', '\r');
            t.add(' ');
            break;
        }
        if (kind >= 'A' && kind <= 'Z') {
            t.negate();
            t.other = true;
        }
        out.merge(t);
    }

    static std::string loose(const std::string &n) {
        std::string o;
        for (char ch : n)
            if (ch != '_' && ch != '-' && ch != ' ') o += (char)(ch >= 'A' && ch <= 'Z' ? ch + 32 : ch);
        return o;
    }

    static bool lookup_prop(const std::string &name, CharSet &t) {
        const std::string n = loose(name);
        for (const char *sc : SCRIPTS_WITHOUT_ASCII)
            if (n == sc) {
                t.other = true;
                return true;
            }
        for (const PropName &p : PROPS)
            if (n == p.name) {
                for (const char *m = p.members; m[0]; m += 2) t.add_range((unsigned char)m[0], (unsigned char)m[1]);
                if (prop_has_nul(n)) t.add(0);
                // every Unicode class except ASCII also has (or, for empty ASCII parts, only has) members beyond ASCII
                t.other = n != "ascii" && n != "asciihexdigit" && n != "ahex";
                return true;
            }
        return false;
    }

    // after "\p" or "\P"
    void unicode_class(CharSet &out, bool negated) {
        std::string name;
        if (peek() == '{') {
            pos_++;
            while (more() && peek() != '}') name += (char)take();
            if (!more()) throw ParseError("unclosed Unicode class");
            pos_++;
        } else {
            if (!more()) throw ParseError("incomplete Unicode class");
            name = std::string(1, (char)take());
        }
        if (!name.empty() && name[0] == '^') {
            negated = !negated;
            name.erase(0, 1);
        }
        size_t eq = name.find_first_of("=:");
        if (eq != std::string::npos) {
            bool ne = eq > 0 && name[eq - 1] == '!';
            std::string key = loose(name.substr(0, ne ? eq - 1 : eq));
            if (key != "gc" && key != "generalcategory" && key != "sc" && key != "script" && key != "scx" && key != "scriptextensions")
                throw ParseError("unsupported Unicode property");
            if (ne) negated = !negated;
            name = name.substr(eq + 1);
        }
        CharSet t;
        if (!lookup_prop(name, t)) throw ParseError("unsupported Unicode class name");
        if (negated) t.negate();
        out.merge(t);
    }

    static int hexval(int c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    int hex_escape(int digits) {
        int v = 0;
        if (peek() == '{') {
            pos_++;
            int n = 0;
            while (more() && peek() != '}') {
                int h = hexval(take());
                if (h < 0 || ++n > 8) throw ParseError("bad hexadecimal escape");
                v = v * 16 + h;
            }
            if (!more() || n == 0) throw ParseError("bad hexadecimal escape");
            pos_++;
        } else {
            for (int k = 0; k < digits; k++) {
                int h = hexval(peek());
                if (h < 0) throw ParseError("bad hexadecimal escape");
                pos_++;
                v = v * 16 + h;
            }
        }
        if (v > 0x10FFFF || (v >= 0xD800 && v <= 0xDFFF)) throw ParseError("escape is not a Unicode scalar value");
        return v;
    }

    // After the backslash.  Returns a scalar value, or -2 after merging a class (\d \pL ...) into cls.
    int escape(CharSet &cls, const Flags &f) {
        if (!more()) throw ParseError("trailing backslash");
        int c = take_scalar();
        switch (c) {
        case 'd': case 'D': case 'w': case 'W': case 's': case 'S':
            perl_class(cls, c);
            return -2;
        case 'p': case 'P':
            unicode_class(cls, c == 'P');
            return -2;
        case 'n': return '\n';
        case 't': return '\t';
        case 'r': return '\r';
        case 'f': return '\f';
        case 'v': return '\v';
        case 'a': return 7;
        case 'x': return hex_escape(2);
        case 'u': return hex_escape(4);
        case 'U': return hex_escape(8);
        case ' ':
            if (f.x) return ' ';
            throw ParseError("unrecognized escape sequence");
        default:
            if (c >= 128 || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'))
                throw ParseError("unrecognized escape sequence");
            return c;   // escaped punctuation
        }
    }

    // "[:name:]" / "[:^name:]" at pos_ (just after the '[' of the item)?  Merges it and returns true.
    bool posix_class(CharSet &cs) {
        if (peek() != ':') return false;
        size_t end = s_.find(":]", pos_ + 1);
        if (end == std::string::npos) return false;
        std::string name = s_.substr(pos_ + 1, end - pos_ - 1);
        bool neg = !name.empty() && name[0] == '^';
        if (neg) name.erase(0, 1);
        CharSet t;
        if (name == "alnum") { t.add_range('0', '9'); t.add_range('A', 'Z'); t.add_range('a', 'z'); }
        else if (name == "alpha") { t.add_range('A', 'Z'); t.add_range('a', 'z'); }
        else if (name == "ascii") t.add_range(0, 127);
        else if (name == "blank") { t.add(' '); t.add('\t'); }
        else if (name == "cntrl") { t.add_range(0, 31); t.add(127); }
        else if (name == "digit") t.add_range('0', '9');
        else if (name == "graph") t.add_range('!', '~');
        else if (name == "lower") t.add_range('a', 'z');
        else if (name == "print") t.add_range(' ', '~');
        else if (name == "punct") { t.add_range('!', '/'); t.add_range(':', '@'); t.add_range('[', '`'); t.add_range('{', '~'); }
        else if (name == "space") { t.add_range('\t', '\r'); t.add(' '); }
        else if (name == "upper") t.add_range('A', 'Z');
        else if (name == "word") { t.add_range('0', '9'); t.add_range('A', 'Z'); t.add_range('a', 'z'); t.add('_'); }
        else if (name == "xdigit") { t.add_range('0', '9'); t.add_range('A', 'F'); t.add_range('a', 'f'); }
        else return false;   // not a POSIX class: an ordinary nested class starting with ':'
        if (neg) {
            t.negate();
            t.other = true;
        }
        cs.merge(t);
        pos_ = end + 2;
        return true;
    }

    // Bracketed class; pos_ is just after '['.  Precedence (regex-syntax): ranges, then union, then && -- ~~ left to
    // right, then negation.  Case folding applies to both operands of a set operator and to every bracketed
    // class before its negation, as the crate's translator does.
    CharSet char_class(const Flags &f) {
        if (++depth_ > 64) throw ParseError("class nesting too deep");
        bool neg = false;
        skip_space(f);
        if (peek() == '^') {
            neg = true;
            pos_++;
        }
        CharSet acc, cur;
        int pending = 0;   // 0 none, '&' '-' '~'
        bool have_acc = false, first = true;
        auto combine = [&]() {
            if (f.i) cur.fold_case();
            if (!have_acc) {
                acc = cur;
                have_acc = true;
            } else {
                if (f.i) acc.fold_case();
                if (pending == '&') acc.intersect(cur);
                else if (pending == '-') acc.subtract(cur);
                else acc.symdiff(cur);
            }
            cur = CharSet();
        };
        for (;;) {
            skip_space(f);
            if (!more()) throw ParseError("unclosed character class");
            int c = peek();
            if (c == ']' && !first) {
                pos_++;
                break;
            }
            if (!first && (looking_at("&&") || looking_at("--") || looking_at("~~"))) {
                combine();
                pending = c;
                pos_ += 2;
                continue;
            }
            first = false;
            if (c == '[') {
                pos_++;
                if (posix_class(cur)) continue;
                cur.merge(char_class(f));
                continue;
            }
            int lo;
            if (c == '\\') {
                pos_++;
                lo = escape(cur, f);
                if (lo == -2) continue;
            } else {
                lo = take_scalar();
            }
            int hi = lo;
            skip_space(f);
            if (peek() == '-' && peek(1) != -1 && peek(1) != ']' && !looking_at("--")) {
                pos_++;
                skip_space(f);
                if (peek() == '\\') {
                    pos_++;
                    CharSet dummy;
                    hi = escape(dummy, f);
                    if (hi < 0) throw ParseError("invalid class range");
                } else if (peek() == '[') {
                    throw ParseError("invalid class range");
                } else {
                    hi = take_scalar();
                }
                if (hi < lo) throw ParseError("invalid class range");
            }
            if (lo < 128) cur.add_range(lo, std::min(hi, 127));
            if (hi >= 128) {
                cur.other = true;
                if (f.i && lo <= 0x17F && hi >= 0x17F) cur.add('s');
                if (f.i && lo <= 0x212A && hi >= 0x212A) cur.add('k');
            }
        }
        combine();
        if (f.i) acc.fold_case();
        if (neg) acc.negate();
        depth_--;
        return acc;
    }

    int integer(const Flags &f) {
        skip_space(f);
        if (peek() < '0' || peek() > '9') return -1;
        long v = 0;
        while (peek() >= '0' && peek() <= '9') {
            v = v * 10 + (take() - '0');
            if (v > 1000) throw ParseError("repetition count too large");
        }
        skip_space(f);
        return (int)v;
    }

    static NodeP set_node(const CharSet &cs) {
        NodeP n = mk(Node::SET);
        n->set = cs;
        return n;
    }
    static NodeP assert_node(Assert a) {
        NodeP n = mk(Node::ASSERT);
        n->as = a;
        return n;
    }

    // "(?flags)" / "(?flags:" — pos_ is after "(?"; applies the flags to f.  Returns true for the directive form,
    // false after consuming the ':' of a scoped group.
    bool flag_group(Flags &f) {
        bool on = true, any = false, any_neg = false;
        Flags nf = f;
        while (more() && peek() != ':' && peek() != ')') {
            int c = take();
            switch (c) {
            case '-':
                if (!on) throw ParseError("repeated negation in flag group");
                on = false;
                continue;
            case 'i': nf.i = on; break;
            case 'm': nf.m = on; break;
            case 's': nf.s = on; break;
            case 'x': nf.x = on; break;
            case 'U': nf.U = on; break;
            case 'u': break;   // Unicode mode: no difference on ASCII haystacks
            case 'R': throw ParseError("CRLF mode (flag R) unsupported");
            default: throw ParseError("unrecognized flag");
            }
            any = true;
            if (!on) any_neg = true;
        }
        if (!more()) throw ParseError("unclosed group");
        if (!on && !any_neg) throw ParseError("dangling flag negation");
        if (peek() == ')') {
            if (!any) throw ParseError("empty flag group");
            pos_++;
            f = nf;
            return true;
        }
        pos_++;   // ':'
        f = nf;
        return false;
    }

    NodeP atom(Flags &f) {
        int c = peek();
        if (c == '(') {
            pos_++;
            if (++depth_ > 250) throw ParseError("group nesting too deep");
            Flags inner = f;
            if (peek() == '?') {
                pos_++;
                if ((peek() == 'P' && peek(1) == '<') || (peek() == '<' && peek(1) != '=' && peek(1) != '!')) {
                    if (peek() == 'P') pos_++;
                    pos_++;
                    size_t start = pos_;
                    while (more() && peek() != '>') pos_++;
                    if (!more() || pos_ == start) throw ParseError("bad group name");
                    pos_++;
                } else if (peek() == '=' || peek() == '!' || peek() == '<') {
                    throw ParseError("look-around is not supported");   // nor by the regex crate
                } else {
                    Flags nf = f;
                    if (flag_group(nf)) {   // "(?flags)": changes the enclosing group's flags from here on
                        f = nf;
                        depth_--;
                        return mk(Node::EMPTY);
                    }
                    inner = nf;             // "(?flags:...)": for this group only
                }
            }
            NodeP n = alternation(inner);
            if (peek() != ')') throw ParseError("unclosed group");
            pos_++;
            depth_--;
            return n;
        }
        if (c == '[') {
            pos_++;
            return set_node(char_class(f));
        }
        if (c == '.') {
            pos_++;
            CharSet cs;
            cs.negate();
            if (!f.s) cs.bits[0] &= ~(1ull << '\n');
            return set_node(cs);
        }
        if (c == '^') {
            pos_++;
            return assert_node(f.m ? A_LINE_START : A_TEXT_START);
        }
        if (c == '$') {
            pos_++;
            return assert_node(f.m ? A_LINE_END : A_TEXT_END);
        }
        if (c == '\\') {
            pos_++;
            int e = peek();
            if (e == 'A') { pos_++; return assert_node(A_TEXT_START); }
            if (e == 'z') { pos_++; return assert_node(A_TEXT_END); }
            if (e == 'B') { pos_++; return assert_node(A_NOT_WORD); }
            if (e == '<') { pos_++; return assert_node(A_WORD_START); }
            if (e == '>') { pos_++; return assert_node(A_WORD_END); }
            if (e == 'b') {
                pos_++;
                if (peek() == '{') {
                    // \b{start} etc.; anything else after \b is an ordinary (invalid here) repetition of \b
                    static const struct { const char *lit; Assert a; } forms[] = {
                        {"{start}", A_WORD_START}, {"{end}", A_WORD_END}, {"{start-half}", A_WORD_START_HALF}, {"{end-half}", A_WORD_END_HALF}};
                    for (auto &fm : forms)
                        if (looking_at(fm.lit)) {
                            pos_ += strlen(fm.lit);
                            return assert_node(fm.a);
                        }
                }
                return assert_node(A_WORD);
            }
            CharSet cs;
            int lit = escape(cs, f);
            if (lit >= 0) add_scalar(cs, lit, f.i);
            else if (f.i) cs.fold_case();
            return set_node(cs);
        }
        if (c == '*' || c == '+' || c == '?') throw ParseError("repetition operator missing expression");
        if (c == '{') throw ParseError("repetition operator missing expression");
        CharSet cs;
        add_scalar(cs, take_scalar(), f.i);
        return set_node(cs);
    }

    NodeP repeat(Flags &f) {
        NodeP a = atom(f);
        for (;;) {
            skip_space(f);
            int c = peek();
            if (c == '*' || c == '+' || c == '?') {
                pos_++;
                a = mk(c == '*' ? Node::STAR : c == '+' ? Node::PLUS : Node::QUEST, a);
                if (peek() == '?') pos_++;   // lazy marker: irrelevant for is_match
            } else if (c == '{') {
                pos_++;
                int lo = integer(f);
                if (lo < 0) throw ParseError("invalid repetition");
                int hi = lo;
                if (peek() == ',') {
                    pos_++;
                    skip_space(f);
                    if (peek() == '}') hi = -1;
                    else {
                        hi = integer(f);
                        if (hi < 0) throw ParseError("invalid repetition");
                    }
                }
                if (peek() != '}' || (hi >= 0 && hi < lo)) throw ParseError("invalid repetition");
                pos_++;
                if (peek() == '?') pos_++;
                NodeP r = mk(Node::REPEAT, a);
                r->lo = lo;
                r->hi = hi;
                a = r;
            } else {
                break;
            }
        }
        return a;
    }

    NodeP concat(Flags &f) {
        NodeP res = mk(Node::EMPTY);
        for (;;) {
            skip_space(f);
            if (!more() || peek() == '|' || peek() == ')') break;
            res = mk(Node::CAT, res, repeat(f));
        }
        return res;
    }

    // `f` is the flag state of the enclosing group: a "(?i)" directive changes it for everything that follows in
    // that group, across '|' as well (regex-syntax keeps one flag state per group).
    NodeP alternation(Flags &f) {
        NodeP left = concat(f);
        while (peek() == '|') {
            pos_++;
            left = mk(Node::ALT, left, concat(f));
        }
        return left;
    }
};

// ---- NFA -------------------------------------------------------------------------------------------------

struct NState {
    enum Type { CHAR, SPLIT, ASSERT, MATCH } type;
    CharSet set;
    Assert as = A_TEXT_START;
    int a = -1, b = -1;
};

struct Nfa {
    std::vector<NState> st;
    int add(NState::Type t, int a = -1, int b = -1) {
        if (st.size() > 200000) throw ParseError("pattern too large");
        NState s;
        s.type = t;
        s.a = a;
        s.b = b;
        st.push_back(s);
        return (int)st.size() - 1;
    }
    // emits `n` so that it continues at `next`; returns the entry state
    int emit(const NodeP &n, int next) {
        switch (n->type) {
        case Node::EMPTY:
            return next;
        case Node::SET: {
            int s = add(NState::CHAR, next);
            st[s].set = n->set;
            return s;
        }
        case Node::CAT:
            return emit(n->a, emit(n->b, next));
        case Node::ALT: {
            int l = emit(n->a, next), r = emit(n->b, next);
            return add(NState::SPLIT, l, r);
        }
        case Node::STAR: {
            int s = add(NState::SPLIT, -1, next);
            int body = emit(n->a, s);
            st[s].a = body;
            return s;
        }
        case Node::PLUS: {
            int s = add(NState::SPLIT, -1, next);
            int body = emit(n->a, s);
            st[s].a = body;
            return body;
        }
        case Node::QUEST: {
            int body = emit(n->a, next);
            return add(NState::SPLIT, body, next);
        }
        case Node::ASSERT: {
            int s = add(NState::ASSERT, next);
            st[s].as = n->as;
            return s;
        }
        case Node::REPEAT: {
            int cur = next;
            if (n->hi < 0) {
                NodeP star = mk(Node::STAR, n->a);
                cur = emit(star, cur);
            } else {
                for (int i = n->lo; i < n->hi; i++) {
                    int body = emit(n->a, cur);
                    cur = add(NState::SPLIT, body, next);
                }
            }
            for (int i = 0; i < n->lo; i++) cur = emit(n->a, cur);
            return cur;
        }
        }
        return next;
    }
};

// Epsilon closure of `seeds` at a position whose previous symbol has kind `prev` and whose next symbol has kind
// `next` (K_UNKNOWN: not looked at yet — assertions that need it stay in the result as pending states).
// Result: sorted list of CHAR / MATCH / pending ASSERT states.
void closure(const Nfa &nfa, const std::vector<int> &seeds, Kind prev, Kind next, std::vector<int> &out,
             std::vector<uint32_t> &mark, uint32_t &epoch) {
    epoch++;
    out.clear();
    std::vector<int> stack(seeds.rbegin(), seeds.rend());
    while (!stack.empty()) {
        int s = stack.back();
        stack.pop_back();
        if (s < 0 || mark[s] == epoch) continue;
        mark[s] = epoch;
        const NState &n = nfa.st[s];
        switch (n.type) {
        case NState::SPLIT:
            stack.push_back(n.b);
            stack.push_back(n.a);
            break;
        case NState::ASSERT:
            if (next == K_UNKNOWN && needs_next(n.as)) out.push_back(s);
            else if (assert_holds(n.as, prev, next)) stack.push_back(n.a);
            break;
        default:
            out.push_back(s);
        }
    }
    std::sort(out.begin(), out.end());
}

struct TooManyStates {};

}  // namespace

// The pattern as its NFA, for patterns regex_compile does not determinise (regex_dfa.h).
struct LazyNfa {
    Nfa nfa;
    int start = -1, match = -1;

    // The walk regex_compile tabulates, done on the haystack itself: the set of NFA states after each byte.
    bool is_match(const char *text) const {
        std::vector<uint32_t> mark(nfa.st.size(), 0);
        uint32_t epoch = 0;
        std::vector<int> cur, resolved, seeds;
        closure(nfa, {start}, K_EDGE, K_UNKNOWN, cur, mark, epoch);
        Kind prev = K_EDGE;
        for (const unsigned char *p = (const unsigned char *)text; *p; p++) {
            if (std::binary_search(cur.begin(), cur.end(), match)) return true;
            const int c = *p < 128 ? *p : 128;
            const Kind nk = kind_of(c);
            closure(nfa, cur, prev, nk, resolved, mark, epoch);
            if (std::binary_search(resolved.begin(), resolved.end(), match)) return true;
            seeds.clear();
            for (int s : resolved) {
                const NState &n = nfa.st[s];
                if (n.type == NState::CHAR && n.set.has(c)) seeds.push_back(n.a);
            }
            seeds.push_back(start);   // unanchored search
            closure(nfa, seeds, nk, K_UNKNOWN, cur, mark, epoch);
            prev = nk;
        }
        if (std::binary_search(cur.begin(), cur.end(), match)) return true;
        closure(nfa, cur, prev, K_EDGE, resolved, mark, epoch);
        return std::binary_search(resolved.begin(), resolved.end(), match);
    }
};

bool Dfa::is_match(const char *text) const {
    if (lazy) return lazy->is_match(text);
    uint32_t s = 0;
    for (const unsigned char *p = (const unsigned char *)text; *p; p++) {
        if (match_now[s]) return true;
        if (dead[s]) return false;
        s = trans[(size_t)s * n_cls + cls[*p]];
    }
    return match_now[s] || match_at_end[s];
}

bool regex_compile(const std::string &pattern, bool case_insensitive, Dfa &out, std::string &err) {
    if (pattern.empty()) {
        err = "Pattern cannot be empty";
        return false;
    }
    try {
        Parser parser(pattern, case_insensitive);
        NodeP ast = parser.parse();
        Nfa nfa;
        int match = nfa.add(NState::MATCH);
        int start = nfa.emit(ast, match);
        bool any_assert = false;
        for (auto &s : nfa.st) any_assert = any_assert || s.type == NState::ASSERT;

        // symbol classes: bytes with identical membership in every CHAR set and of the same kind (newline / word /
        // other — the kinds only matter when the pattern has assertions)
        std::vector<const CharSet *> sets;
        for (auto &s : nfa.st)
            if (s.type == NState::CHAR) sets.push_back(&s.set);
        std::map<std::vector<bool>, uint8_t> sig2cls;
        std::vector<int> cls_rep;   // representative byte per class
        for (int c = 0; c <= 128; c++) {
            std::vector<bool> sig(sets.size() + 2);
            for (size_t i = 0; i < sets.size(); i++) sig[i] = sets[i]->has(c);
            const Kind k = kind_of(c);
            sig[sets.size()] = any_assert && k == K_NEWLINE;
            sig[sets.size() + 1] = any_assert && k == K_WORD;
            auto it = sig2cls.find(sig);
            uint8_t id;
            if (it == sig2cls.end()) {
                if (sig2cls.size() >= 255) throw ParseError("pattern needs too many symbol classes");
                id = (uint8_t)sig2cls.size();
                sig2cls[sig] = id;
                cls_rep.push_back(c);
            } else {
                id = it->second;
            }
            if (c < 128) out.cls[c] = id;
            else
                for (int b = 128; b < 256; b++) out.cls[b] = id;
        }
        const uint32_t n_cls = (uint32_t)cls_rep.size();

        // Subset construction.  A DFA state is (NFA states with their look-behind assertions already resolved and the
        // look-ahead ones pending, kind of the previous symbol); the kind is dropped (K_OTHER) when the pattern has no
        // assertion at all.  State 0 is the start of the haystack.
        std::vector<uint32_t> mark(nfa.st.size(), 0);
        uint32_t epoch = 0;
        std::map<std::pair<std::vector<int>, int>, uint32_t> ids;
        std::vector<std::pair<std::vector<int>, int>> states;
        std::vector<int> tmp, resolved;

        auto intern = [&](const std::vector<int> &set, Kind prev) -> uint32_t {
            auto key = std::make_pair(set, any_assert ? (int)prev : (int)K_OTHER);
            auto it = ids.find(key);
            if (it != ids.end()) return it->second;
            if (states.size() >= 20000) throw TooManyStates();
            uint32_t id = (uint32_t)states.size();
            ids[key] = id;
            states.push_back(key);
            return id;
        };

        out.lazy.reset();
        try {
        closure(nfa, {start}, K_EDGE, K_UNKNOWN, tmp, mark, epoch);
        {
            // the start state keeps K_EDGE even without assertions in the pattern: nothing depends on it then
            auto key = std::make_pair(tmp, (int)K_EDGE);
            ids[key] = 0;
            states.push_back(key);
        }
        std::vector<uint32_t> trans;
        std::vector<uint8_t> match_now, match_at_end;
        for (uint32_t si = 0; si < states.size(); si++) {
            const std::vector<int> cur = states[si].first;
            const Kind prev = si == 0 ? K_EDGE : (Kind)states[si].second;
            const bool now = std::binary_search(cur.begin(), cur.end(), match);
            // end of the haystack: the pending assertions see K_EDGE as the next symbol
            closure(nfa, cur, prev, K_EDGE, resolved, mark, epoch);
            match_now.push_back(now);
            match_at_end.push_back(std::binary_search(resolved.begin(), resolved.end(), match));
            trans.resize((size_t)(si + 1) * n_cls);
            for (uint32_t c = 0; c < n_cls; c++) {
                if (now) {
                    trans[(size_t)si * n_cls + c] = si;   // absorbing
                    continue;
                }
                const int rep = cls_rep[c];
                const Kind nk = kind_of(rep);
                // resolve what was waiting for this symbol, then step over it
                closure(nfa, cur, prev, nk, resolved, mark, epoch);
                if (std::binary_search(resolved.begin(), resolved.end(), match)) {
                    // matched just before this symbol: go to an absorbing match state
                    std::vector<int> m{match};
                    trans[(size_t)si * n_cls + c] = intern(m, K_OTHER);
                    continue;
                }
                std::vector<int> seeds;
                for (int s : resolved) {
                    const NState &n = nfa.st[s];
                    if (n.type == NState::CHAR && n.set.has(rep)) seeds.push_back(n.a);
                }
                seeds.push_back(start);   // unanchored search: a new attempt may begin at every offset
                closure(nfa, seeds, nk, K_UNKNOWN, tmp, mark, epoch);
                trans[(size_t)si * n_cls + c] = intern(tmp, nk);
            }
        }
        uint32_t n_states = (uint32_t)states.size();

        // Minimisation (Moore): states that accept the same continuations merge — among them the copies that differ
        // only in a previous-symbol kind no pending assertion looks at.  Block of state 0 stays number 0.
        std::vector<uint32_t> block(n_states);
        for (uint32_t s = 0; s < n_states; s++) block[s] = (match_now[s] ? 1u : 0u) | (match_at_end[s] ? 2u : 0u);
        uint32_t n_blocks = 0;
        for (;;) {
            std::map<std::vector<uint32_t>, uint32_t> sig_ids;
            std::vector<uint32_t> nb(n_states);
            std::vector<uint32_t> sig(n_cls + 1);
            for (uint32_t s = 0; s < n_states; s++) {
                sig[0] = block[s];
                for (uint32_t c = 0; c < n_cls; c++) sig[c + 1] = block[trans[(size_t)s * n_cls + c]];
                auto it = sig_ids.find(sig);
                if (it == sig_ids.end()) it = sig_ids.emplace(sig, (uint32_t)sig_ids.size()).first;
                nb[s] = it->second;
            }
            const bool stable = sig_ids.size() == n_blocks;
            n_blocks = (uint32_t)sig_ids.size();
            block = nb;
            if (stable) break;
        }
        // state 0 was the first to get a signature id in every round, so block[0] == 0
        out.n_cls = n_cls;
        out.n_states = n_blocks;
        out.trans.assign((size_t)n_blocks * n_cls, 0);
        out.match_now.assign(n_blocks, 0);
        out.match_at_end.assign(n_blocks, 0);
        for (uint32_t s = 0; s < n_states; s++) {
            const uint32_t b = block[s];
            out.match_now[b] = match_now[s];
            out.match_at_end[b] = match_at_end[s];
            for (uint32_t c = 0; c < n_cls; c++) out.trans[(size_t)b * n_cls + c] = block[trans[(size_t)s * n_cls + c]];
        }

        // dead states: cannot reach a state with match_now or match_at_end (reverse reachability)
        std::vector<uint8_t> live(out.n_states, 0);
        bool changed = true;
        for (uint32_t s = 0; s < out.n_states; s++) live[s] = out.match_now[s] || out.match_at_end[s];
        while (changed) {
            changed = false;
            for (uint32_t s = 0; s < out.n_states; s++) {
                if (live[s]) continue;
                for (uint32_t c = 0; c < out.n_cls; c++)
                    if (live[out.trans[(size_t)s * out.n_cls + c]]) {
                        live[s] = 1;
                        changed = true;
                        break;
                    }
            }
        }
        out.dead.assign(out.n_states, 0);
        for (uint32_t s = 0; s < out.n_states; s++) out.dead[s] = !live[s];
        return true;
        } catch (const TooManyStates &) {
            // no table for this one: keep the NFA and walk it per haystack (regex_dfa.h)
            auto lz = std::make_shared<LazyNfa>();
            lz->nfa = std::move(nfa);
            lz->start = start;
            lz->match = match;
            out.lazy = lz;
            out.n_cls = n_cls;
            out.n_states = 0;
            out.trans.clear();
            out.match_now.clear();
            out.match_at_end.clear();
            out.dead.clear();
            return true;
        }
    } catch (const ParseError &e) {
        err = "Invalid regex pattern: " + pattern + " (" + e.what() + ")";
        return false;
    }
}

}  // namespace vg
