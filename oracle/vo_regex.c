/*
 * vo_regex.c — Pattern::new / Pattern::matches for the parity oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Restates the filter semantics of the reference, src/pattern.rs:21-45:
 *   - empty pattern is an error (:22-24)
 *   - case-insensitive patterns are compiled as "(?i)" + pattern (:26-30)
 *   - matches() is regex::Regex::is_match, i.e. an UNANCHORED search over the whole address
 *     string (:43-45).
 * The regex crate (1.12.2, regex-syntax 0.8.8) itself is not on disk; this is an independent Thompson-NFA
 * simulation (zero-width assertions are evaluated against the text while threads are added) of the crate's
 * syntax as far as it can be meaningful on ASCII address strings:
 *   literals (non-ASCII ones never match), escapes (\n \t \r \f \v \a, \xHH \x{H..} \uHHHH \u{H..} \UHHHHHHHH,
 *   escaped punctuation), '.', bracket classes with ranges / negation / nesting / POSIX names / the set operators
 *   && -- ~~, \d \w \s, Unicode classes \pX \p{Name} \P{..} \p{^..} \p{gc=..} \p{sc=..} (ASCII members, from the
 *   general category of every ASCII character), groups (capturing, (?:..), named), alternation,
 *   * + ? {n} {n,} {n,m} (lazy forms accepted: laziness cannot change is_match), ^ $ \A \z,
 *   \b \B \< \> \b{start} \b{end} \b{start-half} \b{end-half}, and the inline flags i m s x U u
 *   ((?i), (?-i), (?i:..), combinations).
 * Not supported, rejected with an error rather than guessed at: the CRLF flag R, Unicode class names that are
 * not in the table below, look-around (the crate rejects it too).
 */
#include "vgen_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { N_EMPTY, N_SET, N_CAT, N_ALT, N_STAR, N_PLUS, N_QUEST, N_LOOK };

/* zero-width assertions */
enum { L_TEXT_START, L_TEXT_END, L_LINE_START, L_LINE_END, L_WORD, L_NOT_WORD, L_WORD_START, L_WORD_END,
       L_WORD_START_HALF, L_WORD_END_HALF };

typedef struct node {
    int type;
    int look;        /* N_LOOK */
    uint8_t set[16]; /* 128-bit ASCII membership for N_SET */
    struct node *a, *b;
} node;

typedef struct {
    int i, m, s, x;
} flags;

typedef struct {
    const char *p;
    char err[160];
    int failed;
    int nodes;
    int depth;
} parser;

static void set_add(uint8_t s[16], int c) { s[c >> 3] |= (uint8_t)(1u << (c & 7)); }
static int set_has(const uint8_t s[16], int c) { return c < 128 && (s[c >> 3] >> (c & 7)) & 1; }
static void set_range(uint8_t s[16], int lo, int hi) {
    for (int c = lo; c <= hi && c < 128; c++) set_add(s, c);
}

static node *mk(parser *ps, int type, node *a, node *b) {
    node *n = (node *)calloc(1, sizeof *n);
    n->type = type;
    n->a = a;
    n->b = b;
    if (++ps->nodes > 200000 && !ps->failed) {
        ps->failed = 1;
        snprintf(ps->err, sizeof ps->err, "pattern too large");
    }
    return n;
}

static void free_node(node *n) {
    if (!n) return;
    free_node(n->a);
    free_node(n->b);
    free(n);
}

static node *clone(parser *ps, const node *n) {
    if (!n) return NULL;
    node *c = mk(ps, n->type, clone(ps, n->a), clone(ps, n->b));
    memcpy(c->set, n->set, 16);
    c->look = n->look;
    return c;
}

static void fail(parser *ps, const char *msg) {
    if (!ps->failed) {
        ps->failed = 1;
        snprintf(ps->err, sizeof ps->err, "%s", msg);
    }
}

static void set_fold_case(uint8_t s[16]) {
    for (int c = 'a'; c <= 'z'; c++) {
        int u = c - 'a' + 'A';
        if (set_has(s, c) || set_has(s, u)) {
            set_add(s, c);
            set_add(s, u);
        }
    }
}

static void set_negate(uint8_t s[16]) {
    for (int i = 0; i < 16; i++) s[i] = (uint8_t)~s[i];
}

static void set_perl(uint8_t s[16], char kind) {
    uint8_t t[16] = {0};
    switch (kind | 0x20) {
    case 'd':
        set_range(t, '0', '9');
        break;
    case 'w':
        set_range(t, '0', '9');
        set_range(t, 'a', 'z');
        set_range(t, 'A', 'Z');
        set_add(t, '_');
        break;
    case 's': /* White_Space */
        set_add(t, ' ');
        set_add(t, '\t');
        set_add(t, '\n');
        set_add(t, '\r');
        set_add(t, '\f');
        set_add(t, '\v');
        break;
    }
    if (kind >= 'A' && kind <= 'Z') set_negate(t);
    for (int i = 0; i < 16; i++) s[i] |= t[i];
}

static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

/* ---- Unicode classes on ASCII: the general category of every ASCII character (UnicodeData.txt) -------------- */

static const char *ascii_gc(int c) {
    if (c < 0x20 || c == 0x7F) return "Cc";
    if (c == ' ') return "Zs";
    if (c >= '0' && c <= '9') return "Nd";
    if (c >= 'A' && c <= 'Z') return "Lu";
    if (c >= 'a' && c <= 'z') return "Ll";
    switch (c) {
    case '$': return "Sc";
    case '+': case '<': case '=': case '>': case '|': case '~': return "Sm";
    case '^': case '`': return "Sk";
    case '(': case '[': case '{': return "Ps";
    case ')': case ']': case '}': return "Pe";
    case '-': return "Pd";
    case '_': return "Pc";
    default: return "Po"; /* ! " # % & ' * , . / : ; ? @ \ */
    }
}

/* loose name matching (UAX44-LM3): case, '_', '-' and spaces do not count */
static void loose_name(const char *in, size_t n, char *out, size_t cap) {
    size_t o = 0;
    for (size_t i = 0; i < n && o + 1 < cap; i++) {
        char ch = in[i];
        if (ch == '_' || ch == '-' || ch == ' ') continue;
        out[o++] = (char)((ch >= 'A' && ch <= 'Z') ? ch + 32 : ch);
    }
    out[o] = 0;
}

static int name_in(const char *n, const char *const *list) {
    for (; *list; list++)
        if (!strcmp(n, *list)) return 1;
    return 0;
}

/* fills t with the ASCII members of the Unicode class `name` (already loosened); 0 = unknown name */
static int unicode_class_members(const char *n, uint8_t t[16]) {
    static const struct {
        const char *longname, *abbr;
    } cats[] = {{"uppercaseletter", "Lu"}, {"lowercaseletter", "Ll"}, {"decimalnumber", "Nd"}, {"connectorpunctuation", "Pc"},
                {"dashpunctuation", "Pd"}, {"openpunctuation", "Ps"}, {"closepunctuation", "Pe"}, {"otherpunctuation", "Po"},
                {"mathsymbol", "Sm"}, {"currencysymbol", "Sc"}, {"modifiersymbol", "Sk"}, {"spaceseparator", "Zs"}, {"control", "Cc"},
                {"digit", "Nd"}, {"cntrl", "Cc"}};
    /* two-letter categories and one-letter groups that occur in ASCII */
    static const char *const two[] = {"lu", "ll", "nd", "pc", "pd", "ps", "pe", "po", "sm", "sc", "sk", "zs", "cc", NULL};
    static const struct {
        const char *name;
        char group;
    } groups[] = {{"l", 'L'}, {"letter", 'L'}, {"n", 'N'}, {"number", 'N'}, {"p", 'P'}, {"punctuation", 'P'}, {"punct", 'P'},
                  {"s", 'S'}, {"symbol", 'S'}, {"z", 'Z'}, {"separator", 'Z'}, {"c", 'C'}, {"other", 'C'}};
    /* categories, scripts and properties without any ASCII member */
    static const char *const empty[] = {"lt", "titlecaseletter", "lm", "modifierletter", "lo", "otherletter", "m", "mark", "mn", "nonspacingmark",
        "mc", "spacingmark", "me", "enclosingmark", "nl", "letternumber", "no", "othernumber", "pi", "initialpunctuation", "pf",
        "finalpunctuation", "so", "othersymbol", "zl", "lineseparator", "zp", "paragraphseparator", "cf", "format", "cs", "surrogate",
        "co", "privateuse", "cn", "unassigned", "greek", "grek", "cyrillic", "cyrl", "han", "hani", "arabic", "arab", "hebrew", "hebr",
        "hiragana", "hira", "katakana", "kana", "hangul", "hang", "thai", "devanagari", "deva", "armenian", "armn", "georgian", "geor",
        "inherited", "zinh", "unknown", "zzzz",
        /* the other scripts of Unicode 16 (long names): none has an ASCII member */
        "adlam", "ahom", "anatolianhieroglyphs", "avestan", "balinese", "bamum", "bassavah", "batak", "bengali",
        "bhaiksuki", "bopomofo", "brahmi", "braille", "buginese", "buhid", "canadianaboriginal", "carian",
        "caucasianalbanian", "chakma", "cham", "cherokee", "chorasmian", "coptic", "cuneiform", "cypriot",
        "cyprominoan", "deseret", "divesakuru", "dogra", "duployan", "egyptianhieroglyphs", "elbasan", "elymaic",
        "ethiopic", "garay", "glagolitic", "gothic", "grantha", "gujarati", "gunjalagondi", "gurmukhi",
        "gurungkhema", "hanifirohingya", "hanunoo", "hatran", "imperialaramaic", "inscriptionalpahlavi",
        "inscriptionalparthian", "javanese", "kaithi", "kannada", "kawi", "kayahli", "kharoshthi",
        "khitansmallscript", "khmer", "khojki", "khudawadi", "kiratrai", "lao", "lepcha", "limbu", "lineara",
        "linearb", "lisu", "lycian", "lydian", "mahajani", "makasar", "malayalam", "mandaic", "manichaean",
        "marchen", "masaramgondi", "medefaidrin", "meeteimayek", "mendekikakui", "meroiticcursive",
        "meroitichieroglyphs", "miao", "modi", "mongolian", "mro", "multani", "myanmar", "nabataean", "nagmundari",
        "nandinagari", "newa", "newtailue", "nko", "nushu", "nyiakengpuachuehmong", "ogham", "olchiki",
        "oldhungarian", "olditalic", "oldnortharabian", "oldpermic", "oldpersian", "oldsogdian", "oldsoutharabian",
        "oldturkic", "olduyghur", "olonal", "oriya", "osage", "osmanya", "pahawhhmong", "palmyrene", "paucinhau",
        "phagspa", "phoenician", "psalterpahlavi", "rejang", "runic", "samaritan", "saurashtra", "sharada",
        "shavian", "siddham", "signwriting", "sinhala", "sogdian", "sorasompeng", "soyombo", "sundanese", "sunuwar",
        "sylotinagri", "syriac", "tagalog", "tagbanwa", "taile", "taitham", "taiviet", "takri", "tamil", "tangsa",
        "tangut", "telugu", "thaana", "tibetan", "tifinagh", "tirhuta", "todhri", "toto", "tulutigalari",
        "ugaritic", "vai", "vithkuqi", "wancho", "warangciti", "yezidi", "yi", "zanabazarsquare", NULL};
    memset(t, 0, 16);
    if (name_in(n, empty)) return 1;
    if (!strcmp(n, "any") || !strcmp(n, "ascii") || !strcmp(n, "assigned")) {
        memset(t, 0xFF, 16);
        return 1;
    }
    const char *want = NULL;
    for (size_t i = 0; i < sizeof cats / sizeof cats[0]; i++)
        if (!strcmp(n, cats[i].longname)) want = cats[i].abbr;
    if (!want && name_in(n, two)) want = n;
    if (want) {
        for (int c = 0; c < 128; c++) {
            const char *g = ascii_gc(c);
            if ((g[0] | 0x20) == (want[0] | 0x20) && (g[1] | 0x20) == (want[1] | 0x20)) set_add(t, c);
        }
        return 1;
    }
    for (size_t i = 0; i < sizeof groups / sizeof groups[0]; i++)
        if (!strcmp(n, groups[i].name)) {
            for (int c = 0; c < 128; c++)
                if (ascii_gc(c)[0] == groups[i].group) set_add(t, c);
            return 1;
        }
    if (!strcmp(n, "lc") || !strcmp(n, "casedletter") || !strcmp(n, "alphabetic") || !strcmp(n, "alpha") || !strcmp(n, "cased") ||
        !strcmp(n, "latin") || !strcmp(n, "latn") || !strcmp(n, "idstart") || !strcmp(n, "xidstart")) {
        for (int c = 0; c < 128; c++)
            if (ascii_gc(c)[0] == 'L') set_add(t, c);
        return 1;
    }
    if (!strcmp(n, "uppercase") || !strcmp(n, "upper")) return unicode_class_members("lu", t);
    if (!strcmp(n, "lowercase") || !strcmp(n, "lower")) return unicode_class_members("ll", t);
    if (!strcmp(n, "common") || !strcmp(n, "zyyy")) {
        for (int c = 0; c < 128; c++)
            if (ascii_gc(c)[0] != 'L') set_add(t, c);
        return 1;
    }
    if (!strcmp(n, "whitespace") || !strcmp(n, "space") || !strcmp(n, "wspace") || !strcmp(n, "patternwhitespace")) {
        set_perl(t, 's');
        return 1;
    }
    if (!strcmp(n, "hexdigit") || !strcmp(n, "hex") || !strcmp(n, "asciihexdigit") || !strcmp(n, "ahex")) {
        for (int c = 0; c < 128; c++)
            if (hexval(c) >= 0) set_add(t, c);
        return 1;
    }
    if (!strcmp(n, "idcontinue") || !strcmp(n, "xidcontinue")) {
        set_perl(t, 'w');
        return 1;
    }
    /* a few binary properties, by their ASCII members (DerivedCoreProperties.txt / PropList.txt / emoji-data.txt) */
    static const struct {
        const char *name, *chars;
    } props[] = {{"caseignorable", "'.:^`"}, {"emoji", "#*0123456789"}, {"math", "+<=>^|~"}, {"dash", "-"}, {"quotationmark", "\"'"},
                 {"terminalpunctuation", "!,.:;?"}, {"patternsyntax", "!\"#$%&'()*+,-./:;<=>?@[\\]^`{|}~"}};
    for (size_t i = 0; i < sizeof props / sizeof props[0]; i++)
        if (!strcmp(n, props[i].name)) {
            for (const char *q = props[i].chars; *q; q++) set_add(t, (unsigned char)*q);
            return 1;
        }
    return 0;
}

/* after "\p" / "\P": merges the class into s; 0 on error */
static int parse_unicode_class(parser *ps, uint8_t s[16], int negated) {
    char raw[96], name[96];
    size_t n = 0;
    if (*ps->p == '{') {
        ps->p++;
        while (*ps->p && *ps->p != '}' && n + 1 < sizeof raw) raw[n++] = *ps->p++;
        if (*ps->p != '}') {
            fail(ps, "unclosed Unicode class");
            return 0;
        }
        ps->p++;
    } else if (*ps->p) {
        raw[n++] = *ps->p++;
    } else {
        fail(ps, "incomplete Unicode class");
        return 0;
    }
    raw[n] = 0;
    const char *v = raw;
    if (*v == '^') {
        negated = !negated;
        v++;
    }
    const char *eq = strpbrk(v, "=:");
    if (eq) {
        char key[96];
        size_t kl = (size_t)(eq - v);
        int ne = kl > 0 && v[kl - 1] == '!';
        loose_name(v, ne ? kl - 1 : kl, key, sizeof key);
        static const char *const keys[] = {"gc", "generalcategory", "sc", "script", "scx", "scriptextensions", NULL};
        if (!name_in(key, keys)) {
            fail(ps, "unsupported Unicode property");
            return 0;
        }
        if (ne) negated = !negated;
        v = eq + 1;
    }
    loose_name(v, strlen(v), name, sizeof name);
    uint8_t t[16];
    if (!unicode_class_members(name, t)) {
        fail(ps, "unsupported Unicode class name");
        return 0;
    }
    if (negated) set_negate(t);
    for (int i = 0; i < 16; i++) s[i] |= t[i];
    return 1;
}

/* ---- parser -------------------------------------------------------------------------------------------- */

static void skip_space(parser *ps, const flags *f) {
    if (!f->x) return;
    for (;;) {
        char c = *ps->p;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') ps->p++;
        else if (c == '#')
            while (*ps->p && *ps->p != '\n') ps->p++;
        else
            break;
    }
}

/* one scalar value of the UTF-8 pattern; -1 on malformed input */
static int take_scalar(parser *ps) {
    int c = (unsigned char)*ps->p++;
    if (c < 0x80) return c;
    int extra = c >= 0xF0 ? 3 : c >= 0xE0 ? 2 : c >= 0xC0 ? 1 : -1;
    if (extra < 0) {
        fail(ps, "pattern is not valid UTF-8");
        return -1;
    }
    int cp = c & (0x3F >> extra);
    while (extra--) {
        if (((unsigned char)*ps->p & 0xC0) != 0x80) {
            fail(ps, "pattern is not valid UTF-8");
            return -1;
        }
        cp = (cp << 6) | ((unsigned char)*ps->p++ & 0x3F);
    }
    return cp;
}

/* adds a scalar value to a set: ASCII as itself; of the non-ASCII ones only the two whose simple case folding
 * reaches ASCII matter (U+017F long s, U+212A Kelvin sign), and only case-insensitively */
static void set_add_scalar(uint8_t s[16], int cp, int ci) {
    if (cp < 128) set_add(s, cp);
    else if (ci && cp == 0x17F) set_add(s, 's');
    else if (ci && cp == 0x212A) set_add(s, 'k');
    if (ci) set_fold_case(s);
}

static int parse_hex(parser *ps, int digits) {
    long v = 0;
    if (*ps->p == '{') {
        ps->p++;
        int n = 0;
        while (*ps->p && *ps->p != '}') {
            int h = hexval((unsigned char)*ps->p++);
            if (h < 0 || ++n > 8) {
                fail(ps, "bad hexadecimal escape");
                return -1;
            }
            v = v * 16 + h;
        }
        if (*ps->p != '}' || n == 0) {
            fail(ps, "bad hexadecimal escape");
            return -1;
        }
        ps->p++;
    } else {
        for (int k = 0; k < digits; k++) {
            int h = hexval((unsigned char)*ps->p);
            if (h < 0) {
                fail(ps, "bad hexadecimal escape");
                return -1;
            }
            ps->p++;
            v = v * 16 + h;
        }
    }
    if (v > 0x10FFFF || (v >= 0xD800 && v <= 0xDFFF)) {
        fail(ps, "escape is not a Unicode scalar value");
        return -1;
    }
    return (int)v;
}

/* parses one escape after the backslash; returns a scalar value, or -2 after adding a class
 * to `s`, or -1 on error */
static int parse_escape(parser *ps, uint8_t s[16], const flags *f) {
    if (!*ps->p) {
        fail(ps, "trailing backslash");
        return -1;
    }
    int c = take_scalar(ps);
    if (c < 0) return -1;
    switch (c) {
    case 'd': case 'D': case 'w': case 'W': case 's': case 'S':
        set_perl(s, (char)c);
        return -2;
    case 'p': case 'P':
        return parse_unicode_class(ps, s, c == 'P') ? -2 : -1;
    case 'n': return '\n';
    case 't': return '\t';
    case 'r': return '\r';
    case 'f': return '\f';
    case 'v': return '\v';
    case 'a': return 7;
    case 'x': return parse_hex(ps, 2);
    case 'u': return parse_hex(ps, 4);
    case 'U': return parse_hex(ps, 8);
    case ' ':
        if (f->x) return ' ';
        fail(ps, "unrecognized escape sequence");
        return -1;
    default:
        if (c >= 128 || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) {
            fail(ps, "unrecognized escape sequence");
            return -1;
        }
        return c; /* escaped punctuation */
    }
}

static node *parse_alt(parser *ps, flags *f);

/* "[:name:]" right after the '[' of a class item?  1 = merged, 0 = not a POSIX class */
static int parse_posix(parser *ps, uint8_t s[16]) {
    if (*ps->p != ':') return 0;
    const char *end = strstr(ps->p + 1, ":]");
    if (!end) return 0;
    char name[16];
    const char *b = ps->p + 1;
    int neg = *b == '^';
    if (neg) b++;
    size_t n = (size_t)(end - b);
    if (n == 0 || n >= sizeof name) return 0;
    memcpy(name, b, n);
    name[n] = 0;
    uint8_t t[16] = {0};
    int (*pred)(int) = NULL;
    (void)pred;
    int known = 1;
    for (int c = 0; c < 128; c++) {
        int up = c >= 'A' && c <= 'Z', lo = c >= 'a' && c <= 'z', dg = c >= '0' && c <= '9';
        int in;
        if (!strcmp(name, "alnum")) in = up || lo || dg;
        else if (!strcmp(name, "alpha")) in = up || lo;
        else if (!strcmp(name, "ascii")) in = 1;
        else if (!strcmp(name, "blank")) in = c == ' ' || c == '\t';
        else if (!strcmp(name, "cntrl")) in = c < 32 || c == 127;
        else if (!strcmp(name, "digit")) in = dg;
        else if (!strcmp(name, "graph")) in = c > 32 && c < 127;
        else if (!strcmp(name, "lower")) in = lo;
        else if (!strcmp(name, "print")) in = c >= 32 && c < 127;
        else if (!strcmp(name, "punct")) in = c > 32 && c < 127 && !(up || lo || dg);
        else if (!strcmp(name, "space")) in = c == ' ' || (c >= 9 && c <= 13);
        else if (!strcmp(name, "upper")) in = up;
        else if (!strcmp(name, "word")) in = up || lo || dg || c == '_';
        else if (!strcmp(name, "xdigit")) in = hexval(c) >= 0;
        else {
            known = 0;
            break;
        }
        if (in) set_add(t, c);
    }
    if (!known) return 0;
    if (neg) set_negate(t);
    for (int i = 0; i < 16; i++) s[i] |= t[i];
    ps->p = end + 2;
    return 1;
}

/* bracketed class after '['.  Result in out.  Precedence: ranges, union, then && -- ~~ left to right, then
 * negation; both operands of a set operator and every bracketed class are case-folded when i is on. */
static int parse_class_set(parser *ps, const flags *f, uint8_t out[16]) {
    if (++ps->depth > 64) {
        fail(ps, "class nesting too deep");
        return 0;
    }
    uint8_t acc[16] = {0}, cur[16] = {0};
    int have_acc = 0, op = 0, neg = 0, first = 1;
    skip_space(ps, f);
    if (*ps->p == '^') {
        neg = 1;
        ps->p++;
    }
    for (;;) {
        skip_space(ps, f);
        int c = (unsigned char)*ps->p;
        if (!c) {
            fail(ps, "unclosed character class");
            return 0;
        }
        int is_op = !first && ((c == '&' && ps->p[1] == '&') || (c == '-' && ps->p[1] == '-') || (c == '~' && ps->p[1] == '~'));
        if ((c == ']' && !first) || is_op) {
            if (f->i) set_fold_case(cur);
            if (!have_acc) {
                memcpy(acc, cur, 16);
                have_acc = 1;
            } else {
                if (f->i) set_fold_case(acc);
                for (int k = 0; k < 16; k++)
                    acc[k] = (uint8_t)(op == '&' ? acc[k] & cur[k] : op == '-' ? acc[k] & ~cur[k] : acc[k] ^ cur[k]);
            }
            memset(cur, 0, 16);
            if (!is_op) {
                ps->p++;
                break;
            }
            op = c;
            ps->p += 2;
            continue;
        }
        first = 0;
        if (c == '[') {
            ps->p++;
            if (parse_posix(ps, cur)) continue;
            uint8_t inner[16];
            if (!parse_class_set(ps, f, inner)) return 0;
            for (int k = 0; k < 16; k++) cur[k] |= inner[k];
            continue;
        }
        int lo;
        if (c == '\\') {
            ps->p++;
            lo = parse_escape(ps, cur, f);
            if (lo == -1) return 0;
            if (lo == -2) continue;
        } else {
            lo = take_scalar(ps);
            if (lo < 0) return 0;
        }
        int hi = lo;
        skip_space(ps, f);
        if (ps->p[0] == '-' && ps->p[1] && ps->p[1] != ']' && ps->p[1] != '-') {
            ps->p++;
            skip_space(ps, f);
            if (*ps->p == '\\') {
                ps->p++;
                uint8_t dummy[16] = {0};
                hi = parse_escape(ps, dummy, f);
                if (hi < 0) {
                    fail(ps, "invalid class range");
                    return 0;
                }
            } else if (*ps->p == '[') {
                fail(ps, "invalid class range");
                return 0;
            } else {
                hi = take_scalar(ps);
                if (hi < 0) return 0;
            }
            if (hi < lo) {
                fail(ps, "invalid class range");
                return 0;
            }
        }
        if (lo < 128) set_range(cur, lo, hi);
        if (f->i && lo <= 0x17F && hi >= 0x17F) set_add(cur, 's');
        if (f->i && lo <= 0x212A && hi >= 0x212A) set_add(cur, 'k');
    }
    if (f->i) set_fold_case(acc);
    if (neg) set_negate(acc);
    memcpy(out, acc, 16);
    ps->depth--;
    return 1;
}

static int parse_int(parser *ps, const flags *f) {
    skip_space(ps, f);
    if (*ps->p < '0' || *ps->p > '9') return -1;
    long v = 0;
    while (*ps->p >= '0' && *ps->p <= '9') {
        v = v * 10 + (*ps->p - '0');
        if (v > 1000) return -2;
        ps->p++;
    }
    skip_space(ps, f);
    return (int)v;
}

static node *look(parser *ps, int kind) {
    node *n = mk(ps, N_LOOK, NULL, NULL);
    n->look = kind;
    return n;
}

/* "(?" has been consumed and what follows is a flag list.  Returns 1 = "(?flags)" directive (flags of the
 * enclosing group updated in *f), 2 = "(?flags:" (scoped flags written to *scoped), 0 = error */
static int parse_flags(parser *ps, flags *f, flags *scoped) {
    flags nf = *f;
    int on = 1, any = 0, negated_any = 0;
    while (*ps->p && *ps->p != ':' && *ps->p != ')') {
        char c = *ps->p++;
        if (c == '-') {
            if (!on) {
                fail(ps, "repeated negation in flag group");
                return 0;
            }
            on = 0;
            continue;
        }
        if (c == 'i') nf.i = on;
        else if (c == 'm') nf.m = on;
        else if (c == 's') nf.s = on;
        else if (c == 'x') nf.x = on;
        else if (c == 'U' || c == 'u') { /* swap-greed / Unicode: no effect on is_match over ASCII text */ }
        else {
            fail(ps, c == 'R' ? "CRLF mode (flag R) unsupported" : "unrecognized flag");
            return 0;
        }
        any = 1;
        if (!on) negated_any = 1;
    }
    if (!*ps->p) {
        fail(ps, "unclosed group");
        return 0;
    }
    if (!on && !negated_any) {
        fail(ps, "dangling flag negation");
        return 0;
    }
    if (*ps->p == ')') {
        if (!any) {
            fail(ps, "empty flag group");
            return 0;
        }
        ps->p++;
        *f = nf;
        return 1;
    }
    ps->p++;
    *scoped = nf;
    return 2;
}

static node *set_atom(parser *ps, const uint8_t s[16]) {
    node *n = mk(ps, N_SET, NULL, NULL);
    memcpy(n->set, s, 16);
    return n;
}

static node *parse_atom(parser *ps, flags *f) {
    int c = (unsigned char)*ps->p;
    if (c == '(') {
        ps->p++;
        if (++ps->depth > 250) {
            fail(ps, "group nesting too deep");
            return NULL;
        }
        flags inner = *f;
        if (*ps->p == '?') {
            ps->p++;
            int named = (*ps->p == 'P' && ps->p[1] == '<') || (*ps->p == '<' && ps->p[1] != '=' && ps->p[1] != '!');
            if (named) {
                ps->p += *ps->p == 'P' ? 2 : 1;
                const char *b = ps->p;
                while (*ps->p && *ps->p != '>') ps->p++;
                if (!*ps->p || ps->p == b) {
                    fail(ps, "bad group name");
                    return NULL;
                }
                ps->p++;
            } else if (*ps->p == '=' || *ps->p == '!' || *ps->p == '<') {
                fail(ps, "look-around is not supported");
                return NULL;
            } else {
                int r = parse_flags(ps, f, &inner);
                if (r == 0) return NULL;
                if (r == 1) {
                    ps->depth--;
                    return mk(ps, N_EMPTY, NULL, NULL);
                }
            }
        }
        node *body = parse_alt(ps, &inner);
        if (ps->failed) {
            free_node(body);
            return NULL;
        }
        if (*ps->p != ')') {
            fail(ps, "unclosed group");
            free_node(body);
            return NULL;
        }
        ps->p++;
        ps->depth--;
        return body;
    }
    if (c == '[') {
        ps->p++;
        uint8_t s[16];
        if (!parse_class_set(ps, f, s)) return NULL;
        return set_atom(ps, s);
    }
    if (c == '.') {
        ps->p++;
        uint8_t s[16];
        memset(s, 0xFF, 16);
        if (!f->s) s['\n' >> 3] &= (uint8_t)~(1u << ('\n' & 7));
        return set_atom(ps, s);
    }
    if (c == '^') {
        ps->p++;
        return look(ps, f->m ? L_LINE_START : L_TEXT_START);
    }
    if (c == '$') {
        ps->p++;
        return look(ps, f->m ? L_LINE_END : L_TEXT_END);
    }
    if (c == '\\') {
        ps->p++;
        switch (*ps->p) {
        case 'A': ps->p++; return look(ps, L_TEXT_START);
        case 'z': ps->p++; return look(ps, L_TEXT_END);
        case 'B': ps->p++; return look(ps, L_NOT_WORD);
        case '<': ps->p++; return look(ps, L_WORD_START);
        case '>': ps->p++; return look(ps, L_WORD_END);
        case 'b':
            ps->p++;
            if (!strncmp(ps->p, "{start}", 7)) { ps->p += 7; return look(ps, L_WORD_START); }
            if (!strncmp(ps->p, "{end}", 5)) { ps->p += 5; return look(ps, L_WORD_END); }
            if (!strncmp(ps->p, "{start-half}", 12)) { ps->p += 12; return look(ps, L_WORD_START_HALF); }
            if (!strncmp(ps->p, "{end-half}", 10)) { ps->p += 10; return look(ps, L_WORD_END_HALF); }
            return look(ps, L_WORD);
        default: break;
        }
        uint8_t s[16] = {0};
        int lit = parse_escape(ps, s, f);
        if (lit == -1) return NULL;
        if (lit >= 0) set_add_scalar(s, lit, f->i);
        else if (f->i) set_fold_case(s);
        return set_atom(ps, s);
    }
    if (c == '*' || c == '+' || c == '?' || c == '{') {
        fail(ps, "repetition operator missing expression");
        return NULL;
    }
    int cp = take_scalar(ps);
    if (cp < 0) return NULL;
    uint8_t s[16] = {0};
    set_add_scalar(s, cp, f->i);
    return set_atom(ps, s);
}

static node *repeat_node(parser *ps, node *atom, int lo, int hi) {
    /* hi < 0: unbounded */
    node *res = mk(ps, N_EMPTY, NULL, NULL);
    for (int i = 0; i < lo; i++) res = mk(ps, N_CAT, res, clone(ps, atom));
    if (hi < 0) {
        res = mk(ps, N_CAT, res, mk(ps, N_STAR, clone(ps, atom), NULL));
    } else {
        /* (a(a(a)?)?)? nested optionals */
        node *opt = NULL;
        for (int i = lo; i < hi; i++) {
            node *inner = clone(ps, atom);
            if (opt) inner = mk(ps, N_CAT, inner, opt);
            opt = mk(ps, N_QUEST, inner, NULL);
        }
        if (opt) res = mk(ps, N_CAT, res, opt);
    }
    free_node(atom);
    return res;
}

static node *parse_repeat(parser *ps, flags *f) {
    node *atom = parse_atom(ps, f);
    if (!atom || ps->failed) {
        free_node(atom);
        return NULL;
    }
    for (;;) {
        skip_space(ps, f);
        int c = *ps->p;
        if (c == '*' || c == '+' || c == '?') {
            ps->p++;
            atom = mk(ps, c == '*' ? N_STAR : c == '+' ? N_PLUS : N_QUEST, atom, NULL);
            if (*ps->p == '?') ps->p++; /* lazy marker: irrelevant for is_match */
        } else if (c == '{') {
            ps->p++;
            int lo = parse_int(ps, f), hi;
            if (lo == -2) {
                fail(ps, "repetition count too large");
                free_node(atom);
                return NULL;
            }
            if (lo < 0) {
                /* regex crate: a '{' that does not start a counted repetition is an error */
                fail(ps, "invalid repetition");
                free_node(atom);
                return NULL;
            }
            hi = lo;
            if (*ps->p == ',') {
                ps->p++;
                skip_space(ps, f);
                if (*ps->p == '}')
                    hi = -1;
                else {
                    hi = parse_int(ps, f);
                    if (hi < 0) {
                        fail(ps, hi == -2 ? "repetition count too large" : "invalid repetition");
                        free_node(atom);
                        return NULL;
                    }
                }
            }
            if (*ps->p != '}' || (hi >= 0 && hi < lo)) {
                fail(ps, "invalid repetition");
                free_node(atom);
                return NULL;
            }
            ps->p++;
            if (*ps->p == '?') ps->p++;
            atom = repeat_node(ps, atom, lo, hi);
        } else
            break;
        if (ps->failed) {
            free_node(atom);
            return NULL;
        }
    }
    return atom;
}

static node *parse_cat(parser *ps, flags *f) {
    node *res = mk(ps, N_EMPTY, NULL, NULL);
    for (;;) {
        skip_space(ps, f);
        if (!*ps->p || *ps->p == '|' || *ps->p == ')') break;
        node *r = parse_repeat(ps, f);
        if (!r || ps->failed) {
            free_node(r);
            free_node(res);
            return NULL;
        }
        res = mk(ps, N_CAT, res, r);
    }
    return res;
}

/* *f is the flag state of the group being parsed: an inline "(?i)" inside one branch keeps applying to the later
 * branches of the same group (the crate scopes flags to the group, not the branch) */
static node *parse_alt(parser *ps, flags *f) {
    node *left = parse_cat(ps, f);
    if (!left) return NULL;
    while (*ps->p == '|') {
        ps->p++;
        node *right = parse_cat(ps, f);
        if (!right) {
            free_node(left);
            return NULL;
        }
        left = mk(ps, N_ALT, left, right);
    }
    return left;
}

/* ---- NFA program --------------------------------------------------------------------------- */

enum { I_SET, I_SPLIT, I_JMP, I_LOOK, I_MATCH };

typedef struct {
    int op, x, y;
    uint8_t set[16];
} inst;

struct vo_regex {
    inst *prog;
    int n, cap;
    /* scratch for the simulation is allocated per call (thread-safe) */
};

static int emit(vo_regex *re, int op) {
    if (re->n == re->cap) {
        re->cap = re->cap ? re->cap * 2 : 64;
        re->prog = (inst *)realloc(re->prog, (size_t)re->cap * sizeof(inst));
    }
    memset(&re->prog[re->n], 0, sizeof(inst));
    re->prog[re->n].op = op;
    return re->n++;
}

static void compile(vo_regex *re, const node *n) {
    switch (n->type) {
    case N_EMPTY:
        break;
    case N_SET: {
        int i = emit(re, I_SET);
        memcpy(re->prog[i].set, n->set, 16);
        break;
    }
    case N_LOOK: {
        int i = emit(re, I_LOOK);
        re->prog[i].x = n->look;
        break;
    }
    case N_CAT:
        compile(re, n->a);
        compile(re, n->b);
        break;
    case N_ALT: {
        int s = emit(re, I_SPLIT);
        re->prog[s].x = re->n;
        compile(re, n->a);
        int j = emit(re, I_JMP);
        re->prog[s].y = re->n;
        compile(re, n->b);
        re->prog[j].x = re->n;
        break;
    }
    case N_STAR: {
        int s = emit(re, I_SPLIT);
        re->prog[s].x = re->n;
        compile(re, n->a);
        int j = emit(re, I_JMP);
        re->prog[j].x = s;
        re->prog[s].y = re->n;
        break;
    }
    case N_PLUS: {
        int start = re->n;
        compile(re, n->a);
        int s = emit(re, I_SPLIT);
        re->prog[s].x = start;
        re->prog[s].y = re->n;
        break;
    }
    case N_QUEST: {
        int s = emit(re, I_SPLIT);
        re->prog[s].x = re->n;
        compile(re, n->a);
        re->prog[s].y = re->n;
        break;
    }
    }
}

vo_regex *vo_regex_new(const char *pattern, int case_insensitive, char *err, size_t errcap) {
    if (err && errcap) err[0] = 0;
    if (!pattern || !*pattern) {
        if (err) snprintf(err, errcap, "Pattern cannot be empty");
        return NULL;
    }
    parser ps;
    memset(&ps, 0, sizeof ps);
    ps.p = pattern;
    flags top = {case_insensitive ? 1 : 0, 0, 0, 0};
    node *ast = parse_alt(&ps, &top);
    if (!ps.failed && *ps.p == ')') fail(&ps, "unopened group");
    if (ps.failed || !ast) {
        if (err) snprintf(err, errcap, "Invalid regex pattern: %s (%s)", pattern, ps.err);
        free_node(ast);
        return NULL;
    }
    vo_regex *re = (vo_regex *)calloc(1, sizeof *re);
    compile(re, ast);
    emit(re, I_MATCH);
    free_node(ast);
    if (re->n > 100000) {
        if (err) snprintf(err, errcap, "Invalid regex pattern: %s (program too large)", pattern);
        vo_regex_free(re);
        return NULL;
    }
    return re;
}

void vo_regex_free(vo_regex *re) {
    if (!re) return;
    free(re->prog);
    free(re);
}

typedef struct {
    int *list;
    int n;
    uint8_t *mark;
} tset;

static int word_char(int c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }

/* does assertion `kind` hold between text[pos-1] and text[pos]? */
static int look_holds(int kind, const char *text, size_t pos, size_t len) {
    int pw = pos > 0 && word_char((unsigned char)text[pos - 1]);
    int nw = pos < len && word_char((unsigned char)text[pos]);
    switch (kind) {
    case L_TEXT_START: return pos == 0;
    case L_TEXT_END: return pos == len;
    case L_LINE_START: return pos == 0 || text[pos - 1] == '\n';
    case L_LINE_END: return pos == len || text[pos] == '\n';
    case L_WORD: return pw != nw;
    case L_NOT_WORD: return pw == nw;
    case L_WORD_START: return !pw && nw;
    case L_WORD_END: return pw && !nw;
    case L_WORD_START_HALF: return !pw;
    case L_WORD_END_HALF: return !nw;
    }
    return 0;
}

/* follow epsilon edges from pc at haystack position pos (of len); returns 1 if MATCH reached */
static int add_thread(const vo_regex *re, tset *t, int pc, const char *text, size_t pos, size_t len) {
    /* explicit stack: programs can be large */
    int sp = 0, cap = 64;
    int *stack = (int *)malloc(sizeof(int) * (size_t)cap);
    int matched = 0;
    stack[sp++] = pc;
    while (sp) {
        int q = stack[--sp];
        if (t->mark[q]) continue;
        t->mark[q] = 1;
        const inst *in = &re->prog[q];
        int push1 = -1, push2 = -1;
        switch (in->op) {
        case I_JMP: push1 = in->x; break;
        case I_SPLIT: push1 = in->y; push2 = in->x; break;
        case I_LOOK: if (look_holds(in->x, text, pos, len)) push1 = q + 1; break;
        case I_MATCH: matched = 1; break;
        default: t->list[t->n++] = q; break;
        }
        if (sp + 2 > cap) {
            cap *= 2;
            stack = (int *)realloc(stack, sizeof(int) * (size_t)cap);
        }
        if (push1 >= 0) stack[sp++] = push1;
        if (push2 >= 0) stack[sp++] = push2;
    }
    free(stack);
    return matched;
}

int vo_regex_is_match(const vo_regex *re, const char *text) {
    size_t len = strlen(text);
    tset cur, nxt;
    cur.list = (int *)malloc(sizeof(int) * (size_t)re->n);
    nxt.list = (int *)malloc(sizeof(int) * (size_t)re->n);
    cur.mark = (uint8_t *)calloc((size_t)re->n, 1);
    nxt.mark = (uint8_t *)calloc((size_t)re->n, 1);
    cur.n = nxt.n = 0;
    int matched = 0;
    for (size_t pos = 0; pos <= len && !matched; pos++) {
        /* unanchored search: a new attempt may start at every position */
        matched |= add_thread(re, &cur, 0, text, pos, len);
        if (matched || pos == len) break;
        int c = (unsigned char)text[pos];
        nxt.n = 0;
        memset(nxt.mark, 0, (size_t)re->n);
        for (int i = 0; i < cur.n && !matched; i++) {
            const inst *in = &re->prog[cur.list[i]];
            if (in->op == I_SET && set_has(in->set, c))
                matched |= add_thread(re, &nxt, cur.list[i] + 1, text, pos + 1, len);
        }
        tset tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    free(cur.list);
    free(nxt.list);
    free(cur.mark);
    free(nxt.mark);
    return matched;
}
