/*
 * vo_regex.c — Pattern::new / Pattern::matches for the parity oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Restates the filter semantics of the reference, src/pattern.rs:21-45:
 *   - empty pattern is an error (:22-24)
 *   - case-insensitive patterns are compiled as "(?i)" + pattern (:26-30)
 *   - matches() is regex::Regex::is_match, i.e. an UNANCHORED search over the whole address
 *     string (:43-45); '^' and '$' bind to the ends of the haystack (no multi-line mode).
 * The regex crate (1.12.2) itself is not on disk; this is an independent Thompson-NFA
 * simulation of the syntax subset that can be meaningful on ASCII address strings:
 * literals, escapes, '.', bracket classes with ranges/negation, \d \w \s (+ negations),
 * groups (capturing, (?:..), named), alternation, * + ? {n} {n,} {n,m} (lazy forms accepted:
 * laziness cannot change is_match), ^ $ \A \z, and the inline flag i ((?i), (?-i), (?i:..)).
 * Unsupported syntax (\b, \p{..}, class set operations, flags other than i) is rejected with an
 * error rather than guessed at.
 */
#include "vgen_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { N_EMPTY, N_SET, N_CAT, N_ALT, N_STAR, N_PLUS, N_QUEST, N_BOL, N_EOL };

typedef struct node {
    int type;
    uint8_t set[16]; /* 128-bit ASCII membership for N_SET */
    struct node *a, *b;
} node;

typedef struct {
    const char *p;
    int ci;
    char err[160];
    int failed;
    int nodes;
} parser;

static void set_add(uint8_t s[16], int c) { s[c >> 3] |= (uint8_t)(1u << (c & 7)); }
static int set_has(const uint8_t s[16], int c) { return c < 128 && (s[c >> 3] >> (c & 7)) & 1; }

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
        for (int c = '0'; c <= '9'; c++) set_add(t, c);
        break;
    case 'w':
        for (int c = '0'; c <= '9'; c++) set_add(t, c);
        for (int c = 'a'; c <= 'z'; c++) set_add(t, c);
        for (int c = 'A'; c <= 'Z'; c++) set_add(t, c);
        set_add(t, '_');
        break;
    case 's':
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

/* parses one escape after the backslash; returns a literal byte, or -2 after adding a perl class
 * to `s`, or -1 on error */
static int parse_escape(parser *ps, uint8_t s[16]) {
    int c = (unsigned char)*ps->p;
    if (!c) {
        fail(ps, "trailing backslash");
        return -1;
    }
    ps->p++;
    switch (c) {
    case 'd': case 'D': case 'w': case 'W': case 's': case 'S':
        set_perl(s, (char)c);
        return -2;
    case 'n': return '\n';
    case 't': return '\t';
    case 'r': return '\r';
    case 'f': return '\f';
    case 'v': return '\v';
    case 'x': {
        int h1 = hexval((unsigned char)ps->p[0]);
        int h2 = h1 >= 0 ? hexval((unsigned char)ps->p[1]) : -1;
        if (h1 < 0 || h2 < 0) {
            fail(ps, "bad \\x escape");
            return -1;
        }
        ps->p += 2;
        if (h1 * 16 + h2 > 127) {
            fail(ps, "non-ASCII escape unsupported");
            return -1;
        }
        return h1 * 16 + h2;
    }
    default:
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) {
            fail(ps, "unsupported escape sequence");
            return -1;
        }
        if (c > 127) {
            fail(ps, "non-ASCII pattern unsupported");
            return -1;
        }
        return c; /* escaped punctuation */
    }
}

static node *parse_alt(parser *ps);

static node *parse_class(parser *ps) {
    /* after '[' */
    uint8_t s[16] = {0};
    int neg = 0;
    if (*ps->p == '^') {
        neg = 1;
        ps->p++;
    }
    int first = 1;
    for (;;) {
        int c = (unsigned char)*ps->p;
        if (!c) {
            fail(ps, "unclosed character class");
            return NULL;
        }
        if (c == ']' && !first) {
            ps->p++;
            break;
        }
        first = 0;
        if (c == '[') {
            fail(ps, "nested/POSIX character classes unsupported");
            return NULL;
        }
        if (c == '&' && ps->p[1] == '&') {
            fail(ps, "class set operations unsupported");
            return NULL;
        }
        int lo;
        ps->p++;
        if (c == '\\') {
            lo = parse_escape(ps, s);
            if (lo == -1) return NULL;
            if (lo == -2) continue;
        } else {
            if (c > 127) {
                fail(ps, "non-ASCII pattern unsupported");
                return NULL;
            }
            lo = c;
        }
        int hi = lo;
        if (ps->p[0] == '-' && ps->p[1] && ps->p[1] != ']') {
            ps->p++;
            int d = (unsigned char)*ps->p++;
            if (d == '\\') {
                uint8_t dummy[16] = {0};
                hi = parse_escape(ps, dummy);
                if (hi < 0) {
                    fail(ps, "bad class range");
                    return NULL;
                }
            } else {
                if (d > 127) {
                    fail(ps, "non-ASCII pattern unsupported");
                    return NULL;
                }
                hi = d;
            }
            if (hi < lo) {
                fail(ps, "invalid class range");
                return NULL;
            }
        }
        for (int k = lo; k <= hi; k++) set_add(s, k);
    }
    if (ps->ci) set_fold_case(s);
    if (neg) set_negate(s);
    node *n = mk(ps, N_SET, NULL, NULL);
    memcpy(n->set, s, 16);
    return n;
}

static int parse_int(parser *ps) {
    if (*ps->p < '0' || *ps->p > '9') return -1;
    long v = 0;
    while (*ps->p >= '0' && *ps->p <= '9') {
        v = v * 10 + (*ps->p - '0');
        if (v > 1000) return -2;
        ps->p++;
    }
    return (int)v;
}

static node *parse_atom(parser *ps) {
    int c = (unsigned char)*ps->p;
    if (c == '(') {
        ps->p++;
        int saved_ci = ps->ci;
        if (*ps->p == '?') {
            ps->p++;
            if (*ps->p == 'P' && ps->p[1] == '<') {
                ps->p += 2;
                while (*ps->p && *ps->p != '>') ps->p++;
                if (!*ps->p) {
                    fail(ps, "unclosed group name");
                    return NULL;
                }
                ps->p++;
            } else if (*ps->p == '<') {
                ps->p++;
                while (*ps->p && *ps->p != '>') ps->p++;
                if (!*ps->p) {
                    fail(ps, "unclosed group name");
                    return NULL;
                }
                ps->p++;
            } else {
                /* flags: only i / -i */
                int on = 1, newci = ps->ci, any = 0;
                while (*ps->p && *ps->p != ':' && *ps->p != ')') {
                    if (*ps->p == '-')
                        on = 0;
                    else if (*ps->p == 'i')
                        newci = on;
                    else {
                        fail(ps, "unsupported inline flag");
                        return NULL;
                    }
                    any = 1;
                    ps->p++;
                }
                if (*ps->p == ')') {
                    if (!any) {
                        fail(ps, "empty flag group");
                        return NULL;
                    }
                    ps->p++;
                    ps->ci = newci; /* applies to the rest of the enclosing group */
                    return mk(ps, N_EMPTY, NULL, NULL);
                }
                if (*ps->p != ':') {
                    fail(ps, "unclosed group");
                    return NULL;
                }
                ps->p++;
                ps->ci = newci;
            }
        }
        node *inner = parse_alt(ps);
        if (ps->failed) {
            free_node(inner);
            return NULL;
        }
        if (*ps->p != ')') {
            fail(ps, "unclosed group");
            free_node(inner);
            return NULL;
        }
        ps->p++;
        ps->ci = saved_ci;
        return inner;
    }
    if (c == '[') {
        ps->p++;
        return parse_class(ps);
    }
    if (c == '.') {
        ps->p++;
        node *n = mk(ps, N_SET, NULL, NULL);
        memset(n->set, 0xFF, 16);
        n->set['\n' >> 3] &= (uint8_t)~(1u << ('\n' & 7));
        return n;
    }
    if (c == '^') {
        ps->p++;
        return mk(ps, N_BOL, NULL, NULL);
    }
    if (c == '$') {
        ps->p++;
        return mk(ps, N_EOL, NULL, NULL);
    }
    if (c == '\\') {
        ps->p++;
        if (*ps->p == 'A') {
            ps->p++;
            return mk(ps, N_BOL, NULL, NULL);
        }
        if (*ps->p == 'z') {
            ps->p++;
            return mk(ps, N_EOL, NULL, NULL);
        }
        if (*ps->p == 'b' || *ps->p == 'B' || *ps->p == 'p' || *ps->p == 'P') {
            fail(ps, "unsupported escape (\\b, \\B, \\p)");
            return NULL;
        }
        uint8_t s[16] = {0};
        int lit = parse_escape(ps, s);
        if (lit == -1) return NULL;
        if (lit >= 0) set_add(s, lit);
        if (ps->ci) set_fold_case(s);
        node *n = mk(ps, N_SET, NULL, NULL);
        memcpy(n->set, s, 16);
        return n;
    }
    if (c == '*' || c == '+' || c == '?') {
        fail(ps, "repetition operator missing expression");
        return NULL;
    }
    if (c > 127) {
        fail(ps, "non-ASCII pattern unsupported");
        return NULL;
    }
    ps->p++;
    node *n = mk(ps, N_SET, NULL, NULL);
    set_add(n->set, c);
    if (ps->ci) set_fold_case(n->set);
    return n;
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

static node *parse_repeat(parser *ps) {
    node *atom = parse_atom(ps);
    if (!atom || ps->failed) {
        free_node(atom);
        return NULL;
    }
    for (;;) {
        int c = *ps->p;
        if (c == '*' || c == '+' || c == '?') {
            if (atom->type == N_BOL || atom->type == N_EOL) {
                /* regex crate allows it; semantics are odd but well-defined: keep generic handling */
            }
            ps->p++;
            atom = mk(ps, c == '*' ? N_STAR : c == '+' ? N_PLUS : N_QUEST, atom, NULL);
            if (*ps->p == '?') ps->p++; /* lazy marker: irrelevant for is_match */
        } else if (c == '{') {
            const char *save = ps->p;
            ps->p++;
            int lo = parse_int(ps), hi;
            if (lo == -2) {
                fail(ps, "repetition count too large");
                free_node(atom);
                return NULL;
            }
            if (lo < 0) {
                /* regex crate: a '{' that does not start a counted repetition is an error */
                ps->p = save;
                fail(ps, "invalid repetition");
                free_node(atom);
                return NULL;
            }
            hi = lo;
            if (*ps->p == ',') {
                ps->p++;
                if (*ps->p == '}')
                    hi = -1;
                else {
                    hi = parse_int(ps);
                    if (hi < 0) {
                        fail(ps, "invalid repetition");
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

static node *parse_cat(parser *ps) {
    node *res = mk(ps, N_EMPTY, NULL, NULL);
    while (*ps->p && *ps->p != '|' && *ps->p != ')') {
        node *r = parse_repeat(ps);
        if (!r || ps->failed) {
            free_node(r);
            free_node(res);
            return NULL;
        }
        res = mk(ps, N_CAT, res, r);
    }
    return res;
}

static node *parse_alt(parser *ps) {
    int ci_at_entry = ps->ci;
    node *left = parse_cat(ps);
    if (!left) return NULL;
    while (*ps->p == '|') {
        ps->p++;
        /* an inline (?i) inside one branch keeps applying to later branches of the same group
         * in the regex crate (flags are scoped to the group, not the branch) */
        node *right = parse_cat(ps);
        if (!right) {
            free_node(left);
            return NULL;
        }
        left = mk(ps, N_ALT, left, right);
    }
    (void)ci_at_entry;
    return left;
}

/* ---- NFA program --------------------------------------------------------------------------- */

enum { I_SET, I_SPLIT, I_JMP, I_BOL, I_EOL, I_MATCH };

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
    case N_BOL:
        emit(re, I_BOL);
        break;
    case N_EOL:
        emit(re, I_EOL);
        break;
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
    ps.ci = case_insensitive ? 1 : 0;
    node *ast = parse_alt(&ps);
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

/* follow epsilon edges from pc at haystack position pos (of len); returns 1 if MATCH reached */
static int add_thread(const vo_regex *re, tset *t, int pc, size_t pos, size_t len) {
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
        case I_BOL: if (pos == 0) push1 = q + 1; break;
        case I_EOL: if (pos == len) push1 = q + 1; break;
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
        matched |= add_thread(re, &cur, 0, pos, len);
        if (matched || pos == len) break;
        int c = (unsigned char)text[pos];
        nxt.n = 0;
        memset(nxt.mark, 0, (size_t)re->n);
        for (int i = 0; i < cur.n && !matched; i++) {
            const inst *in = &re->prog[cur.list[i]];
            if (in->op == I_SET && set_has(in->set, c))
                matched |= add_thread(re, &nxt, cur.list[i] + 1, pos + 1, len);
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
