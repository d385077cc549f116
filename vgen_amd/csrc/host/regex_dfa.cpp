// regex_dfa.cpp — see regex_dfa.h.
#include "regex_dfa.h"

#include <algorithm>
#include <map>
#include <memory>
#include <stdexcept>

namespace vg {

namespace {

struct CharSet {
    uint64_t bits[2] = {0, 0};   // ASCII 0..127
    bool other = false;          // bytes >= 128
    void add(int c) { bits[c >> 6] |= 1ull << (c & 63); }
    bool has(int c) const { return c < 128 ? (bits[c >> 6] >> (c & 63)) & 1 : other; }
    void merge(const CharSet &o) {
        bits[0] |= o.bits[0];
        bits[1] |= o.bits[1];
        other = other || o.other;
    }
    void negate() {
        bits[0] = ~bits[0];
        bits[1] = ~bits[1];
        other = !other;
    }
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

struct Node;
using NodeP = std::shared_ptr<Node>;
struct Node {
    enum Type { EMPTY, SET, CAT, ALT, STAR, PLUS, QUEST, BOL, EOL, REPEAT } type = EMPTY;
    CharSet set;
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

class Parser {
  public:
    Parser(const std::string &p, bool ci) : s_(p), ci_(ci) {}

    NodeP parse() {
        NodeP n = alternation();
        if (pos_ < s_.size()) throw ParseError(s_[pos_] == ')' ? "unopened group" : "unexpected character");
        return n;
    }

  private:
    const std::string &s_;
    size_t pos_ = 0;
    bool ci_;

    bool more() const { return pos_ < s_.size(); }
    int peek(size_t k = 0) const { return pos_ + k < s_.size() ? (unsigned char)s_[pos_ + k] : -1; }
    int take() { return (unsigned char)s_[pos_++]; }

    static void perl_class(CharSet &out, int kind) {
        CharSet t;
        switch (kind | 0x20) {
        case 'd':
            for (int c = '0'; c <= '9'; c++) t.add(c);
            break;
        case 'w':
            for (int c = '0'; c <= '9'; c++) t.add(c);
            for (int c = 'a'; c <= 'z'; c++) t.add(c);
            for (int c = 'A'; c <= 'Z'; c++) t.add(c);
            t.add('_');
            break;
        default:   // 's'
            for (int c : {' ', '\t', '\n', '\r', '\f', '\v'}) t.add(c);
            break;
        }
        if (kind >= 'A' && kind <= 'Z') t.negate();
        out.merge(t);
    }

    static int hexval(int c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    // after the backslash; returns a literal byte, or -2 after merging a perl class into cls
    int escape(CharSet &cls) {
        if (!more()) throw ParseError("trailing backslash");
        int c = take();
        switch (c) {
        case 'd': case 'D': case 'w': case 'W': case 's': case 'S':
            perl_class(cls, c);
            return -2;
        case 'n': return '\n';
        case 't': return '\t';
        case 'r': return '\r';
        case 'f': return '\f';
        case 'v': return '\v';
        case 'x': {
            int h1 = hexval(peek(0)), h2 = h1 >= 0 ? hexval(peek(1)) : -1;
            if (h1 < 0 || h2 < 0) throw ParseError("bad \\x escape");
            pos_ += 2;
            if (h1 * 16 + h2 > 127) throw ParseError("non-ASCII escape unsupported");
            return h1 * 16 + h2;
        }
        default:
            if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'))
                throw ParseError("unsupported escape sequence");
            if (c > 127) throw ParseError("non-ASCII pattern unsupported");
            return c;
        }
    }

    NodeP char_class() {
        CharSet cs;
        bool neg = false;
        if (peek() == '^') {
            neg = true;
            pos_++;
        }
        bool first = true;
        for (;;) {
            if (!more()) throw ParseError("unclosed character class");
            int c = peek();
            if (c == ']' && !first) {
                pos_++;
                break;
            }
            first = false;
            if (c == '[') throw ParseError("nested/POSIX character classes unsupported");
            if (c == '&' && peek(1) == '&') throw ParseError("class set operations unsupported");
            pos_++;
            int lo;
            if (c == '\\') {
                lo = escape(cs);
                if (lo == -2) continue;
            } else {
                if (c > 127) throw ParseError("non-ASCII pattern unsupported");
                lo = c;
            }
            int hi = lo;
            if (peek() == '-' && peek(1) != -1 && peek(1) != ']') {
                pos_++;
                int d = take();
                if (d == '\\') {
                    CharSet dummy;
                    hi = escape(dummy);
                    if (hi < 0) throw ParseError("bad class range");
                } else {
                    if (d > 127) throw ParseError("non-ASCII pattern unsupported");
                    hi = d;
                }
                if (hi < lo) throw ParseError("invalid class range");
            }
            for (int k = lo; k <= hi; k++) cs.add(k);
        }
        if (ci_) cs.fold_case();
        if (neg) cs.negate();
        NodeP n = mk(Node::SET);
        n->set = cs;
        return n;
    }

    int integer() {
        if (peek() < '0' || peek() > '9') return -1;
        long v = 0;
        while (peek() >= '0' && peek() <= '9') {
            v = v * 10 + (take() - '0');
            if (v > 1000) throw ParseError("repetition count too large");
        }
        return (int)v;
    }

    NodeP atom() {
        int c = peek();
        if (c == '(') {
            pos_++;
            bool saved = ci_;
            if (peek() == '?') {
                pos_++;
                if ((peek() == 'P' && peek(1) == '<') || peek() == '<') {
                    while (more() && peek() != '>') pos_++;
                    if (!more()) throw ParseError("unclosed group name");
                    pos_++;
                } else {
                    bool on = true, newci = ci_, any = false;
                    while (more() && peek() != ':' && peek() != ')') {
                        int f = take();
                        if (f == '-') on = false;
                        else if (f == 'i') newci = on;
                        else throw ParseError("unsupported inline flag");
                        any = true;
                    }
                    if (peek() == ')') {
                        if (!any) throw ParseError("empty flag group");
                        pos_++;
                        ci_ = newci;   // until the end of the enclosing group
                        return mk(Node::EMPTY);
                    }
                    if (peek() != ':') throw ParseError("unclosed group");
                    pos_++;
                    ci_ = newci;
                }
            }
            NodeP inner = alternation();
            if (peek() != ')') throw ParseError("unclosed group");
            pos_++;
            ci_ = saved;
            return inner;
        }
        if (c == '[') {
            pos_++;
            return char_class();
        }
        if (c == '.') {
            pos_++;
            NodeP n = mk(Node::SET);
            n->set.negate();   // everything ...
            n->set.bits[0] &= ~(1ull << '\n');   // ... except newline
            return n;
        }
        if (c == '^') {
            pos_++;
            return mk(Node::BOL);
        }
        if (c == '$') {
            pos_++;
            return mk(Node::EOL);
        }
        if (c == '\\') {
            pos_++;
            int e = peek();
            if (e == 'A') {
                pos_++;
                return mk(Node::BOL);
            }
            if (e == 'z') {
                pos_++;
                return mk(Node::EOL);
            }
            if (e == 'b' || e == 'B' || e == 'p' || e == 'P') throw ParseError("unsupported escape (\\b, \\B, \\p)");
            CharSet cs;
            int lit = escape(cs);
            if (lit >= 0) cs.add(lit);
            if (ci_) cs.fold_case();
            NodeP n = mk(Node::SET);
            n->set = cs;
            return n;
        }
        if (c == '*' || c == '+' || c == '?') throw ParseError("repetition operator missing expression");
        if (c > 127) throw ParseError("non-ASCII pattern unsupported");
        pos_++;
        NodeP n = mk(Node::SET);
        n->set.add(c);
        if (ci_) n->set.fold_case();
        return n;
    }

    NodeP repeat() {
        NodeP a = atom();
        for (;;) {
            int c = peek();
            if (c == '*' || c == '+' || c == '?') {
                pos_++;
                a = mk(c == '*' ? Node::STAR : c == '+' ? Node::PLUS : Node::QUEST, a);
                if (peek() == '?') pos_++;   // lazy marker: irrelevant for is_match
            } else if (c == '{') {
                pos_++;
                int lo = integer();
                if (lo < 0) throw ParseError("invalid repetition");
                int hi = lo;
                if (peek() == ',') {
                    pos_++;
                    if (peek() == '}') hi = -1;
                    else {
                        hi = integer();
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

    NodeP concat() {
        NodeP res = mk(Node::EMPTY);
        while (more() && peek() != '|' && peek() != ')') res = mk(Node::CAT, res, repeat());
        return res;
    }

    NodeP alternation() {
        NodeP left = concat();
        while (peek() == '|') {
            pos_++;
            left = mk(Node::ALT, left, concat());
        }
        return left;
    }
};

// ---- NFA -------------------------------------------------------------------------------------------------

struct NState {
    enum Type { CHAR, SPLIT, BOL, EOL, MATCH } type;
    CharSet set;
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
        case Node::BOL:
            return add(NState::BOL, next);
        case Node::EOL:
            return add(NState::EOL, next);
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

// epsilon closure of `seeds`; BOL edges are followed only when at_start, EOL edges only when at_end.
// Result: sorted list of CHAR / MATCH / (pending) EOL states.
void closure(const Nfa &nfa, const std::vector<int> &seeds, bool at_start, bool at_end, std::vector<int> &out,
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
        case NState::BOL:
            if (at_start) stack.push_back(n.a);
            break;
        case NState::EOL:
            if (at_end) stack.push_back(n.a);
            else out.push_back(s);
            break;
        default:
            out.push_back(s);
        }
    }
    std::sort(out.begin(), out.end());
}

}  // namespace

bool Dfa::is_match(const char *text) const {
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

        // symbol classes: bytes with identical membership in every CHAR set
        std::vector<const CharSet *> sets;
        for (auto &s : nfa.st)
            if (s.type == NState::CHAR) sets.push_back(&s.set);
        std::map<std::vector<bool>, uint8_t> sig2cls;
        std::vector<int> cls_rep;   // representative byte per class
        for (int c = 0; c <= 128; c++) {
            std::vector<bool> sig(sets.size());
            for (size_t i = 0; i < sets.size(); i++) sig[i] = sets[i]->has(c);
            auto it = sig2cls.find(sig);
            uint8_t id;
            if (it == sig2cls.end()) {
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
        out.n_cls = (uint32_t)cls_rep.size();

        std::vector<uint32_t> mark(nfa.st.size(), 0);
        uint32_t epoch = 0;
        std::map<std::vector<int>, uint32_t> ids;
        std::vector<std::vector<int>> states;
        std::vector<int> tmp, tmp2;

        auto intern = [&](const std::vector<int> &set) -> uint32_t {
            auto it = ids.find(set);
            if (it != ids.end()) return it->second;
            if (states.size() >= 20000) throw ParseError("pattern needs too many DFA states");
            uint32_t id = (uint32_t)states.size();
            ids[set] = id;
            states.push_back(set);
            return id;
        };

        // state 0 is the start-of-haystack state; the -1 sentinel keeps it distinct from a later state
        // with the same NFA set (which differs in that '^' can no longer hold)
        closure(nfa, {start}, true, false, tmp, mark, epoch);
        tmp.insert(tmp.begin(), -1);
        intern(tmp);
        out.trans.clear();
        out.match_now.clear();
        out.match_at_end.clear();
        for (uint32_t si = 0; si < states.size(); si++) {
            const std::vector<int> cur = states[si];
            bool now = std::binary_search(cur.begin(), cur.end(), match);
            // end-of-haystack acceptance: follow pending EOL edges (and BOL when nothing was consumed,
            // i.e. only from the start state)
            closure(nfa, cur, si == 0, true, tmp2, mark, epoch);
            bool at_end = std::binary_search(tmp2.begin(), tmp2.end(), match);
            out.match_now.push_back(now);
            out.match_at_end.push_back(at_end);
            out.trans.resize((size_t)(si + 1) * out.n_cls);
            for (uint32_t c = 0; c < out.n_cls; c++) {
                std::vector<int> seeds;
                int rep = cls_rep[c];
                for (int s : cur) {
                    if (s < 0) continue;
                    const NState &n = nfa.st[s];
                    if (n.type == NState::CHAR && n.set.has(rep)) seeds.push_back(n.a);
                }
                seeds.push_back(start);   // unanchored search: a new attempt may begin at every offset
                closure(nfa, seeds, false, false, tmp, mark, epoch);
                out.trans[(size_t)si * out.n_cls + c] = now ? si : intern(tmp);
            }
        }
        out.n_states = (uint32_t)states.size();

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
    } catch (const ParseError &e) {
        err = "Invalid regex pattern: " + pattern + " (" + e.what() + ")";
        return false;
    }
}

}  // namespace vg
