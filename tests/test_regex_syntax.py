"""Pattern::new accepts whatever regex::Regex::new accepts (reference src/pattern.rs:21-45).  These CPU tests cover
the syntax beyond the basics — flags s m x U u with their scoped / negated forms, word boundaries, Unicode and POSIX
classes restricted to ASCII, nested classes and the set operators — in BOTH independent implementations: the
product's DFA (vgen_amd/csrc/host/regex_dfa.cpp, via tests/native) and the oracle's NFA simulation
(oracle/vo_regex.c).  Three-way differential against Python's `re` wherever the two dialects can express the same
thing (patterns are generated from a small AST and printed in both dialects); product-vs-oracle plus hand-written
expectations (from the regex crate's documentation) for what Python lacks.
"""
import ctypes
import os
import random
import re
import subprocess

import pytest

from conftest import locked_make

from oracle import pyoracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def core():
    locked_make("-s", "-C", os.path.join(HERE, "native"), "libcoretest.so")
    return ctypes.CDLL(os.path.join(HERE, "native", "libcoretest.so"))


def product(core, pat, ci, text):
    return core.core_regex_match(pat.encode(), int(ci), text.encode())


def oracle(pat, ci, text):
    try:
        return int(vo.Regex(pat, ci).matches(text))
    except ValueError:
        return -1


# (pattern, text, expected) — semantics as documented for the regex crate
EXPECT = [
    (r"\bcat\b", "a cat b", 1), (r"\bcat\b", "concat", 0), (r"\Bcat", "concat", 1), (r"\Bcat", "a cat", 0),
    (r"(?m)^b$", "a\nb\nc", 1), (r"^b$", "a\nb\nc", 0), (r"(?m)^$", "a\n\nb", 1), (r"(?m:^)b", "a\nb", 1),
    (r"(?s)a.b", "a\nb", 1), (r"a.b", "a\nb", 0), (r"(?s:a.)b", "a\nb", 1), (r"(?s)a(?-s:.)b", "a\nb", 0),
    ("(?x) a b # comment\n c", "abc", 1), (r"(?x)a\ b", "a b", 1), (r"(?x)a b", "a b", 0), (r"(?x)[ a b ]", "a", 1),
    (r"(?x)[ a b ]", " ", 0), (r"(?x) a {2} ", "aa", 1), (r"(?x:a b)c d", "abc d", 1), (r"(?x:a b)c d", "abcd", 0),
    (r"[[:alpha:]]+\d", "ab1", 1), (r"[[:^alpha:]]", "abc", 0), (r"[[:^alpha:]]", "ab1", 1), (r"[[:xdigit:]]{3}", "fG09a", 1),
    (r"[[:punct:]]", "ab_", 1), (r"[[:word:]-]+$", "a-b_c", 1), (r"[[:space:]]", "a\tb", 1), (r"[[:alnum:][:punct:]]$", " ", 0),
    (r"[a-z&&[^aeiou]]", "a", 0), (r"[a-z&&[^aeiou]]", "b", 1), (r"[a-z--m-z]", "n", 0), (r"[a-z--m-z]", "c", 1),
    (r"[a-c~~b-d]", "a", 1), (r"[a-c~~b-d]", "b", 0), (r"[a-c~~b-d]", "d", 1), (r"[a-z&&b-y--c]", "c", 0), (r"[a-z&&b-y--c]", "d", 1),
    (r"[^a-z&&b]", "b", 0), (r"[^a-z&&b]", "a", 1), (r"[\w&&[^\d_]]+", "__1a", 1), (r"[\w&&[^\d_]]+$", "a1", 0),
    (r"[a&&b]", "a", 0), (r"[[a-c][x-z]]", "y", 1), (r"[a[b[c]]]+$", "abc", 1), (r"[\[\]]", "]", 1), (r"[]a]", "]", 1), (r"[^]a]", "]", 0),
    (r"\p{Lu}\p{Ll}+", "Hello", 1), (r"\pN", "a1", 1), (r"\PN", "11", 0), (r"\p{^L}", "ab", 0), (r"\p{^L}", "a1", 1),
    (r"\p{Greek}", "abc", 0), (r"\P{Greek}", "a", 1), (r"\p{Tamil}", "abc", 0), (r"\P{Old_Turkic}+$", "a1_", 1), (r"[\p{Hangul}a]", "a", 1),
    (r"\p{sc=Canadian_Aboriginal}", "x", 0), (r"\p{gc=Lu}", "aB", 1), (r"\p{gc!=Lu}", "AB", 0), (r"\p{sc=Latin}+$", "Ab", 1),
    (r"\p{Uppercase_Letter}", "ab", 0), (r"\p{ uppercase-LETTER }", "aB", 1), (r"[\p{Nd}\p{Lu}]+$", "A1B2", 1), (r"\p{P}", "a-b", 1),
    (r"\p{S}", "a-b", 0), (r"\p{S}", "a+b", 1), (r"\p{Zs}", "a b", 1), (r"\p{Cc}", "a\tb", 1), (r"\p{Alphabetic}+$", "ab", 1),
    (r"\p{ASCII_Hex_Digit}{2}$", "fF", 1), (r"\p{Any}", "", 0), (r"\p{ascii}+$", "a~", 1), (r"\p{Common}", "ab", 0), (r"\p{Pc}", "a_b", 1),
    (r"(?i:a)b", "Ab", 1), (r"(?i:a)b", "AB", 0), (r"(?i)a(?-i)b", "AB", 0), (r"(?i)a(?-i)b", "Ab", 1), (r"a(?i)b|c", "aB", 1),
    (r"a(?i)b|c", "C", 1), (r"(a(?i)b)c", "aBC", 0), (r"(a(?i)b)c", "aBc", 1), (r"(?is-m:A.b)", "a\nB", 1), (r"(?i-s:A.b)", "a\nB", 0),
    (r"\x41\x{42}\u0043\u{44}\U00000045", "ABCDE", 1), (r"\a", "\a", 1), (r"\x{1F600}", "a", 0), (r"[\x41-\x{5A}]+$", "AZ", 1),
    (r"(?U)a+?", "aa", 1), (r"(?U)a*b", "aab", 1), (r"(?u)\w", "a", 1), (r"(?-u:\w)", "a", 1),
    (r"\b{start}cat", "a cat", 1), (r"\b{start}at", "a cat", 0), (r"\b{end}", "cat", 1), (r"\b{end}", " ", 0), (r"\<c", "a c", 1),
    (r"a\>", " ab", 0), (r"b\>", " ab", 1), (r"\b{start-half}a", "ba", 0), (r"\b{start-half}a", " a", 1), (r"a\b{end-half}", "ab", 0),
    (r"a\b{end-half}", "a!", 1), ("é", "e", 0), ("[^é]", "e", 1), ("[a-é]", "k", 1), ("(?i)ſ", "S", 1), ("(?i)K", "k", 1), ("ſ", "s", 0),
    ("(?i)[ſ]", "s", 1), (r"$^", "", 1), (r"(?m)$^", "a\n", 1), (r"x*", "", 1), (r"\b", "", 0), (r"\B", "", 1), (r"\b", "a", 1),
    (r"(?i)[^a]", "A", 0), (r"(?i)[^a-c&&b]", "B", 0), (r"(?i)[a-c&&[^b]]", "B", 0), (r"(?i)[a-c--b]", "B", 0), (r"(?i)[a-c--b]", "C", 1),
    (r"(?m)^1Cat$", "x\n1Cat\ny", 1), (r"^1Cat$", "x\n1Cat\ny", 0), (r"\A1", "1x", 1), (r"x\z", "1x", 1), (r"x\z", "1x\n", 0),
    (r"(?P<n>a)(?<m>b)(?:c)", "abc", 1), (r"a{2}{3}$", "aaaaaa", 1), (r"a**", "", 1), (r"(?:)+", "", 1), (r"a{1,}?b", "aab", 1),
]

# patterns regex::Regex::new rejects (or that need non-ASCII-only features this front end does not have)
REJECT = [r"(?R)a", r"\p{Foo}", r"\p{gcx=L}", r"(?z)", r"(?=a)", r"(?!a)", r"(?<=a)", r"a{2,1}", r"[z-a]", r"(a", r"a)", r"\q", r"\Z", r"\1",
          r"*a", r"+", r"(?i", r"(?i-)", r"(?)", r"(?--i)", r"[a", r"[]", r"a{,3}", r"a{x}", r"\x4", r"\x{110000}", r"\u{D800}", r"\p{", r"\p",
          r"[a-\d]", r"\ ", r"(?P<>a)", r"[[:alpha:]", ""]


def test_documented_semantics_in_both_implementations(core):
    for pat, text, want in EXPECT:
        assert product(core, pat, False, text) == want, ("product", pat, text)
        assert oracle(pat, False, text) == want, ("oracle", pat, text)


def test_invalid_patterns_are_errors_in_both_implementations(core):
    for pat in REJECT:
        assert product(core, pat, False, "a") == -1, ("product", pat)
        assert oracle(pat, False, "a") == -1, ("oracle", pat)


# ---- generated patterns, printed in the regex-crate dialect and in Python's ---------------------------------------

ALPHA = "ab1_ -\n"


class Gen:
    def __init__(self, rng):
        self.rng = rng

    def lit(self):
        c = self.rng.choice("ab1_ -")   # none of these needs escaping outside a class ("\\ " is only valid in x mode)
        return c, c

    def cls(self):
        rng = self.rng
        items, neg = [], rng.random() < 0.3
        for _ in range(rng.randrange(1, 4)):
            k = rng.random()
            if k < 0.5:
                items.append(re.escape(rng.choice("ab1_-")))
            elif k < 0.75:
                items.append(rng.choice(["a-b", "0-9", "a-z", "A-Z"]))
            else:
                items.append(rng.choice([r"\d", r"\w", r"\s", r"\D"]))
        body = ("^" if neg else "") + "".join(items)
        return "[" + body + "]", "[" + body + "]"

    def atom(self, depth, flags):
        rng = self.rng
        k = rng.random()
        if k < 0.35:
            return self.lit()
        if k < 0.5:
            return self.cls()
        if k < 0.58:
            return ".", "."
        if k < 0.70:
            a = rng.choice([r"\b", r"\B", "^", "$", r"\A", r"\z"])
            py = {"$": "$" if "m" in flags else r"\Z", r"\z": r"\Z"}.get(a, a)
            return a, py
        if k < 0.78:
            return rng.choice([(r"\d", r"\d"), (r"\w", r"\w"), (r"\s", r"\s"), (r"\W", r"\W")])
        if depth <= 0:
            return self.lit()
        if k < 0.9:
            r, p = self.alt(depth - 1, flags)
            return "(" + r + ")", "(" + p + ")"
        # scoped flag group
        on = "".join(sorted(set(rng.sample("ims", rng.randrange(1, 3)))))
        off = "".join(f for f in "ims" if f not in on and f in flags and rng.random() < 0.5)
        spec = on + ("-" + off if off else "")
        nf = (flags | set(on)) - set(off)
        r, p = self.alt(depth - 1, nf)
        return "(?" + spec + ":" + r + ")", "(?" + spec + ":" + p + ")"

    def rep(self, depth, flags):
        r, p = self.atom(depth, flags)
        k = self.rng.random()
        if k < 0.6:
            return r, p
        q = self.rng.choice(["*", "+", "?", "{2}", "{1,2}", "{0,1}", "*?", "+?"])
        # Python refuses to repeat a bare assertion or a repetition; wrap both dialects alike
        return "(?:" + r + ")" + q, "(?:" + p + ")" + q

    def cat(self, depth, flags):
        parts = [self.rep(depth, flags) for _ in range(self.rng.randrange(1, 4))]
        return "".join(x for x, _ in parts), "".join(y for _, y in parts)

    def alt(self, depth, flags):
        parts = [self.cat(depth, flags) for _ in range(1 if self.rng.random() < 0.7 else 2)]
        return "|".join(x for x, _ in parts), "|".join(y for _, y in parts)


def test_three_way_differential_on_generated_patterns(core):
    rng = random.Random(20261004)
    g = Gen(rng)
    texts = ["", "a", "b", "ab", "a b", "a\nb", "\n", "a_1", "1-a", " a", "b ", "ab\n", "\nab", "a\n\nb", "A", "aB1"]
    for _ in range(60):
        texts.append("".join(rng.choice(ALPHA + "AB") for _ in range(rng.randrange(1, 9))))
    checked = 0
    for _ in range(700):
        rust, py = g.alt(2, set())
        for ci in (False, True):
            try:
                pyre = re.compile(("(?i)" if ci else "") + py)
            except re.error:
                continue
            o = vo.Regex(rust, ci)     # must compile: the generator only emits syntax both dialects have
            for t in texts:
                if t == "" and "\\B" in rust:
                    continue   # Python's \B never matches in an empty string; the regex crate's does
                want = int(pyre.search(t) is not None)
                assert product(core, rust, ci, t) == want, ("product", rust, py, ci, t)
                assert int(o.matches(t)) == want, ("oracle", rust, py, ci, t)
                checked += 1
    assert checked > 50000


def rust_only_pattern(rng, depth=2):
    """Random patterns over the syntax Python lacks: class set operators, nesting, POSIX and Unicode classes,
    \\b{..} forms, verbose mode.  Compared product-vs-oracle only."""
    def cls(d):
        parts = []
        for _ in range(rng.randrange(1, 4)):
            k = rng.random()
            if k < 0.3:
                parts.append(rng.choice(["a-c", "b", "0-9", "_", "A-Z", r"\-"]))
            elif k < 0.5:
                parts.append(rng.choice(["[:alpha:]", "[:digit:]", "[:^lower:]", "[:punct:]", "[:word:]", "[:space:]", "[:xdigit:]"]))
            elif k < 0.7:
                parts.append(rng.choice([r"\pL", r"\p{Lu}", r"\PN", r"\p{Punct}", r"\p{^Ll}", r"\d", r"\W", r"\p{Greek}", r"\p{S}"]))
            elif d > 0:
                parts.append(cls(d - 1))
            else:
                parts.append("x")
        body = "".join(parts)
        if d > 0 and rng.random() < 0.5:
            body += rng.choice(["&&", "--", "~~"]) + cls(d - 1)
        return "[" + ("^" if rng.random() < 0.3 else "") + body + "]"

    pieces = []
    for _ in range(rng.randrange(1, 4)):
        k = rng.random()
        if k < 0.5:
            pieces.append(cls(depth) + rng.choice(["", "", "+", "*", "?", "{2}"]))
        elif k < 0.65:
            pieces.append(rng.choice([r"\b{start}", r"\b{end}", r"\<", r"\>", r"\b{start-half}", r"\b{end-half}", r"\b", r"\B"]))
        elif k < 0.8:
            pieces.append(rng.choice([r"\pL", r"\p{Nd}", r"\P{L}", r"\p{gc=P}", "a", "1", "_", " "]))
        else:
            pieces.append("(?x: a [ b c ] # note\n )")
    return ("(?i)" if rng.random() < 0.25 else "") + "".join(pieces)


def test_product_and_oracle_agree_on_syntax_python_lacks(core):
    rng = random.Random(77)
    texts = ["", "a", "B", "ab", "a b", "a_1", "1-a", " a", "b ", "aB1", "A-", "_", "x y", "abc", "ABC", "a1!", "~", "é".encode().decode("latin-1")[:0]]
    for _ in range(40):
        texts.append("".join(rng.choice("abcABC019_ -!+x") for _ in range(rng.randrange(1, 8))))
    n = 0
    for _ in range(1200):
        pat = rust_only_pattern(rng)
        a = product(core, pat, False, "a")
        b = oracle(pat, False, "a")
        assert (a == -1) == (b == -1), pat
        if a == -1:
            continue
        o = vo.Regex(pat, False)
        for t in texts:
            assert product(core, pat, False, t) == int(o.matches(t)), (pat, t)
            n += 1
    assert n > 20000


def test_exotic_patterns_still_compile_to_device_filters(core):
    """The extended syntax flows through filter.cpp: a pattern with flags / classes keeps its prefix prefilter."""
    lib = core
    lib.core_filter_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int,
                                      ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    rng = random.Random(5)
    payloads = [bytes(rng.randrange(256) for _ in range(20)) for _ in range(2000)]
    for pat, want_kinds in [(r"(?x) ^ 1 C a t", (1,)), (r"(?m)^1[[:upper:]]at", (1, 4)), (r"\A1\p{Lu}[a-z&&[^b-z]]", (1,)),
                            (r"(?s)^1Cat.*", (1,)), (r"\b1Cat", (1, 4))]:
        flags = ctypes.create_string_buffer(len(payloads))
        kind, sel = ctypes.c_int(), ctypes.c_double()
        assert lib.core_filter_check(pat.encode(), 0, 0, b"".join(payloads), len(payloads), flags, ctypes.byref(kind), ctypes.byref(sel)) == 0
        assert kind.value in want_kinds, (pat, kind.value)
        for fl in flags.raw:
            assert (fl & 1) or not (fl & 2)      # the device prefilter never rejects an exact match


# ---- patterns whose DFA is not built: an end anchor (or a word boundary) behind a counted wildcard ---------------
LAZY = [("a[ab]{14}$", False), ("a.{20}$", False), ("[ab]*a[ab]{30}$", False), (r"(?i)x.{15}\b", False), ("A.{16}$", True),
        ("^1.*a.{18}$", False), (r"(a|bb).{17}$", False)]


@pytest.mark.parametrize("pat, ci", LAZY)
def test_patterns_too_large_to_determinise_are_matched_by_walking_the_nfa(core, pat, ci):
    """regex::Regex compiles "a.{20}$" without blinking (its own engines fall back to an NFA walk); a table of the 2^21 sets
    of offsets such a pattern remembers is not built here either: regex_compile keeps the NFA (Dfa::lazy) and is_match walks
    it.  Three-way differential: product, oracle (an NFA simulation of its own), Python's re."""
    rng = random.Random(len(pat))
    flags = re.I if ci else 0
    hits = 0
    ora = vo.Regex(pat, ci)
    for _ in range(90):   # (the test shim compiles the pattern per call: ~25 ms each)
        head = "".join(rng.choice("abAxX1 c") for _ in range(rng.randrange(0, 9)))
        tail = "".join(rng.choice("abxA") if rng.random() < 0.04 else rng.choice("ab") for _ in range(rng.randrange(8, 36)))
        text = rng.choice(["", "1"]) + head + rng.choice(["a", "x", "bb", "A", "X"]) + tail
        want = int(re.search(pat, text, flags) is not None)
        assert product(core, pat, ci, text) == want == int(ora.matches(text)), (pat, text)
        hits += want
    assert 0 < hits < 90, hits
