"""CPU tests of the product's pattern front end (vgen_amd/csrc/host/regex_dfa.cpp, filter.cpp) and of
the device prefilter program, evaluated on the host by the same filter_eval() the kernel runs.

Contract under test: (1) the DFA decides exactly what the oracle's regex and Python's `re` decide;
(2) the device prefilter NEVER rejects an address the exact DFA accepts (superset), and for
prefix/suffix patterns it is tight.
"""
import ctypes
import json
import os
import random
import re
import subprocess

import pytest

from conftest import locked_make

from oracle import pyoracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


@pytest.fixture(scope="module")
def core():
    locked_make("-s", "-C", os.path.join(HERE, "native"))
    lib = ctypes.CDLL(os.path.join(HERE, "native", "libcoretest.so"))
    lib.core_filter_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int,
                                      ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    return lib


def rmatch(core, pat, ci, text):
    return core.core_regex_match(pat.encode(), int(ci), text.encode())


def test_reference_pattern_cases(core):
    for p in KAT["pattern"]["valid"]:
        assert rmatch(core, p, False, "x") >= 0
    for p in KAT["pattern"]["invalid"]:
        assert rmatch(core, p, False, "x") == -1
    for c in KAT["pattern"]["cases"]:
        assert rmatch(core, c["pattern"], c["ci"], c["text"]) == int(c["match"]), c


PATTERNS = [
    "^1Cat", "dead$", "^bc1q.*dead$", "1[Oo]ri", "^1[a-z]{3}7", "(?i)^1cat", "^1(Cat|Dog)s?",
    "^0x[0-9a-f]{4}dead", "a{2,3}b", "^1.*z$", "x+y*z?", "[^a-z]{5}$", "^(1|3)[A-H]", "\\d{3}",
    "^bc1q(aa|zz)+", "q$|^1A", "(?i)DEAD$", "^1[^0-9]+9", "\\w\\d\\w$", "^.{34}$", "(ab)*c", "a|",
    "^$", "^1(?i:cat)X", "$^", "a{0}b", "(a|b){2,}c$", "^1A?B?C?D", ".", "^x*$",
]


def test_dfa_agrees_with_oracle_and_python_re(core):
    rng = random.Random(13)
    alphabet = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz0x"
    texts = ["", "1", "1Cat", "bc1qdead", "0xdeadbeef", "1CATX", "x", "xx", "ab", "abab", "ababc", "b"]
    for _ in range(250):
        texts.append("".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 45))))
    texts += ["1Cat" + t for t in texts[12:40]] + [t + "dead" for t in texts[40:80]]
    texts += ["bc1q" + t + "dead" for t in texts[80:100]] + ["1" + t + "z" for t in texts[100:120]]
    for pat in PATTERNS:
        for ci in (False, True):
            oracle = vo.Regex(pat, ci)
            py = re.compile(("(?i)" if ci else "") + pat.replace("\\d", "[0-9]").replace("\\w", "[0-9A-Za-z_]").replace("$", "\\Z"))
            for t in texts:
                got = rmatch(core, pat, ci, t)
                assert got == int(oracle.matches(t)) == int(py.search(t) is not None), (pat, ci, t)


def test_unsupported_syntax_rejected(core):
    # (word boundaries, Unicode / POSIX classes, class set operators and the flags m s x U are supported:
    # tests/test_regex_syntax.py; what is left is what the regex crate rejects too, and the CRLF flag)
    for pat in ["(?R)^a", "\\p{NoSuchClass}", "(?=a)", "a{,3}", "*a", "(a", "a)", "", "[z-a]", "\\"]:
        assert rmatch(core, pat, False, "x") == -1, pat


def check(core, pat, ci, fmt, payloads):
    flags = ctypes.create_string_buffer(len(payloads))
    kind, sel = ctypes.c_int(), ctypes.c_double()
    blob = b"".join(payloads)
    assert core.core_filter_check(pat.encode(), int(ci), fmt, blob, len(payloads), flags, ctypes.byref(kind),
                                  ctypes.byref(sel)) == 0
    dev = [b & 1 for b in flags.raw]
    exact = [(b >> 1) & 1 for b in flags.raw]
    return kind.value, sel.value, dev, exact


def address(core, fmt, payload):
    out = ctypes.create_string_buffer(128)
    core.core_address(fmt, payload, out)
    return out.value.decode()


def oracle_address(fmt, payload):
    return vo.segwit_addr("bc", 1, payload) if fmt == 3 else vo.address_from_hash160(fmt, payload)


@pytest.mark.parametrize("fmt", [0, 1, 2, 3, 4, 5])
def test_prefilter_is_a_tight_superset_for_prefixes_and_suffixes(core, fmt):
    rng = random.Random(100 + fmt)
    pb = 32 if fmt == 3 else 20                       # P2TR: the payload is the x-only output key
    payloads = [bytes(rng.randrange(256) for _ in range(pb)) for _ in range(3000)]
    # leading-zero payloads change Base58 address lengths ('11...' prefixes)
    payloads += [bytes(k) + bytes(rng.randrange(256) for _ in range(pb - k)) for k in (1, 2, 3, 5) for _ in range(50)]
    payloads += [bytes(pb), bytes([255] * pb)]
    addrs = [address(core, fmt, p) for p in payloads]
    for a, p in zip(addrs, payloads):
        assert a == oracle_address(fmt, p)            # product encoder == oracle encoder
    head = {0: 1, 4: 1, 2: 1, 1: 4, 3: 4, 5: 2}[fmt]  # characters every address of the format shares
    pats = []
    for a in rng.sample(addrs, 40):
        k = rng.randrange(1, 5)
        pats.append(("^" + re.escape(a[:head + k]), False))
        if fmt in (1, 3, 5):
            pats.append((re.escape(a[-k:]) + "$", fmt == 5))
            pats.append(("^" + re.escape(a[:head + 1]) + ".*" + re.escape(a[-2:]) + "$", fmt == 5))
    pats += [("^" + re.escape(addrs[0][:head + 2]) + "|^" + re.escape(addrs[1][:head + 3]), False)]
    if fmt in (0, 4):
        pats += [("^11", False), ("^111", False), ("^1[1-3]", False)]
    if fmt == 5:
        pats += [("^0xDEAD", False), ("^0xdead", True), ("(?i)^0xAbC", False)]
    if fmt == 3:
        # the 52nd data symbol holds one payload bit and four pad bits: only 'q' and 's' can appear there
        pats += [("^bc1p" + "." * 51 + "q", False), ("^bc1p" + "." * 51 + "[ac]", False), ("^bc1q", False)]
    for pat, ci in pats:
        kind, sel, dev, exact = check(core, pat, ci, fmt, payloads)
        assert kind in (1, 2, 3) or (kind == 4 and fmt == 3), (pat, kind)   # unselective ones go to the device DFA
        for d, e, a in zip(dev, exact, addrs):
            assert d or not e, f"prefilter rejected a real match: {pat} {a}"
        oracle = vo.Regex(pat, ci)
        assert exact == [int(oracle.matches(a)) for a in addrs], pat
        if kind != 4:
            assert sum(dev) <= 3 * sum(exact) + 12, (pat, sum(dev), sum(exact))


def test_unanchored_and_base58_suffix_patterns_use_the_device_dfa(core):
    payloads = [bytes(20)]
    for pat, fmt in [("Cat", 0), ("abc$", 0), ("dead", 1), ("[0-9]{6}", 5)]:
        kind, sel, dev, exact = check(core, pat, False, fmt, payloads)
        assert kind == 4 and all(dev), (pat, kind)    # full on-device match; the prefilter stage passes everything
    for pat, fmt in [(".", 0), ("^1", 0), ("^bc1q", 1), ("^0x", 5), ("^3", 2)]:
        kind, sel, dev, exact = check(core, pat, False, fmt, payloads)
        assert kind == 3 and all(dev) and all(exact)
    kind, sel, dev, exact = check(core, "^1Cat", False, 0, payloads)
    assert kind == 1 and 1e-7 < sel < 1e-4            # ~ 58^-3 plus length variants
    kind, sel, dev, exact = check(core, "^3Cat", False, 0, payloads)   # impossible for P2PKH
    assert not any(dev) and not any(exact)


def test_suffix_prefilter_uses_the_whole_literal(core):
    payloads = [bytes(20)]
    kind, sel, _, _ = check(core, "dead$", False, 1, payloads)          # BASELINE config 3
    assert kind == 2 and abs(sel - 32.0 ** -4) < 1e-9
    kind, sel, _, _ = check(core, "deadbeef$", True, 5, payloads)
    assert kind == 2 and abs(sel - 16.0 ** -8) < 1e-12
    kind, sel, _, _ = check(core, "^bc1qq.*xyz$", False, 1, payloads)    # prefix x suffix
    assert kind == 2 and abs(sel - 32.0 ** -4) < 1e-9
    kind, sel, _, _ = check(core, "(aa|zz)$", False, 1, payloads)
    assert kind == 2 and abs(sel - 2 * 32.0 ** -2) < 1e-9


@pytest.mark.parametrize("fmt", [0, 1, 2, 3, 4, 5])
def test_device_full_match_algorithm_equals_exact_dfa(core, fmt):
    """DEVF_DFA: the on-device encode + DFA walk (core/dfa_eval.h, run here on the host) decides exactly what
    the DFA decides on the encoded address — Base58Check incl. checksum digits and leading '1's, Bech32 incl.
    checksum symbols; for Ethereum it is the case-folded language (a superset of the exact one)."""
    core.core_dfa_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int,
                                    ctypes.c_char_p]
    rng = random.Random(500 + fmt)
    pb = 32 if fmt == 3 else 20
    payloads = [bytes(rng.randrange(256) for _ in range(pb)) for _ in range(1500)]
    payloads += [bytes(k) + bytes(rng.randrange(256) for _ in range(pb - k)) for k in (1, 2, 3, 7, 19) for _ in range(20)]
    payloads += [bytes(pb), bytes([255] * pb), bytes(pb - 1) + b"\x01"]
    addrs = [address(core, fmt, p) for p in payloads]
    pats = {0: ["Cat", "1[Oo]ri", "abc$", "[0-9]{4}$", "AA.*zz", "^1.*7$", "(?i)dead", "11"],
            4: ["Cat", "xyz$", "1111"],
            2: ["Cat", "abc$", "^3.*9$", "(?i)beef"],
            1: ["dead", "[0-9]{5}", "q{3}", "xyz.*acd", "de.*ad"],
            3: ["dead", "[0-9]{5}", "q{3}", "xyz.*acd", "de.*ad", "bc1p.*p$"],
            5: ["dead", "[0-9]{6}", "00.*ff", "(?i)BEEF.*f$", "Aa"]}[fmt]
    # patterns built from real addresses so that matches exist
    for a in rng.sample(addrs, 6):
        mid = len(a) // 2
        pats.append(re.escape(a[mid:mid + 3]))
        pats.append(re.escape(a[-3:]) + "$")
    blob = b"".join(payloads)
    n_dfa = 0
    for pat in pats:
        flags = ctypes.create_string_buffer(len(payloads))
        kind = core.core_dfa_check(pat.encode(), 0, fmt, blob, len(payloads), flags)
        if kind != 4:      # the pattern has a cheap prefilter (suffix masks on Bech32 / hex): covered elsewhere
            assert kind in (1, 2, 3), (pat, kind)
            continue
        n_dfa += 1
        dev = [b & 1 for b in flags.raw]
        exact = [(b >> 1) & 1 for b in flags.raw]
        oracle = vo.Regex(pat, False)
        assert exact == [int(oracle.matches(a)) for a in addrs], pat
        if fmt == 5:
            folded = re.compile(pat, re.I)
            assert dev == [int(folded.search(a) is not None) for a in addrs], pat
            assert all(d or not e for d, e in zip(dev, exact))
        else:
            assert dev == exact, (pat, [a for a, d, e in zip(addrs, dev, exact) if d != e][:3])
    assert n_dfa >= 5


def generalise(rng, a, head, alphabet, fmt):
    """A pattern that matches address `a`, made from it the way users write vanity patterns: an anchored prefix and/or
    suffix (or a piece of the middle) in which characters are replaced by classes, dots, alternatives, optional characters
    and counted wildcards.  Returns (pattern, case_insensitive)."""
    def soften(piece):
        out = []
        for c in piece:
            x = rng.random()
            if x < 0.55:
                out.append(re.escape(c))
            elif x < 0.75:
                members = {c} | {rng.choice(alphabet) for _ in range(rng.randrange(1, 4))}
                out.append("[" + "".join(sorted(members)) + "]")
            elif x < 0.85:
                out.append(".")
            elif x < 0.92:
                out.append("(?:" + re.escape(c) + "|" + re.escape(rng.choice(alphabet)) + rng.choice(["", re.escape(rng.choice(alphabet))]) + ")")
            else:
                out.append(re.escape(c) + re.escape(rng.choice(alphabet)) + "?")
        return "".join(out)
    ci = rng.random() < 0.25
    kind = rng.choice(["prefix", "prefix", "suffix", "both", "middle", "gap"])
    k = rng.randrange(1, 5)
    body = a[head:]
    if kind == "prefix":
        pat = "^" + re.escape(a[:head]) + soften(body[:k])
    elif kind == "suffix":
        pat = soften(a[-k:]) + "$"
    elif kind == "both":
        pat = "^" + re.escape(a[:head]) + soften(body[:k]) + ".*" + soften(a[-rng.randrange(1, 3):]) + "$"
    elif kind == "middle":
        i = rng.randrange(head, len(a) - k)
        pat = soften(a[i:i + k])
    else:
        g = rng.randrange(0, 4)
        pat = "^" + re.escape(a[:head]) + ".{%d}" % g + soften(body[g:g + k])
    if ci:
        pat = "".join(ch.swapcase() if ch.isalpha() and rng.random() < 0.5 and fmt != 5 else ch for ch in pat) if "\\" not in pat and "(?" not in pat else pat
    return pat, ci


@pytest.mark.parametrize("fmt", [0, 1, 2, 3, 4, 5])
def test_generalised_patterns_never_lose_a_match_to_the_device_test(core, fmt):
    """The failure a vanity scanner must not have is the silent one: a device test that rejects an address the pattern
    accepts.  50 patterns per format (VGEN_PATTERN_WALK: more by hand) grown from real addresses (classes, dots, alternatives, optional characters, gaps,
    either case) — each accepts at least the address it grew from —: exact DFA == oracle regex on every address, and the
    device test (hash160 ranges / bit masks / checksum masks, or 'pass everything' ahead of the on-device automaton)
    accepts whatever the DFA accepts."""
    rng = random.Random(900 + fmt)
    pb = 32 if fmt == 3 else 20
    payloads = [bytes(rng.randrange(256) for _ in range(pb)) for _ in range(700)]
    payloads += [bytes(k) + bytes(rng.randrange(256) for _ in range(pb - k)) for k in (1, 2, 3) for _ in range(30)]
    addrs = [address(core, fmt, p) for p in payloads]
    head = {0: 1, 4: 1, 2: 1, 1: 4, 3: 4, 5: 2}[fmt]
    alphabet = {0: "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz", 1: "qpzry9x8gf2tvdw0s3jn54khce6mua7l",
                5: "0123456789abcdefABCDEF"}
    alphabet = alphabet.get(fmt, alphabet[0] if fmt in (2, 4) else alphabet[1])
    kinds = {}
    blob = b"".join(payloads)
    core.core_dfa_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    for n in range(int(os.environ.get("VGEN_PATTERN_WALK", "50"))):
        a = rng.choice(addrs)
        pat, ci = generalise(rng, a, head, alphabet, fmt)
        kind, sel, dev, exact = check(core, pat, ci, fmt, payloads)
        kinds[kind] = kinds.get(kind, 0) + 1
        ore = vo.Regex(pat, ci)
        assert exact == [int(ore.matches(x)) for x in addrs], (pat, ci)
        assert exact[addrs.index(a)] == 1, (pat, ci, a)      # it accepts the address it grew from
        for d, e, x in zip(dev, exact, addrs):
            assert d or not e, f"device test rejected a real match: {pat!r} ci={ci} {x}"
        if kind == 4:
            # the on-device automaton over the encoded address (core/dfa_eval.h, run on the host): exactly the DFA's verdict
            # (Ethereum: the case-folded language, a superset the host's exact DFA then confirms)
            flags = ctypes.create_string_buffer(len(payloads))
            assert core.core_dfa_check(pat.encode(), int(ci), fmt, blob, len(payloads), flags) == 4
            full = [b & 1 for b in flags.raw]
            if fmt == 5:
                assert all(f or not e for f, e in zip(full, exact)), pat
            else:
                assert full == exact, (pat, ci)
    assert len(kinds) >= 2, kinds    # the walk reached more than one kind of device test
