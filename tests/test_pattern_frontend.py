"""Pattern front-end: validate_charset / estimate_difficulty / charset_name of the product
(vgen_amd/csrc/host/pattern_info.cpp through the C ABI) against the reference's own unit-test cases
(tests/golden/pattern_frontend.json <- src/pattern.rs:357-635, src/address.rs:268-275) and, on random pattern
strings, against the character-by-character restatement in oracle/pattern_info.py."""
import ctypes
import json
import os
import random

import pytest

import vgen_amd as vg
from vgen_amd import api
from oracle import pattern_info as po

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "pattern_frontend.json")))


def invalid_chars(pattern, ci, fmt):
    buf, n = ctypes.create_string_buffer(300), ctypes.c_size_t()
    assert api._L.vgen_pattern_invalid_chars(pattern.encode(), int(ci), fmt, buf, 300, ctypes.byref(n)) == 0
    assert n.value == len(buf.value)
    return list(buf.value.decode())


def difficulty(pattern, ci, fmt):
    d = ctypes.c_uint64()
    assert api._L.vgen_pattern_difficulty(pattern.encode(), int(ci), fmt, ctypes.byref(d)) == 0
    return d.value


def test_oracle_restatement_on_the_reference_cases():
    for pat, want in GOLD["fixed_chars"]:
        assert po.count_fixed_chars(pat) == want
    for c in GOLD["difficulty"]:
        assert po.estimate_difficulty(c["pattern"], c["ci"], c["fmt"]) == c["base"] ** c["exp"], c
    for c in GOLD["charset"]:
        got = po.validate_charset(c["pattern"], False, c["fmt"])
        if "exact" in c:
            assert got == c["exact"], c
        else:
            assert set(c["set"]) <= set(got) and len(got) == c.get("len", len(got)), c
    for fmt, name in GOLD["charset_name"]:
        assert po.charset_name(fmt) == name


def test_product_on_the_reference_cases():
    for c in GOLD["difficulty"]:
        p = vg.Pattern(c["pattern"], c["ci"], vg.AddressFormat(c["fmt"]))
        assert p.estimate_difficulty() == c["base"] ** c["exp"], c
        assert p.original == c["pattern"] and p.is_case_insensitive() == c["ci"]      # pattern.rs:418-430
    for c in GOLD["charset"]:
        p = vg.Pattern(c["pattern"], False, vg.AddressFormat(c["fmt"]))
        got = p.validate_charset()
        if "exact" in c:
            assert got == c["exact"], c
        else:
            assert set(c["set"]) <= set(got) and len(got) == c.get("len", len(got)), c
    for fmt, name in GOLD["charset_name"]:
        assert vg.AddressFormat(fmt).charset_name() == name
    # the format argument is independent of the one the device prefilter was compiled for
    p = vg.Pattern("^bc1qab", False, vg.AddressFormat.P2wpkh)
    assert p.estimate_difficulty(vg.AddressFormat.P2tr) == 32 ** 3 and p.validate_charset(vg.AddressFormat.P2pkh) == []


def test_product_equals_restatement_on_random_patterns():
    """Not necessarily valid regexes: both functions are plain string walks that never compile the pattern."""
    rng = random.Random(2024)
    atoms = list("01OIlbioxXAaFfGgZz9qp") + list("^$.*+?(){}|[]-\\_ ,:") + ["[^", "\\.", "\\]", "\\\\", "[a-z]", "[0-9", "-]",
                                                                           "bc1q", "bc1p", "0x", "0X", "{10}", "(?i)"]
    n = 0
    for _ in range(6000):
        pat = "".join(rng.choice(atoms) for _ in range(rng.randrange(1, 12)))
        if rng.random() < 0.5:
            pat = "^" + pat
        for fmt in range(6):
            for ci in (False, True):
                assert invalid_chars(pat, ci, fmt) == po.validate_charset(pat, ci, fmt), (pat, ci, fmt)
                assert difficulty(pat, ci, fmt) == po.estimate_difficulty(pat, ci, fmt), (pat, ci, fmt)
                n += 1
    assert n == 72000


def test_edge_cases():
    assert difficulty("a" * 40, False, 0) == 2 ** 64 - 1                      # saturating_pow
    assert difficulty("a" * 16, False, 5) == 2 ** 64 - 1 and difficulty("a" * 15, False, 5) == 16 ** 15
    assert difficulty("^0Xab", False, 5) == 256 and difficulty("^0ab", False, 5) == 256
    assert difficulty("^3ab", False, 2) == 58 ** 2 and difficulty("^1ab", False, 2) == 58 ** 3
    assert difficulty("^bc1pzz", False, 3) == 32 ** 2 and difficulty("^bc1qzz", False, 3) == 32 ** 3
    assert difficulty("^1a{3}", False, 0) == 58 ** 2                          # quantifier digits count (reference quirk)
    assert invalid_chars("^1a{10}", False, 0) == ["0"]                        # ... and are charset-checked too
    assert invalid_chars("^1\\O", False, 0) == []                             # escapes are skipped outside classes
    assert invalid_chars("^1o", True, 0) == [] and invalid_chars("^1O", True, 0) == []   # 'o' is Base58 when folding
    assert invalid_chars("^1l0", True, 0) == ["0"]                            # 'L' is Base58, zero never is
    # the hrp/separator characters 'b' and '1' are not in the Bech32 data alphabet: flagged even in "^bc1q" (reference quirk)
    assert invalid_chars("^bc1qB", True, 1) == ["b", "1", "B"] and invalid_chars("^bc1qa", False, 1) == ["b", "1"]
    assert invalid_chars("[9-A]", False, 1) == []                             # 9 is Bech32
    assert invalid_chars("[:-A]", False, 1) == [":", "A"]                     # ':' has no range start: literal; 'A' alone
    assert invalid_chars("[a-.z]x", False, 5) == []                           # range survives the dot; a..f are hex
    assert invalid_chars("[G-Z][g-w]", False, 5) == list("GHIJKLMNOPQRSTUVWXYZ") + list("ghijklmnopqrstuvw")
    # errors: unknown format, NULL arguments
    d, n = ctypes.c_uint64(), ctypes.c_size_t()
    assert api._L.vgen_pattern_difficulty(b"a", 0, 9, ctypes.byref(d)) < 0
    assert api._L.vgen_pattern_invalid_chars(b"a", 0, 9, None, 0, ctypes.byref(n)) < 0
    assert api._L.vgen_pattern_invalid_chars(b"O0", 0, 0, None, 0, ctypes.byref(n)) == 0 and n.value == 2
    small = ctypes.create_string_buffer(2)
    assert api._L.vgen_pattern_invalid_chars(b"O0", 0, 0, small, 2, ctypes.byref(n)) == 0 and small.value == b"O" and n.value == 2
    assert api._L.vgen_format_charset_name(17) is None
