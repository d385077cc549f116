"""oracle/pattern_info.py — CPU restatement of the reference's pattern front-end heuristics.

TEST INFRASTRUCTURE ONLY (see oracle/vgen_oracle.h): imported by tests/, never by the product.

Restates, one character at a time exactly as the reference walks the pattern string,
  * Pattern::validate_charset      /root/reference/src/pattern.rs:49-177
  * count_fixed_chars              /root/reference/src/pattern.rs:269-293
  * Pattern::estimate_difficulty   /root/reference/src/pattern.rs:183-253
  * AddressFormat::charset_name    /root/reference/src/address.rs:39-45
Pinned by the reference's own unit tests for these functions (pattern.rs:357-452 difficulty,
pattern.rs:454-635 charset), held as data in tests/golden/pattern_frontend.json.
Formats use the C-ABI numbering: 0 P2PKH, 1 P2WPKH, 2 P2SH-P2WPKH, 3 P2TR, 4 P2PKH-uncompressed, 5 Ethereum.
"""

BASE58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
BECH32 = "023456789acdefghjklmnpqrstuvwxyz"
HEXSET = "0123456789abcdefABCDEFx"
META = "^$.*+?(){}|"


def charset_name(fmt):
    return {0: "Base58", 4: "Base58", 2: "Base58", 1: "Bech32", 3: "Bech32", 5: "Hex"}[fmt]


def _alphabet(fmt):
    return {"Base58": BASE58, "Bech32": BECH32, "Hex": HEXSET}[charset_name(fmt)]


def validate_charset(pattern, case_insensitive, fmt):
    valid = _alphabet(fmt)

    def ok(c):
        if case_insensitive:                                    # pattern.rs:71-78
            return c.lower() in valid or c.upper() in valid
        return c in valid

    invalid = []
    in_class = negated = class_start = escaped = pending = False
    members, prev = [], None
    for c in pattern:
        if escaped:                                             # pattern.rs:81-90
            escaped = False
            if in_class:
                class_start = False
                if c not in members:
                    members.append(c)
            continue
        if c == "\\":
            escaped = True
        elif c == "[":                                          # pattern.rs:95-102
            in_class, class_start, negated, members, prev, pending = True, True, False, [], None, False
        elif c == "]" and in_class:                             # pattern.rs:103-121
            if not negated and not any(ok(m) for m in members):
                for m in members:
                    if m not in invalid:
                        invalid.append(m)
            in_class, prev, pending = False, None, False
        elif c == "^" and in_class and class_start:             # pattern.rs:122-125
            negated, class_start = True, False
        elif c in META and not in_class:                        # pattern.rs:127-131
            class_start = False
        elif c == "-" and in_class:                             # pattern.rs:132-140
            class_start = False
            if prev is not None:
                pending = True
        elif c.isalnum():                                       # pattern.rs:141-170
            class_start = False
            if in_class:
                if pending:
                    lo, hi = min(prev, c), max(prev, c)
                    for v in range(ord(lo), ord(hi) + 1):
                        if chr(v) not in members:
                            members.append(chr(v))
                    pending = False
                elif c not in members:
                    members.append(c)
                prev = c
            elif not ok(c) and c not in invalid:
                invalid.append(c)
        else:                                                   # pattern.rs:171-176
            class_start = False
            if in_class and c not in members:
                members.append(c)
    return invalid


def count_fixed_chars(pattern):
    count, in_class, escaped = 0, False, False
    for c in pattern:
        if escaped:
            escaped = False
        elif c == "\\":
            escaped = True
        elif c == "[":
            in_class = True
        elif c == "]":
            in_class = False
        elif c in META:
            pass
        elif not in_class and c.isalnum():
            count += 1
    return count


def estimate_difficulty(pattern, case_insensitive, fmt):
    name = charset_name(fmt)
    alphabet = (34 if case_insensitive else 58) if name == "Base58" else 32 if name == "Bech32" else 16
    sub = 0
    if pattern.startswith("^"):                                 # pattern.rs:205-244
        rest = pattern[1:]
        if fmt in (0, 4):
            sub = int(rest.startswith("1"))
        elif fmt == 2:
            sub = int(rest.startswith("3"))
        elif fmt in (1, 3):
            full = "bc1q" if fmt == 1 else "bc1p"
            sub = 4 if rest.startswith(full) else 3 if rest.startswith("bc1") else 2 if rest.startswith("bc") else int(rest.startswith("b"))
        else:
            sub = 2 if rest[:2] in ("0x", "0X") else int(rest.startswith("0"))
    eff = max(0, count_fixed_chars(pattern) - sub)
    if eff == 0:
        return 1
    return min(alphabet ** eff, 2 ** 64 - 1)                    # saturating_pow
