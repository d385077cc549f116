"""The generator of the kernels' hash blocks (vgen_amd/csrc/device/hashgen.py), checked on the CPU.

The instruction lists the generator writes into the kernels as asm statements also run in Python: every hash of the path is
compared, before and after register allocation and in every order the generator offers, with hashlib and the oracle
(SHA-256 then RIPEMD-160: src/shaders/sha256.wgsl:43-170, src/shaders/ripemd160.wgsl:10-100).  The same blocks run on
the MI355X in every GPU parity test of the P2PKH / P2WPKH / P2SH-P2WPKH / uncompressed formats."""
import os
import random
import re
import struct
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vgen_amd", "csrc", "device"))
import hashgen as g  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

M = 0xFFFFFFFF


def words_le(x):
    return [(x >> (32 * i)) & M for i in range(8)]


def key_words(first, xw):
    return [(first << 24) | (xw[7] >> 8)] + [((xw[8 - i] << 24) & M) | (xw[7 - i] >> 8) for i in range(1, 8)]


def case_pub33(rng):
    x, prefix = rng.getrandbits(256), 2 + rng.getrandbits(1)
    xw = words_le(x)
    m = key_words(prefix, xw) + [((xw[0] << 24) & M) | 0x00800000]
    return m, po.hash160(bytes([prefix]) + x.to_bytes(32, "big"))


def case_script22(rng):
    h = rng.randbytes(20)
    b = [g.bswap(w) for w in struct.unpack("<5I", h)]
    m = [0x00140000 | (b[0] >> 16)] + [((b[i - 1] << 16) & M) | (b[i] >> 16) for i in range(1, 5)] + [((b[4] << 16) & M) | 0x8000]
    return m, po.hash160(b"\x00\x14" + h)


def case_pub65(rng):
    x, y = rng.getrandbits(256), rng.getrandbits(256)
    xw, yw = words_le(x), words_le(y)
    m = key_words(4, xw) + [((xw[0] << 24) & M) | (yw[7] >> 8)]
    m += [((yw[8 - i] << 24) & M) | (yw[7 - i] >> 8) for i in range(1, 8)] + [((yw[0] << 24) & M) | 0x00800000]
    return m, po.hash160(b"\x04" + x.to_bytes(32, "big") + y.to_bytes(32, "big"))


def case_base58_check(rng):
    version, h = rng.choice([0, 5]), rng.randbytes(20)
    H = struct.unpack(">5I", h)
    m = [(version << 24) | (H[0] >> 8)] + [((H[i - 1] << 24) & M) | (H[i] >> 8) for i in range(1, 5)] + [((H[4] << 24) & M) | 0x00800000]
    import hashlib
    return m, hashlib.sha256(hashlib.sha256(bytes([version]) + h).digest()).digest()[:4][::-1]   # out[0] is a big-endian word


def case_keccak_addr(rng):
    x, y = rng.getrandbits(256), rng.getrandbits(256)
    xw, yw = words_le(x), words_le(y)
    m = []
    for w in (xw, yw):
        for i in range(4):
            m += [g.bswap(w[7 - 2 * i]), g.bswap(w[6 - 2 * i])]
    return m, po.keccak256(x.to_bytes(32, "big") + y.to_bytes(32, "big"))[12:]


CASES = {"keccak_addr_block": case_keccak_addr, "base58_check_block": case_base58_check, "hash160_pub33_block": case_pub33, "hash160_script22_block": case_script22, "hash160_pub65_block": case_pub65}


def test_every_emitted_function_has_a_case():
    assert set(CASES) == set(g.PROGRAMS) | set(g.OPTIONAL)


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("grouped,window,distance,class_window", [(False, 0, 1, 0), (True, 0, 1, 0), (False, 8, 1, 0), (False, 16, 2, 0),
                                                                  (False, 0, 1, 1), (False, 0, 1, 4), (False, 0, 1, 40), (True, 0, 1, 8)])
def test_blocks_compute_the_hashes(name, grouped, window, distance, class_window):
    rng = random.Random(hash((name, grouped, window)) & 0xFFFF)
    p, _, _ = {**g.PROGRAMS, **g.OPTIONAL}[name](grouped)
    if window:
        g.spread(p, window, distance)
    if class_window:
        n = len(p.ins)
        runs = g.by_class(p, class_window)
        assert sum(runs) == n == len(p.ins)
    reg, nreg = g.allocate(p)
    if class_window <= 8:     # (wider windows are an option of the A/B; what ships is DEFAULT_CLASS_WINDOW)
        assert nreg <= (80 if "keccak" in name else 40)     # the pair fits beside the point arithmetic's registers
    assert sorted(reg[i] for i in p.inputs) == list(range(len(p.inputs)))
    for _ in range(12):
        m, want = CASES[name](rng)
        inputs = {f"m{i}": w for i, w in enumerate(m)}
        assert len(inputs) == len(p.inputs)
        for out in (g.evaluate(p, inputs), g.evaluate_allocated(p, reg, nreg, inputs)):
            assert b"".join(struct.pack("<I", w) for w in out) == want


def test_known_answers():
    # the generator key: hash160 of the compressed and of the uncompressed key of k = 1 (the addresses every wallet test knows)
    gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    p, _, _ = g.prog_pub33_h160()
    xw = words_le(gx)
    m = key_words(2, xw) + [((xw[0] << 24) & M) | 0x00800000]
    out = g.evaluate(p, {f"m{i}": w for i, w in enumerate(m)})
    assert b"".join(struct.pack("<I", w) for w in out).hex() == "751e76e8199196d454941c45d1b3a323f1433bd6"
    p, _, _ = g.prog_pub65_h160()
    yw = words_le(gy)
    m = key_words(4, xw) + [((xw[0] << 24) & M) | (yw[7] >> 8)]
    m += [((yw[8 - i] << 24) & M) | (yw[7 - i] >> 8) for i in range(1, 8)] + [((yw[0] << 24) & M) | 0x00800000]
    out = g.evaluate(p, {f"m{i}": w for i, w in enumerate(m)})
    assert b"".join(struct.pack("<I", w) for w in out).hex() == "91b24bf9f5288532960ac687abb035127b1d28a5"


def test_the_instruction_count_is_the_floor():
    # 64 SHA-256 rounds and 160 RIPEMD-160 steps with everything the padded 33-byte message fixes folded away; the count
    # EXPERIMENTS.md (the hash pair as a scheduled block) quotes (hipcc's own schedule of core/hash.h: 2 211)
    p, _, _ = g.prog_pub33_h160()
    c = p.census()
    assert sum(c.values()) == 2196
    assert c["alignbit"] == 868 and c["bitop3"] == 497 and c["add3"] == 370 and c["bswap"] == 8


def test_asm_text_shape():
    src = g.generate()
    for name in g.PROGRAMS:
        assert f"void {name}(" in src
    body = src[src.index("void hash160_pub33_block("):src.index("void hash160_script22_block(")]
    lines = re.findall(r'"([^"]*)\\n\\t"', body)
    valu = [l for l in lines if l.startswith("v_")]
    assert len(valu) == 2196
    # the shipped form (round 5): runs by issue class, a priority change at every class boundary, no yields
    assert "s_nop 0" not in lines
    level, changes = None, 0
    for a, b in zip(lines, lines[1:]):
        assert not (a.startswith("s_setprio") and b.startswith("s_setprio"))
    for l in lines:
        if l.startswith("s_setprio"):
            level, changes = int(l.split()[1]), changes + 1
        elif l.startswith("v_"):
            assert level == (1 if l.startswith(("v_alignbit_b32", "v_add3_u32", "v_perm_b32")) else 0), l
    assert lines[-1] == "s_setprio 1" and 500 < changes < 800
    # VOP3 instructions never carry a 32-bit literal on gfx9: constants come through the SGPR operand or are inline; the s_mov that
    # loads the SGPR sits directly in front of its reader
    for i, l in enumerate(lines):
        if l.startswith(("v_add3_u32", "v_bitop3_b32", "v_perm_b32", "v_alignbit_b32")):
            assert not re.search(r"0x[0-9a-f]{8}", l.split(" bitop3:")[0]), l
        if l.startswith("s_mov_b32"):
            assert "%[k]" in lines[i + 1], (l, lines[i + 1])
    assert '"=&s"(k)' in body and body.count('"+v"') == 9
    # the round-4 form, still generated for the A/B: dependency order, one yield per three VALU instructions, no priority changes
    old = g.generate(yields="every:3", prio=None, class_window=0)
    body = old[old.index("void hash160_pub33_block("):old.index("void hash160_script22_block(")]
    lines = re.findall(r'"([^"]*)\\n\\t"', body)
    assert sum(l == "s_nop 0" for l in lines) == 731 and not any(l.startswith("s_setprio") for l in lines)
    assert "s_nop" not in g.generate(yields="none")


def test_runs_by_issue_class_respect_the_dependencies_and_the_window():
    """by_class only ever moves an instruction in front of instructions it does not depend on, and never further than its window."""
    for window in (1, 4, 12):
        p, _, _ = g.prog_pub33_h160()
        orig = list(p.ins)
        index = {ins[1]: i for i, ins in enumerate(orig)}
        runs = g.by_class(p, window)
        seen = set(p.inputs)
        for pos, (op, d, srcs, imm) in enumerate(p.ins):
            assert all(g.known(x) or x in seen for x in srcs)
            seen.add(d)
        # an instruction is emitted no later than `window` places after everything older than it has gone
        done = set()
        for op, d, srcs, imm in p.ins:
            oldest_open = min(i for i in range(len(orig)) if orig[i][1] not in done)
            assert index[d] < oldest_open + window
            done.add(d)
        if window == 1:
            assert p.ins == orig
        else:
            assert len(runs) < 800 and max(runs) >= 8
