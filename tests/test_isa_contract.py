"""The ISA contract of the device code, checked on the assembly hipcc emits for gfx950 (no GPU needed).

What DESIGN.md and BASELINE.json's north_star say about the kernels is asserted here on `build/lib/device/kernels.s`
(made by vgen_amd/csrc/Makefile with the flags of the shipped object), so that a toolchain or source change that breaks
one of them fails the CPU suite instead of silently costing throughput:

  * no MFMA anywhere (integer / modular arithmetic, not a contraction);
  * the headline kernel seq_bwd_kernel<P2PKH, prefilter> fits 128 VGPRs (four waves per SIMD) with no scratch and the
    9 KB product tree as its only LDS;
  * match compaction is per WAVE: one global atomic per kernel, fed by the popcount of the wave's ballot, the leader's
    result broadcast by v_readlane, ranks by v_mbcnt (kernels.hip: match_slot) — never a per-lane atomic, never system scope;
  * the short first-half kernels of a dispatch raise their issue priority (s_setprio);
  * the hash pair of the headline kernel is the scheduled block device/hashgen.py writes — its instructions in the
    generator's order, one `s_nop 0` yield after every third — not a schedule of hipcc's;
  * no instantiation uses more registers or scratch than the committed table profiles/r05_kernel_resources.txt.

Reference counterpart of the code under test: src/shaders/search.wgsl:2-31 (one storage write per key, no compaction).
"""
import os
import re
import subprocess

import pytest

from conftest import locked_make

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "build", "lib", "device", "kernels.s")
SRC = os.path.join(ROOT, "vgen_amd", "csrc", "device", "kernels.hip")
HEADLINE = "_ZN2vg14seq_bwd_kernelILi0ELb0ELb0ELb0ELb0EEEvNS_7SeqArgsE"
LONE = "_ZN2vg14seq_bwd_kernelILi0ELb0ELb0ELb1ELb0EEEvNS_7SeqArgsE"   # its twin for contexts with one frame in flight


def parse_isa(txt):
    """-> {symbol: {"body": [lines], "vgpr": n, "scratch": n, "lds": n}} for every kernel of the file."""
    out = {}
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.M | re.S):
        sym, meta = m.group(1), m.group(2)

        def field(name, meta=meta):
            f = re.search(r"\." + name + r"\s+(\d+)", meta)
            return int(f.group(1)) if f else None
        body = txt.split("\n" + sym + ":", 1)[1].split(".Lfunc_end", 1)[0].split("\n")
        out[sym] = {"body": body, "vgpr": field("amdhsa_next_free_vgpr"), "scratch": field("amdhsa_private_segment_fixed_size"),
                    "lds": field("amdhsa_group_segment_fixed_size")}
    return out


def count(body, pat):
    return sum(1 for line in body if re.match(r"\s+" + pat, line))


@pytest.fixture(scope="module")
def isa():
    # (a no-op when __graft_entry__.build() has just run; ~1 min of hipcc otherwise)
    locked_make("-s", "-C", os.path.join(ROOT, "vgen_amd", "csrc"), "../../build/lib/device/kernels.s")
    return parse_isa(open(ISA).read())


def check_match_path(sym, k):
    """One wave-aggregated atomic: see the module docstring.  -> list of violations."""
    bad = []
    atom = [i for i, line in enumerate(k["body"]) if re.match(r"\s+(global|flat|buffer)_atomic", line)]
    if len(atom) != 1:
        return [f"{sym}: {len(atom)} atomics on the match path, expected exactly 1"]
    line = k["body"][atom[0]]
    if "global_atomic_add" not in line:
        bad.append(f"{sym}: the slot counter is not a global_atomic_add: {line.strip()}")
    if re.search(r"\bsc1\b", line):
        bad.append(f"{sym}: system-scope atomic (sc1) on the match path: {line.strip()}")
    if count(k["body"], "s_bcnt1_i32_b64") < 1:
        bad.append(f"{sym}: no s_bcnt1 — the atomic's addend is not the popcount of the wave's ballot")
    if count(k["body"], "v_mbcnt_hi_u32_b32") < 1:
        bad.append(f"{sym}: no v_mbcnt — candidates do not take ranks within the wave")
    # the reserved base comes back to every candidate lane from the leader (first set bit of the ballot): s_ff1 + v_readlane
    # in the lines after the atomic — the part only the explicit compaction has (LLVM's own atomic optimizer reads lane 0
    # of the active set with v_readfirstlane instead)
    tail = k["body"][atom[0]:atom[0] + 40]
    if not any(re.match(r"\s+s_ff1_i32_b64", t) for t in tail) or not any(re.match(r"\s+v_readlane_b32", t) for t in tail):
        bad.append(f"{sym}: no s_ff1_i32_b64 + v_readlane_b32 after the atomic — the leader's slot base is not broadcast")
    # and the atomic must not sit inside a loop of its own (a per-lane "waterfall")
    head = k["body"][max(0, atom[0] - 12):atom[0]]
    if any("s_cbranch_execnz" in t for t in head):
        bad.append(f"{sym}: the atomic sits behind an exec loop")
    return bad


def test_no_mfma_anywhere(isa):
    assert len(isa) >= 45, "kernels went missing from the assembly"
    for sym, k in isa.items():
        assert count(k["body"], "v_mfma") == 0 and count(k["body"], "v_smfmac") == 0, sym


def test_headline_kernel_budget(isa):
    k = isa[HEADLINE]
    assert k["vgpr"] <= 128, k["vgpr"]              # four waves per SIMD: four launches of different frames share one
    assert k["scratch"] == 0, k["scratch"]
    assert k["lds"] == 9 * 256 * 4, k["lds"]        # the workgroup's product tree and nothing else
    assert count(k["body"], "scratch_") == 0 and count(k["body"], "buffer_(load|store)") == 0
    # the instruction classes the path is made of: 64-bit multiply-adds, rotates, three-input booleans
    assert count(k["body"], "v_mad_u64_u32") > 500 and count(k["body"], "v_alignbit_b32") > 700 and count(k["body"], "v_bitop3_b32") > 400


def test_match_compaction_is_one_atomic_per_wave(isa):
    with_match = [s for s in isa if re.search(r"(seq_bwd_kernelILi[0245]|keys_bwd_kernelILi[0245]|p2tr_finish_kernel|p2tr_out_kernel|seq_hash_kernel)", s)
                  and not re.search(r"seq_bwd_kernelILi\dELb\dELb\dELb\dELb1E", s)]   # (the SPLIT form of seq_bwd_kernel only parks points: seq_hash_kernel reports)
    assert len(with_match) == 16 + 1 + 16 + 2 + 2 + 4, sorted(with_match)   # + 1: the one-frame twin of the headline kernel; + 4: seq_hash_kernel (A/B switch VGEN_SPLIT)
    bad = []
    for sym in with_match:
        bad += check_match_path(sym, isa[sym])
    assert not bad, "\n".join(bad)
    # kernels that only park intermediate results have no atomic at all
    for sym, k in isa.items():
        if sym not in with_match:
            assert count(k["body"], "(global|flat|buffer)_atomic") == 0, sym


def test_match_compaction_is_written_in_the_source():
    src = open(SRC).read()
    body = src.split("u32 match_slot(", 1)[1].split("\n}\n", 1)[0]
    assert "__ballot(hit)" in body and "__popcll(m)" in body and "mbcnt" in body and "readlane" in body
    assert body.count("atomicAdd(") == 1
    # every match site goes through it; nothing else touches the ring's counter
    rest = src.replace(body, "")
    code = "\n".join(line.split("//")[0] for line in rest.split("\n"))
    assert "atomicAdd" not in code and "__hip_atomic" not in code and "atomic_fetch" not in code
    assert code.count("match_slot(args.mhdr, hit, args.match_base)") == 7


def test_the_checker_rejects_a_per_lane_or_system_scope_atomic():
    """The checker itself: fed the shapes a regression would have, it must complain."""
    good = ["\ts_bcnt1_i32_b64 s52, vcc", "\tv_mbcnt_hi_u32_b32 v1, vcc_hi, v1", "\tglobal_atomic_add v8, v13, v8, s[28:29] sc0",
            "\ts_ff1_i32_b64 s2, vcc", "\tv_readlane_b32 s2, v3, s2"]
    assert check_match_path("k", {"body": good}) == []
    # per-lane atomic with system scope, as the compiler emits it when nothing aggregates
    assert check_match_path("k", {"body": ["\tglobal_atomic_add v8, v13, v8, s[28:29] sc0 sc1"]})
    # what LLVM's optimizer makes of a per-lane agent-scope atomicAdd(…, 1): aggregated, but not by the source
    llvm = ["\tv_mbcnt_lo_u32_b32 v3, s50, 0", "\tv_mbcnt_hi_u32_b32 v3, s51, v3", "\ts_bcnt1_i32_b64 s2, s[50:51]",
            "\tglobal_atomic_add v8, v13, v8, s[28:29] sc0", "\tv_readfirstlane_b32 s2, v8"]
    assert check_match_path("k", {"body": llvm})
    assert check_match_path("k", {"body": good + ["\tglobal_atomic_add v8, v13, v8, s[28:29] sc0"]})     # two atomics
    assert check_match_path("k", {"body": ["\ts_cbranch_execnz .LBB0_1"] + good})                         # waterfall loop


def test_the_hash_pair_is_the_generated_block_in_the_generators_order(isa):
    """The order of the hash instructions — runs of half-rate and of full-rate instructions — and the wave-priority changes between the
    runs are design decisions measured on the MI355X (profiles/r05_prio_ab.txt): the assembly must contain the generator's list verbatim,
    modulo register names."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "vgen_amd", "csrc", "device"))
    import hashgen as g
    p, _, _ = g.prog_pub33_h160()
    g.by_class(p, g.DEFAULT_CLASS_WINDOW)
    reg, _ = g.allocate(p)
    prio = tuple(int(x) for x in g.DEFAULT_PRIO.split(":"))
    want = [re.sub(r"%\[\w+\]", "R", l) for l in g.asm_lines(p, reg, g.DEFAULT_YIELD, prio)[0]]
    norm = []
    for l in isa[HEADLINE]["body"]:
        l = l.split(";")[0].strip()
        if re.match(r"(v_|s_nop|s_mov_b32|s_setprio)", l):
            norm.append(re.sub(r"\b[vs]\d+\b", "R", l))
    first = next(i for i, l in enumerate(norm) if l == want[0] and norm[i:i + 8] == want[:8])
    got = norm[first:first + len(want)]
    assert got == want
    assert sum(l.startswith("v_") for l in want) == 2196 and "s_nop 0" not in want
    # every half-rate run at priority 1, every full-rate run at 0, and the block leaves at the kernel's own level
    level = None
    for l in want:
        if l.startswith("s_setprio"):
            level = int(l.split()[1])
        elif l.startswith("v_"):
            half = l.startswith(("v_alignbit_b32", "v_add3_u32", "v_perm_b32"))
            assert level == (1 if half else 0), l
    assert want[-1] == "s_setprio 1" and 500 < sum(l.startswith("s_setprio") for l in want) < 800
    # the kernel itself runs at priority 1 from its first instructions on (the point arithmetic's multiply-adds take first places too)
    body = [l.split(";")[0].strip() for l in isa[HEADLINE]["body"]]
    first_valu = next(i for i, l in enumerate(body) if l.startswith("v_"))
    assert any(l == "s_setprio 1" for l in body[:first_valu + 40])
    # and the block appears once: the loop over a lane's keys is not unrolled around it
    assert count(isa[HEADLINE]["body"], r"v_alignbit_b32") < 2 * 868


def test_the_one_frame_twin_carries_no_yields(isa):
    """Contexts with one frame in flight launch seq_bwd_kernel<P2PKH, false, false, LONE>: hipcc's schedule of core/hash.h, because a
    wave that has its SIMD to itself pays four cycles per yield.  Same budget as the headline kernel."""
    k = isa[LONE]
    assert k["vgpr"] <= 128 and k["scratch"] == 0
    # one priority change in the whole kernel (the base level at its first instruction), none inside its hash code
    assert count(k["body"], r"s_nop") < 100 and count(k["body"], r"s_setprio") == 1 and count(isa[HEADLINE]["body"], r"s_setprio") > 500
    assert count(k["body"], r"v_alignbit_b32") >= 868


def test_chain_kernels_raise_their_priority(isa):
    for name in ("seq_fwd_kernel", "seq_inv_kernel"):
        sym = next(s for s in isa if name in s)
        body = isa[sym]["body"]
        first_valu = next(i for i, line in enumerate(body) if re.match(r"\s+v_", line))
        prio = [i for i, line in enumerate(body) if re.match(r"\s+s_setprio 3", line)]
        assert prio and prio[0] < first_valu + 40, (name, prio, first_valu)


def test_no_per_key_kernel_touches_scratch(isa):
    """Round 5: the instantiations that spilled 12 - 16 bytes per lane at their 128-register cap (the uncompressed-key format, the
    six-image on-device matcher) keep the running inverse in LDS instead (kernels.hip: PARKI); since then no kernel on the per-key
    path has a private segment.  Only the two generator-table builders (run once per table) keep theirs."""
    for sym, k in isa.items():
        if "gen_table_combine" in sym or "gen_combine_signed" in sym:
            continue
        assert k["scratch"] == 0, f"{sym}: {k['scratch']} B scratch"
        assert count(k["body"], "scratch_") == 0, sym


def test_register_and_scratch_budgets_against_the_committed_table(isa):
    table = {}
    for line in open(os.path.join(ROOT, "profiles", "r05_kernel_resources.txt")):
        m = re.match(r"(\S+)\tVGPRs: (\d+)\tScratchSize: (\d+)", line)
        if m:
            table[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    assert len(table) >= 45
    for sym, k in isa.items():
        short = sym.replace("_ZN2vg", "", 1)
        assert short in table, f"{sym}: not in profiles/r05_kernel_resources.txt (tools/kernel_resources.py regenerates it)"
        v, s = table[short]
        assert k["vgpr"] <= v, f"{sym}: {k['vgpr']} VGPRs, committed {v}"
        assert k["scratch"] <= s, f"{sym}: {k['scratch']} B scratch, committed {s}"
