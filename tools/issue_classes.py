"""Instructions per key of seq_bwd_kernel<P2PKH> by ISSUE CLASS, from the assembly the Makefile emits (build/lib/device/kernels.s; no GPU needed).

The classes are those of the gfx950 issue stage as the probes tools/ubench_phase*.hip measured it (profiles/r05_phase*_ubench.jsonl) and
tools/issue_model.py restates it: a SIMD fills one 4-cycle slot with the next instruction of its highest-priority (then oldest) ready wave and,
behind it, ONE full-rate instruction of another wave.
  X  exclusive   v_mad_u64_u32, v_mul_lo / hi_u32, the 64-bit shifts and adds (v_lshl_add_u64, v_lshrrev_b64): the slot holds nothing else
  C  half rate   v_alignbit, v_add3, v_perm, v_bfe, v_lshlrev_b32, compares, carry-flag adds ...: first place only, a full-rate instruction may follow
  S  full rate   add / sub / logic / right shift / mov, v_bitop3 (in either encoding): either place
so a key needs at least  X + max(C, (C + S) / 2)  slots.  usage: python tools/issue_classes.py > profiles/r05_issue_classes.json"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYM = "_ZN2vg14seq_bwd_kernelILi0ELb0ELb0ELb0ELb0EEEvNS_7SeqArgsE:"
# Which place an opcode takes was measured opcode by opcode (tools/ubench_phase4.hip, profiles/r05_phase4_ubench.jsonl: alone, as the oldest wave
# beside three waves of v_add_u32, as the three younger waves beside one wave of v_alignbit_b32):
#   nothing rides behind   v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, v_lshl_add_u64, v_lshrrev_b64 (the 64-bit and full-width multiplier paths)
#   first place only       v_alignbit, v_add3, v_bfe, v_and_or, v_perm, v_mad_u32_u24, v_mul_u32_u24, v_lshl_add_u32, v_lshl_or, v_lshlrev_b32 (!),
#                          the compares and the carry-flag adds — a full-rate instruction rides behind each of them
#   either place           v_add / v_sub / v_xor / v_and / v_or / v_lshrrev_b32 / v_mov, v_bitop3, VOP3-encoded forms of these, v_fma_f32, v_mul_f32
SIMPLE = {"v_add_u32", "v_and_b32", "v_xor_b32", "v_or_b32", "v_lshrrev_b32", "v_mov_b32", "v_sub_u32", "v_subrev_u32",
          "v_bitop3_b32", "v_cndmask_b32", "v_not_b32", "v_ashrrev_i32"}
EXCLUSIVE = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_mov_b64"}


def classify(line):
    m = re.match(r"([a-z_0-9]+)", line)
    if not m:
        return None
    op = m.group(1)
    if op == "s_setprio":
        return "P"
    if op.startswith(("s_", "global_", "ds_", "buffer_", "flat_")):
        return "N"       # scalar / memory / control: a slot of the wave's own time, not of the vector issue
    if op.startswith("v_cmp"):
        return "C"
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base in EXCLUSIVE:
            return "X"
        if base in SIMPLE and not op.endswith("_sdwa"):
            return "S"
        return "C"
    return None


def stream(path):
    """-> (tokens of the per-pair part of the key loop, tokens of the per-key part: point arithmetic + hash block + the prefilter's fast path)."""
    txt = open(path).read()
    s = txt.index(SYM)
    b = [l.strip() for l in txt[s:txt.index("s_endpgm", s)].split("\n")]
    outer = [i for i, l in enumerate(b) if "=>This Loop Header: Depth=1" in l][-1]
    inner = [i for i, l in enumerate(b) if l.startswith(".LBB") and "Depth=2" in l][0]
    starts = [i for i, l in enumerate(b) if "ASMSTART" in l]
    ends = [i for i, l in enumerate(b) if "ASMEND" in l]
    hash_end = max(zip(starts, ends), key=lambda p: p[1] - p[0])[1]
    code = lambda lines: [t for t in (classify(l) for l in lines if l and not l.startswith((".", ";"))) if t]
    return code(b[outer:inner]), code(b[inner:hash_end + 30])


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "vgen_amd", "csrc"), "../../build/lib/device/kernels.s"])
    per_pair, per_key = stream(os.path.join(ROOT, "build", "lib", "device", "kernels.s"))
    count = {c: (per_pair.count(c) / 2.0 + per_key.count(c)) for c in "XCSNP"}
    valu = count["X"] + count["C"] + count["S"]
    bound = count["X"] + max(count["C"], (count["C"] + count["S"]) / 2.0)
    json.dump({"kernel": "seq_bwd_kernel<P2PKH, prefilter>", "per_key": {"exclusive_X": count["X"], "half_rate_C": count["C"], "full_rate_S": count["S"],
                                                                         "valu": valu, "scalar_memory_control_N": count["N"], "s_setprio": count["P"]},
               "issue_slots_per_key_at_least": bound, "valu_per_slot_at_most": round(valu / bound, 3),
               "how": "static census of the key loop of build/lib/device/kernels.s (per-pair part / 2 + per-key part incl. the generated hash block); "
                      "slots >= X + max(C, (C + S) / 2), one slot = 4 cycles of one SIMD (tools/issue_classes.py, tools/issue_model.py)"},
              sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
