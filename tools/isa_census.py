"""Instruction census of a kernel's ISA (cross-compiles kernels.hip to assembly; no GPU needed).
usage: python tools/isa_census.py [mangled-kernel-substring]   default: seq_bwd_kernel<P2PKH, prefilter>
Cycle estimates use the per-class issue costs measured by tools/ubench_valu (profiles/r01_ubench_valu.jsonl)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
        "v_lshlrev_b32", "v_mov_b32", "v_cndmask_b32", "v_cmp_gt_u32", "v_cmp_lt_u32", "v_cmp_eq_u32", "v_cmp_ne_u32",
        "v_cmp_le_u32", "v_cmp_ge_u32"}


def cost(op, mixed=True):
    """Issue cycles per wave-instruction.  Half-rate ops cost 4.1 (v_mad_u64_u32 4.35) in any stream.  Full-rate
    ops cost 2.2-2.4 only in a stream of their own kind; next to half-rate ops they cost ~3.4 (a 1:1 mix of
    v_add_u32 and v_alignbit_b32 measures 3.7-3.8 per instruction, 2:1 3.5-3.6: profiles/r01_ubench_valu.jsonl),
    which is what the path's code is, so that is the default."""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base == "v_mad_u64_u32":
        return 4.35
    if base == "v_bitop3_b32":
        return 3.4 if mixed else 2.44
    if base in FULL:
        return 3.4 if mixed else 2.25
    return 4.1


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "seq_bwd_kernelILi0ELb0ELb0ELb0ELb0"
    os.makedirs("/tmp/isa", exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include",
                           "-Wno-pass-failed", "--cuda-device-only", "-S", f"{ROOT}/vgen_amd/csrc/device/kernels.hip",
                           "-o", "/tmp/isa/kernels.s"])
    txt = open("/tmp/isa/kernels.s").read()
    sym = [m for m in re.findall(r"^(_Z\w+):", txt, re.M) if name in m][0]
    lines = txt.split(sym + ":")[1].split(".Lfunc_end")[0].split("\n")
    # split at loop headers
    marks = [0] + [i for i, l in enumerate(lines)
                   if re.match(r"^\.LBB\d+_\d+:", l) and "Loop Header" in " ".join(lines[i:i + 3])] + [len(lines)]
    print(sym)
    for lo, hi in zip(marks, marks[1:]):
        c = collections.Counter()
        for l in lines[lo:hi]:
            m = re.match(r"^\s+([vs]_\w+|global_\w+|ds_\w+|scratch_\w+|buffer_\w+)", l)
            if m:
                c[m.group(1)] += 1
        valu = {k: n for k, n in c.items() if k.startswith("v_")}
        cyc = sum(n * cost(k) for k, n in valu.items())
        hdr = lines[lo].strip()[:60] if lo else "(prologue)"
        print(f"\n== lines {lo}-{hi}  {hdr}\n   VALU {sum(valu.values())}  SALU {sum(n for k, n in c.items() if k.startswith('s_'))}"
              f"  mem {sum(n for k, n in c.items() if not k[0] in 'vs' or k.startswith('scratch'))}  est issue cycles {cyc:.0f}")
        print("   " + ", ".join(f"{k} {n}" for k, n in sorted(valu.items(), key=lambda x: -x[1])[:16]))


if __name__ == "__main__":
    main()
