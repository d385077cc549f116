"""VALU instruction census of the loops of tools/ubench_hash_order.hip (cross-compiles to assembly; no GPU needed).
-> per kernel: VGPRs, scratch, VALU instructions per loop iteration (= per hash pair x chains), by class.
usage: python tools/hash_order_census.py [> profiles/r04_hash_order_census.txt]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = ("v_alignbit_b32", "v_add3_u32", "v_perm_b32", "v_lshl_or_b32", "v_lshl_add_u32", "v_and_or_b32", "v_or3_b32", "v_xad_u32", "v_bfe_u32")


def main():
    os.makedirs("/tmp/isa", exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_hash_order.py")], stdout=subprocess.DEVNULL)   # (the .inc is generated)
    out = "/tmp/isa/ubench_hash_order.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "--cuda-device-only", "-S",
                           os.path.join(ROOT, "tools", "ubench_hash_order.hip"), "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.M | re.S):
        sym, meta = m.group(1), m.group(2)
        vg = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        sc = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1))
        body = txt.split("\n" + sym + ":", 1)[1].split(".Lfunc_end", 1)[0].split("\n")
        # the timed loop: the largest span between a label and a backward branch to it
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        best = (0, 0)
        for i, l in enumerate(body):
            b = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if b and b.group(1) in labels and labels[b.group(1)] < i and i - labels[b.group(1)] > best[1] - best[0]:
                best = (labels[b.group(1)], i)
        c = collections.Counter()
        for l in body[best[0]:best[1]]:
            mm = re.match(r"^\s+(v_\w+)", l)
            if mm:
                c[re.sub(r"_(e32|e64|sdwa|dpp)$", "", mm.group(1))] += 1
        n = sum(c.values())
        half = sum(v for k, v in c.items() if k in HALF)
        chains = 2 if "ILi2E" in sym else 1
        name = "compiler" if "k_compiler" in sym else {"0": "asm_natural", "1": "asm_grouped", "2": "asm_x2", "3": "asm_x2_grouped"}[re.search(r"ILi\dELi(\d)E", sym).group(1)]
        print(f"{name:16s} vgpr {vg:3d} scratch {sc:3d}  VALU per loop iteration {n:5d} = {n / chains:7.1f} per hash pair  half-rate {half / n:.3f}  "
              + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda x: -x[1])[:8]))


if __name__ == "__main__":
    main()
