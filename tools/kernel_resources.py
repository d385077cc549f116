"""VGPRs / scratch bytes per lane / occupancy of every kernel instantiation, from the assembly the Makefile emits
(build/lib/device/kernels.s; no GPU needed).  usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
tests/test_isa_contract.py holds every later build to the committed table."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "vgen_amd", "csrc"), "../../build/lib/device/kernels.s"])
    txt = open(os.path.join(ROOT, "build", "lib", "device", "kernels.s")).read()
    rnd = sys.argv[1] if len(sys.argv) > 1 else "4"
    print(f"# VGPRs / scratch bytes per lane / occupancy of every kernel instantiation (tools/kernel_resources.py, round {rnd})")
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.M | re.S):
        sym, meta = m.group(1), m.group(2)
        v = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        s = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1))
        # gfx950: 512 VGPRs per SIMD lane, allocated in blocks of 8, at most 8 waves per SIMD
        occ = min(8, 512 // (-(-v // 8) * 8))
        print(f"{sym.replace('_ZN2vg', '', 1)}\tVGPRs: {v}\tScratchSize: {s}\tOccupancy: {occ}")


if __name__ == "__main__":
    main()
