"""Where do the register-capped kernels spill?  Cross-compiles kernels.hip to assembly and, for every kernel with
scratch traffic, counts the scratch loads / stores by loop depth (0 = outside loops, 1 = per denominator step,
2 = per key).  usage: python tools/spill_report.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = "/tmp/vgen_kernels.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                       "-Wno-pass-failed", "--cuda-device-only", "-S", os.path.join(ROOT, "vgen_amd/csrc/device/kernels.hip"), "-o", asm],
                      stderr=subprocess.DEVNULL)
name, depth, stats, order = None, 0, {}, []
for line in open(asm):
    m = re.match(r"^(_ZN2vg\w+):", line)
    if m:
        name, depth = m.group(1), 0
        stats[name] = {}
        order.append(name)
        continue
    if name is None:
        continue
    if line.startswith(".LBB"):
        d = re.search(r"Depth=(\d+)", line)
        if "Loop Header" in line and d:
            depth = int(d.group(1))
        elif "in Loop" in line and d:
            depth = int(d.group(1))
        elif "Parent Loop" in line:
            pass
        elif not d:
            depth = 0
    if "s_endpgm" in line:
        name = None
        continue
    m = re.search(r"\b(scratch_(?:load|store)_dword\w*)", line)
    if m:
        key = (depth, "load" if "load" in m.group(1) else "store")
        stats[name][key] = stats[name].get(key, 0) + 1
for n in order:
    if not stats[n]:
        continue
    short = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    parts = ", ".join(f"depth {d}: {c} {k}s" for (d, k), c in sorted(stats[n].items()))
    print(f"{short}: {parts}")
