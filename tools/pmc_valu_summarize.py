"""Turns the counter passes of tools/pmc_valu.sh into profiles/pmc_valu.json (the object bench.py reads).

(Round 5: the figure called valu_busy is VALU instructions per 4-cycle issue slot; it exceeds 1 when slots carry two instructions.)
VALU-busy follows rocprof's derived metric VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / busy cycles: the counter
tallies VALU wave-instructions (it equals SQ_INSTS_VALU on gfx950) and the metric prices each at the 4 SIMD cycles
a wave64 instruction occupies at half rate; SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCD x 4), each
with 32 SIMDs, so busy SIMD cycles = SQ_BUSY_CYCLES x 32.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import csv  # noqa: E402
import glob  # noqa: E402
import re  # noqa: E402
from collections import defaultdict  # noqa: E402


def per_kernel(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch, names = defaultdict(float), {}
        with open(path) as f:
            for row in csv.DictReader(f):
                per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row.get("Grid_Size", 0) or 0))
        for (d, c), v in per_dispatch.items():
            k = re.sub(r"^.*?(\w+_kernel).*$", r"\1", names[d][0])
            acc[(k, names[d][1])][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in acc.items()}


def pick(stats, kernel, grid=None):
    best = None
    for (k, g), v in stats.items():
        if k == kernel and (grid is None or g == grid) and (best is None or v["launches"] > best["launches"]):
            best = v
    return best or {}


def main(root):
    n22, n20, S = 1 << 22, 1 << 20, 8
    a = per_kernel(os.path.join(root, "passA"))
    a20 = per_kernel(os.path.join(root, "passA20"))
    b = per_kernel(os.path.join(root, "passB"))
    c = per_kernel(os.path.join(root, "passC"))
    lanes22, lanes20 = n22 // (2 * S), n20 // (2 * S)
    bwd = pick(a, "seq_bwd_kernel", lanes22)
    fwd = pick(a, "seq_fwd_kernel", lanes22)
    bwd20 = pick(a20, "seq_bwd_kernel", lanes20)
    simd_cycles = bwd["SQ_BUSY_CYCLES"] * 32
    simd_cycles20 = bwd20["SQ_BUSY_CYCLES"] * 32
    fetch = pick(b, "seq_bwd_kernel", lanes20).get("FETCH_SIZE", 0.0)
    write = pick(c, "seq_bwd_kernel", lanes20).get("WRITE_SIZE", 0.0)
    out = {"p2pkh:%d" % n20: {
        "valu_busy": round(bwd["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, 4),
        "valu_busy_lone_2p20_launch": round(bwd20["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles20, 4),
        "simd_cycles_per_valu_instr": round(simd_cycles / bwd["SQ_INSTS_VALU"], 3),
        "valu_instr_per_key": round(bwd["SQ_INSTS_VALU"] * 64 / n22, 1),
        "valu_instr_per_key_all_kernels": round((bwd["SQ_INSTS_VALU"] + fwd.get("SQ_INSTS_VALU", 0.0)) * 64 / n22, 1),
        "salu_instr_per_key": round(bwd["SQ_INSTS_SALU"] * 64 / n22, 1),
        "wait_inst_any_frac_of_wave_cycles": round(bwd["SQ_WAIT_INST_ANY"] / bwd["SQ_WAVE_CYCLES"], 3),
        "waves_per_simd": round(bwd["SQ_WAVES"] / 1024, 2),
        "seq_bwd_kernel_hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
        "seq_bwd_kernel_algorithmic_bytes_per_launch": n20 // (2 * S) * 4 * (18 + 9 * S + 9) + 256 * 9 * 256 * 4,
        "raw": {"passA_2p22_seq_bwd": bwd, "passA_2p22_seq_fwd": fwd, "passA_2p20_seq_bwd": bwd20,
                "passB_2p20_FETCH_SIZE_KiB": fetch, "passC_2p20_WRITE_SIZE_KiB": write},
        "source": "profiles/pmc_valu.json <- tools/pmc_valu.sh (rocprofv3 --pmc, separate passes; kernels are serialised under --pmc)",
        "how": "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x 32 SIMDs per shader engine) of seq_bwd_kernel for a 2^22-key "
               "launch (4 waves per SIMD, the occupancy of the overlapped headline run) = VALU instructions per 4-cycle issue slot: above 1 "
               "since round 5, when full-rate instructions began to ride behind other waves' half-rate ones (rocprof's VALUBusy prices every "
               "instruction at 4 cycles); valu_instr_per_key = SQ_INSTS_VALU x 64 / keys; "
               "HBM bytes = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE of a 2^20-key launch",
    }}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
