"""Turns the passes of tools/pmc_keys.sh into profiles/pmc_keys.json: per kernel of the scalar-multiplication paths the
issue-side counters (instructions per key, VALU-busy, wait fractions), HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE, the
gfx950 correction of MI355X_MICROARCH.md), L2 hit rate, and the kernel's lone-launch duration from the kernel trace of
the same run."""
import csv
import glob
import json
import os
import re
import statistics
import sys
from collections import defaultdict

KEYS = 1 << 20


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def counters(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per, names = defaultdict(float), {}
        with open(path) as f:
            for row in csv.DictReader(f):
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
        for (d, c), v in per.items():
            acc[names[d]][c].append(v)
    # the MEDIAN over the launches: the first one includes cold caches / table build effects, and now and then a launch is
    # held up for milliseconds by something else on the box
    return {k: {c: statistics.median(v) for c, v in cs.items()} for k, cs in acc.items()}


def durations(root):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: round(statistics.median(v), 2) for k, v in acc.items()}


def summarize(root, mode):
    sq, fetch, write, tcc, wait = (counters(os.path.join(root, f"{mode}_{p}")) for p in ("sq", "fetch", "write", "tcc", "wait"))
    dur = durations(os.path.join(root, f"{mode}_sq"))
    out = {}
    for k, c in sq.items():
        if "SQ_INSTS_VALU" not in c or k.startswith(("gen_table", "rtab_build", "clock_probe")):
            continue
        e = {"lone_launch_us_under_pmc": dur.get(k),
             "valu_instr_per_key": round(c["SQ_INSTS_VALU"] * 64 / KEYS, 1),
             "salu_instr_per_key": round(c.get("SQ_INSTS_SALU", 0) * 64 / KEYS, 1),
             "waves": int(c.get("SQ_WAVES", 0)),
             "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (c["SQ_BUSY_CYCLES"] * 32), 4) if c.get("SQ_BUSY_CYCLES") else None,
             "simd_cycles_per_valu_instr": round(c["SQ_BUSY_CYCLES"] * 32 / c["SQ_INSTS_VALU"], 2) if c.get("SQ_INSTS_VALU") else None,
             "wait_inst_any_frac_of_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None,
             "gui_active_cycles": c.get("GRBM_GUI_ACTIVE")}
        f = fetch.get(k, {}).get("FETCH_SIZE")
        w = write.get(k, {}).get("WRITE_SIZE")
        if f is not None:
            e["fetch_bytes_x2"] = int(f * 1024 * 2)
            e["fetch_bytes_per_key_x2"] = round(f * 1024 * 2 / KEYS, 1)
        if w is not None:
            e["write_bytes"] = int(w * 1024)
        t = tcc.get(k, {})
        if t.get("TCC_HIT_sum") is not None and t.get("TCC_MISS_sum") is not None and t["TCC_HIT_sum"] + t["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 4)
            e["l2_requests_per_key"] = round((t["TCC_HIT_sum"] + t["TCC_MISS_sum"]) / KEYS, 2)
        wv = wait.get(k, {})
        if wv.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac_of_wave_cycles"] = round(wv.get("SQ_WAIT_ANY", 0) / wv["SQ_WAVE_CYCLES"], 3)
            e["active_inst_any_frac_of_wave_cycles"] = round(wv.get("SQ_ACTIVE_INST_ANY", 0) / wv["SQ_WAVE_CYCLES"], 3)
            if wv.get("SQ_INSTS_VMEM_RD") is not None:
                e["vmem_rd_instr_per_key"] = round(wv["SQ_INSTS_VMEM_RD"] * 64 / KEYS, 2)
        if e["lone_launch_us_under_pmc"] and f is not None:
            e["hbm_side_gb_per_s"] = round((f * 1024 * 2 + (w or 0) * 1024) / (e["lone_launch_us_under_pmc"] * 1e-6) / 1e9, 1)
        out[k] = e
    return out


def main(root):
    res = {"keys_per_launch": KEYS,
           "how": "rocprofv3 --kernel-trace --pmc, one pass per counter group (tools/pmc_keys.sh), frames = 1, kernels serialised; "
                  "valu_busy = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x 32); bytes = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, "
                  "uncalibrated for 16-byte-per-lane gathers of 64-byte sectors (ratios between table widths are what to read)"}
    for mode in ("keys", "random", "random_endo", "p2tr", "keys16", "keys20", "keys22", "keys24", "keys26", "keys29", "p2tr29"):
        if os.path.isdir(os.path.join(root, f"{mode}_sq")):
            res[mode] = summarize(root, mode)
    if os.path.exists(os.path.join(root, "failed.txt")):
        res["failed_passes"] = open(os.path.join(root, "failed.txt")).read().split("\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
