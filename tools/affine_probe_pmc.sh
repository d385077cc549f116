# Counter passes for tools/affine_probe (run on the GPU box from the repo root): issue-side counters, FETCH_SIZE and WRITE_SIZE in
# passes of their own (rocprofv3 --kernel-trace --pmc only; the program directly after `--`), summarised per kernel.
# usage: bash tools/affine_probe_pmc.sh [tag]     -> gpurun_out/<tag>/affine_pmc.txt
TAG=${1:-affine_pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o p -- $GRAFT_REPO_ROOT/tools/affine_probe 20 > $OUT/$name.log 2>&1 \
    || echo "pass $name failed (see $name.log)" | tee -a $OUT/failed.txt
}
run sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - $OUT > $OUT/affine_pmc.txt <<'PY'
import csv, glob, os, re, statistics, sys
from collections import defaultdict
root = sys.argv[1]
def short(n):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n
def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        per, names = defaultdict(float), {}
        for row in csv.DictReader(open(path)):
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
        for (i, c), v in per.items():
            acc[names[i]][c].append(v)
    return {k: {c: statistics.median(v) for c, v in cs.items()} for k, cs in acc.items()}
def durations(d):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(root, d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: statistics.median(v) for k, v in acc.items()}
sq, fe, wr, du = counters("sq"), counters("fetch"), counters("write"), durations("sq")
N = 1 << 20
print("# per launch of 2^20 keys, median over the launches of the run (rocprofv3 serialises kernels under --pmc: every kernel alone on the chip)")
print("# bytes as tools/pmc_keys_summarize.py derives them: FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide")
print("# coalesced streaming reads (MI355X_MICROARCH.md: double it), while 64-byte gathers are counted in full — both readings are printed")
print(f"# {'kernel':34s} {'us alone':>9s} {'VALU instr/key':>15s} {'VALU busy':>10s} {'fetch B/key (x2)':>17s} {'fetch B/key (raw)':>18s} {'write B/key':>12s}")
for k in sorted(sq):
    c = sq[k]
    if "SQ_INSTS_VALU" not in c or k.startswith(("gen_table", "keys_kernel")):
        continue
    f = fe.get(k, {}).get("FETCH_SIZE", 0) * 1024
    w = wr.get(k, {}).get("WRITE_SIZE", 0) * 1024
    busy = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["SQ_BUSY_CYCLES"] * 32) if c.get("SQ_BUSY_CYCLES") else 0
    print(f"  {k:34s} {du.get(k, 0):9.1f} {c['SQ_INSTS_VALU'] * 64 / N:15.1f} {busy:10.3f} {2 * f / N:17.1f} {f / N:18.1f} {w / N:12.1f}")
PY
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/affine_pmc.txt
