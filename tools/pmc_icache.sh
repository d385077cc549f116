# Instruction-cache and wait counters of seq_bwd_kernel (one launch of 2^22 keys = four waves per SIMD), run on the GPU box from the repo root.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05ac}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VGEN_LONE_VARIANT=0
B="--batch 4194304 --steps 8 --warmup 2 --frames 1 --sustained-seconds 0 --no-other-configs --no-cpu-baseline --multi-leg-seconds 0"
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES -d $OUT/a -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/b -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("a", "b"):
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float); names = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (did, c), v in per.items():
            if "seq_bwd" in names[did]: acc["seq_bwd"][c].append(v)
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())}, "launches", max(len(v) for v in c.values()))
PY
tail -3 $OUT/a.log | cut -c1-200
