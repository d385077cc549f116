# Issue-side and instruction-cache counters of seq_bwd_kernel<ETHEREUM> for the in-tree library and variants (run on the GPU box):
#   bash tools/pmc_eth.sh <outdir> [tag ...]     one launch of 3 x 2^20 keys = 3 waves per SIMD (the kernel's occupancy)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cp $GRAFT_REPO_ROOT/vgen_amd/libvgen_hip.so /tmp/libA.so
cd /tmp && export TMPDIR=/tmp
B="--format ethereum --pattern ^0xdead --ci --batch 3145728 --steps 16 --warmup 4 --frames 1 --sustained-seconds 0 --no-other-configs --no-cpu-baseline --multi-leg-seconds 0"   # (no in-process multi-device leg under the profiler: its launches would join the medians)
for T in intree "$@"; do
  if [ $T != intree ]; then cp $GRAFT_REPO_ROOT/vgen_amd/libvgen_hip.so.$T $GRAFT_REPO_ROOT/vgen_amd/libvgen_hip.so; fi
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    -d $OUT/$T.a -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/$T.a.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH \
    -d $OUT/$T.b -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/$T.b.log 2>&1
  cp /tmp/libA.so $GRAFT_REPO_ROOT/vgen_amd/libvgen_hip.so
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*.[ab]")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "seq_bwd_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[-1], {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", max(len(v) for v in acc.values()) if acc else 0)
PY
find $OUT -name "*.csv" -size +1M -delete
