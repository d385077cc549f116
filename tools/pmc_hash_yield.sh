# Counter passes over tools/ubench_hash_yield (the hash pair alone): what changes on the issue / fetch side when the block carries yields.
#   bash tools/pmc_hash_yield.sh <outdir under gpurun_out>          (run on the GPU box from the repo root, after tools/ubench_hash_yield_gen.sh)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU \
  -d $OUT/a -o p -- $GRAFT_REPO_ROOT/tools/ubench_hash_yield 256 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_CYCLES \
  -d $OUT/b -o p -- $GRAFT_REPO_ROOT/tools/ubench_hash_yield 256 > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("a", "b"):
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size"]) // 256
            acc[(r["Kernel_Name"], g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), c in sorted(acc.items(), key=lambda x: (x[0][1], x[0][0])):
    if g % 256 or g // 256 not in (2, 4, 8):
        continue
    # the timed launches are the ones with the most instructions
    row = {n: max(v) for n, v in c.items()}
    print(k.replace("void ", "").replace("(unsigned int*, int, int, unsigned long long*)", ""), "waves/SIMD", g // 256, {n: int(v) for n, v in sorted(row.items())})
PY
find $OUT -name "*.csv" -size +1M -delete
