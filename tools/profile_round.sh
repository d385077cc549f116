# Regenerates the round's evidence under gpurun_out/prof_<tag>/ (run on the GPU box from the repo root):
#   bench.json                 default bench.py line (with cpu_baseline)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command (no cpu baseline)
#   pmc_summary.json           SQ / TCC counter passes, frames=1 (kernels are serialised under --pmc)
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --frames 1 --no-cpu-baseline > $OUT/pmc$i.log 2>&1 || exit 1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT > $OUT/pmc_summary.json
find $OUT -name "*.csv" -size +1M -delete
rm -rf $OUT/trace/*/*.db
