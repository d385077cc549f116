# Issue-side counters of the endomorphism form of seq_bwd_kernel (run on the GPU box from the repo root): one launch of
# 2^22 curve points x 6 images (4 waves per SIMD), per-kernel averages -> gpurun_out/<tag>/summary.json
TAG=${1:-pmc_endo}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  -d $OUT/pass -o p -- python3 $GRAFT_REPO_ROOT/bench.py --endo --batch 4194304 --steps 8 --warmup 2 --frames 1 --sustained-seconds 0 --no-other-configs --no-cpu-baseline > $OUT/pass.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT > $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.json
