# Issue-side counters and HBM bytes of the scan kernels (run on the GPU box from the repo root):
#   pass A (2^22 keys per launch = 4 waves per SIMD from ONE launch, the occupancy the 16 overlapped frames of
#           the headline run reach; rocprofv3 serialises kernels under --pmc, so overlap itself cannot be profiled):
#           SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
#   pass B, C (2^20 keys per launch, the BASELINE dispatch): FETCH_SIZE, WRITE_SIZE — separate passes (TCC limits)
# Output: gpurun_out/<tag>/pmc_valu.json  (copy to profiles/pmc_valu.json, which bench.py reads)
TAG=${1:-pmc_r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VGEN_LONE_VARIANT=0   # the counters are of the steady-state kernel, not of the twin that frames = 1 contexts launch otherwise
B="--steps 16 --warmup 4 --frames 1 --sustained-seconds 0 --no-other-configs --no-cpu-baseline --multi-leg-seconds 0"   # (no in-process multi-device leg under the profiler: its launches would join the medians)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  -d $OUT/passA -o p -- python3 $GRAFT_REPO_ROOT/bench.py --batch 4194304 $B > $OUT/passA.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  -d $OUT/passA20 -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/passA20.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/passB -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/passB.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/passC -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B > $OUT/passC.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_valu_summarize.py $OUT > $OUT/pmc_valu.json
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/pmc_valu.json
