# SQ counter passes over a single-frame bench run (kernels are serialised under --pmc anyway).
# usage: bash tools/pmc_run.sh <S> <outdir-under-gpurun_out>     (run on the GPU box from the repo root)
set -e
export VGEN_SEQ_S=${1:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --frames 1 --no-cpu-baseline ${BENCH_ARGS} > $OUT/pass$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT > $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
