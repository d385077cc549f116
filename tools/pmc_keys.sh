# Counter passes for the paths that do a scalar multiplication per key (run on the GPU box from the repo root):
#   keys  (vgen_dispatch_keys: keys_fwd_kernel / keys_bwd_kernel), p2tr (seq_bwd_kernel<P2TR> / p2tr_finish_kernel),
#   random (vgen_dispatch_random, when built), at the default 24-bit generator table and, for the gather question, at 16 / 20 / 22 / 26 bits.
# Each pass is its own rocprofv3 run (--pmc with --kernel-trace only; the program directly after `--`).
# Output: gpurun_out/<tag>/pmc_keys.json  (copy to profiles/pmc_keys.json, which bench.py reads)
TAG=${1:-pmc_keys}
MODES=${2:-"keys random random_endo p2tr"}   # all four: bench.py prices its entries from every one of them
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"
run() {   # name, counters..., then the driver's arguments after --
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc "${ctr[@]}" -d $OUT/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_driver.py "$@" > $OUT/$name.log 2>&1 \
    || echo "pass $name failed (see $name.log)" | tee -a $OUT/failed.txt
}
for m in $MODES; do
  run ${m}_sq $SQ -- $m
  run ${m}_fetch FETCH_SIZE -- $m
  run ${m}_write WRITE_SIZE -- $m
  run ${m}_tcc TCC_HIT_sum TCC_MISS_sum -- $m
  run ${m}_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU -- $m
done
# the gather question (2^20 distinct random scalars per launch: the random-stream mode): the same KEYS kernel over tables that fit the 256 MB Infinity Cache (16 bits: 67 MB) or not (20: 872 MB; 24: 11.8 GB)
for bits in 16 20 22 26; do
  export VGEN_GTAB_BITS=$bits
  run keys${bits}_sq $SQ -- random
  run keys${bits}_fetch FETCH_SIZE -- random
  run keys${bits}_tcc TCC_HIT_sum TCC_MISS_sum -- random
done
# the 29-bit signed-window table (8 additions; what long scans move to): the random-key path and the taproot path over it
export VGEN_GTAB_BITS=29
for m in keys29:random p2tr29:p2tr; do
  run ${m%%:*}_sq $SQ -- ${m##*:}
  run ${m%%:*}_fetch FETCH_SIZE -- ${m##*:}
  run ${m%%:*}_write WRITE_SIZE -- ${m##*:}
  run ${m%%:*}_tcc TCC_HIT_sum TCC_MISS_sum -- ${m##*:}
done
unset VGEN_GTAB_BITS
python3 $GRAFT_REPO_ROOT/tools/pmc_keys_summarize.py $OUT > $OUT/pmc_keys.json
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/pmc_keys.json
