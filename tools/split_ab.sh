# Split form (seq_bwd_kernel<.., SPLIT> + seq_hash_kernel) against the fused seq_bwd_kernel on one box, same library (VGEN_SPLIT=0 = fused):
# parity of both, sustained rate per number of frames in flight, and the driver's 20-step region.
# usage (on the GPU box): bash tools/split_ab.sh [frames list] > gpurun_out/.../split_ab.txt
FR=${1:-"1,2,3,4,6,8,12"}
export VGEN_PERF_STEPS=${VGEN_PERF_STEPS:-20000}
for F in 0 1 2; do
  for SP in 1 0; do
    echo "== parity fmt $F VGEN_SPLIT=$SP: $(VGEN_SPLIT=$SP python tests/manual/gpu_smoke.py $F 32768 2>&1 | grep -c 'mismatches 0 /') of 4 starts clean"
  done
done
for R in 1 2; do
  for SP in 1 0; do
    echo "== round $R VGEN_SPLIT=$SP"
    VGEN_SPLIT=$SP python tools/gpu_perf.py 0 $FR 2>&1 | grep Mkeys | sed "s/^S=[^ ]* WG=256 PREG=- //" | cut -c1-110
  done
done
for SP in 1 0 1 0; do
  echo "== bench --steps 20 --warmup 5, VGEN_SPLIT=$SP"
  VGEN_SPLIT=$SP python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value', d['value'], 'sustained', d['sustained']['value'], 'lone', d['roofline'].get('lone_launch',{}).get('avg_launch_ms'), 'ttfm', d.get('time_to_first_match'))
print('  region', {k:v for k,v in d['timing_region'].items() if k not in ('how',)})"
done
