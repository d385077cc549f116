#!/bin/bash
# keys-per-lane sweep: S (VGEN_SEQ_S: a lane tests 2S keys, so one 2^20-key launch is 8/S waves per SIMD) x frames in flight:
# the driver-shaped 20-step value, the sustained rate and the lone-launch duration of seq_bwd.   usage: bash tools/s_sweep.sh [outfile]
OUT=${1:-gpurun_out/s_sweep.txt}
echo "# S frames value_k20 sustained lone_launch_ms lone_frac region_mhz" > $OUT
for S in 2 4 8; do
  for F in 2 3 4 6 8 12; do
    for rep in 1 2; do
      VGEN_SEQ_S=$S python bench.py --steps 20 --warmup 5 --frames $F --no-other-configs --no-cpu-baseline --sustained-seconds 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; l=r.get('lone_launch',{})
print($S, $F, d['value'], d['sustained']['value'], l.get('avg_launch_ms'), l.get('frac'), r.get('shader_clock_mhz_timed_region'))" >> $OUT
    done
  done
done
cat $OUT
