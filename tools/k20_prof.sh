#!/bin/bash
# Kernel timeline of the 20-step window for the fused and the split form (tools/ramp_trace.py under rocprofv3 --kernel-trace; 400 dispatches of
# steady load before every window, as bench.py's sustained leg leaves the device)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for SP in 0 1; do
  export VGEN_SPLIT=$SP
  D=gpurun_out/${1:-r05d}/trace_split$SP
  mkdir -p $D
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -o ramp -- python3 tools/ramp_trace.py ${2:-12} 20 0 400 > $D/stdout.txt 2>&1
  F=$(find $D -name '*kernel_trace.csv' | head -1)
  echo "== VGEN_SPLIT=$SP"; grep "total" $D/stdout.txt
  python3 tools/k20_timeline.py $F 20
done
