#!/bin/bash
# Kernel timeline of the 20-step window: rocprofv3 --kernel-trace around tools/ramp_trace.py (F K idle)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ramp_prof
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ramp_prof -o ramp -- python3 tools/ramp_trace.py ${1:-12} 20 0 ${2:-0} > gpurun_out/ramp_prof/stdout.txt 2>&1
ls -la gpurun_out/ramp_prof | tail -5
tail -3 gpurun_out/ramp_prof/stdout.txt | cut -c1-300
