#!/bin/bash
# The driver's --steps 20 window under different pipeline depths and idle gaps: tools/ramp_trace.py F K idle_s
set -e
mkdir -p gpurun_out
for F in 4 6 8 12 16; do
  for idle in 0.01 0; do
    echo "== F=$F idle=$idle"
    timeout -k 10 120 python tools/ramp_trace.py $F 20 $idle
  done
done > gpurun_out/ramp_sweep.txt 2>&1
grep "total" gpurun_out/ramp_sweep.txt | tail -40
