#!/bin/bash
# parity of every table width, then build time (first P2TR match on a fresh context) and rates of the P2TR / KEYS paths by width
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generator_table or p2tr or keys_mode" 2>&1 | tail -3
for B in 16 20 22 24; do
  echo "== VGEN_GTAB_BITS=$B"
  VGEN_GTAB_BITS=$B VGEN_TRACE_CREATE=1 timeout -k 10 200 python tools/ttfm_formats.py 2>&1 | grep "P2tr" 
  VGEN_GTAB_BITS=$B timeout -k 10 200 python tools/gpu_perf_keys.py 2>&1 | tail -2
done
