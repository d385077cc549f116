#!/bin/bash
# sustained rate by keys per dispatch (the reference's default is 2^19, BASELINE's 2^20)
set -e
mkdir -p gpurun_out
cat > /tmp/show.py <<"PY"
import json, sys
d = json.load(open(sys.argv[1]))
print("keys/dispatch", d["config"]["keys_per_dispatch"], "frames", d["config"]["frames_in_flight"], "value", d["value"], "sustained", d["sustained"]["value"])
PY
{
for B in 65536 131072 262144 524288 1048576 4194304; do
  for F in 12; do
  timeout -k 10 200 python bench.py --batch $B --frames $F --no-cpu-baseline --no-other-configs --sustained-seconds 2 2>/dev/null > gpurun_out/bs.json
  python /tmp/show.py gpurun_out/bs.json
  done
done
} > gpurun_out/batch_sweep.txt 2>&1
cat gpurun_out/batch_sweep.txt
