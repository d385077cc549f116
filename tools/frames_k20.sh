#!/bin/bash
# bench.py with the driver's arguments by pipeline depth
set -e
mkdir -p gpurun_out
cat > /tmp/show.py <<"PY"
import json, sys
d = json.load(open(sys.argv[1]))
print("frames", d["config"]["frames_in_flight"], "steps", d["steps"], "value", d["value"], "sustained", d["sustained"]["value"])
PY
{
for F in 12 16 20; do
  for i in 1 2 3; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --frames $F --no-cpu-baseline --no-other-configs 2>/dev/null > gpurun_out/fk.json
  python /tmp/show.py gpurun_out/fk.json
  done
done
} > gpurun_out/frames_k20.txt 2>&1
cat gpurun_out/frames_k20.txt
