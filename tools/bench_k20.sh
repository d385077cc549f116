#!/bin/bash
# bench.py with the driver's arguments (three runs) and with its defaults
set -e
mkdir -p gpurun_out
cat > /tmp/show.py <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("steps", d["steps"], "value", d["value"], "sustained", d["sustained"]["value"], "MHz sustained", r.get("shader_clock_mhz"),
      "MHz in timed region", r.get("shader_clock_mhz_timed_region"))
PY
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null > gpurun_out/bench_k20_$i.json
  python /tmp/show.py gpurun_out/bench_k20_$i.json
done
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null > gpurun_out/bench_default_quick.json
python /tmp/show.py gpurun_out/bench_default_quick.json
