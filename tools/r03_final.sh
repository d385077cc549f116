#!/bin/bash
# round-3 closing run on the GPU box: full GPU suite, smoke, then the round's evidence (tools/profile_round.sh: default and
# driver-argument bench lines, kernel traces of frames = 1 and of the default configuration, the headline counter passes)
mkdir -p gpurun_out
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03f_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03f_pytest.log; tail -4 gpurun_out/r03f_pytest.log)
grep -q "pytest exit 0" gpurun_out/r03f_pytest.log || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r03 || exit 1
python - <<'PY'
import json
for f in ("gpurun_out/prof_r03/bench.json", "gpurun_out/prof_r03/bench_driver_args.json"):
    d = json.load(open(f))
    print(f, d["steps"], d["value"], d["sustained"]["value"], d["roofline"].get("shader_clock_mhz_timed_region"), d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("cpu_model"))
    for o in d.get("other_configs", []):
        print("   ", o["config"][:70], o.get("value"), o.get("chip_frac"), o.get("error"))
PY
cat gpurun_out/prof_r03/kernel_stats.csv | cut -c1-150
