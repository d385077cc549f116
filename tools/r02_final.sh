#!/bin/bash
# round-2 closing run on the GPU box: full GPU suite, smoke, bench with defaults and with the driver's arguments
mkdir -p gpurun_out
(timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r02f_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02f_pytest.log; tail -3 gpurun_out/r02f_pytest.log)
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err; echo "bench exit $?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02f_bench_driver_args.json 2> gpurun_out/r02f_bench_driver_args.err; echo "bench(driver args) exit $?"
python - <<'PY'
import json
for f in ("gpurun_out/r02f_bench.json", "gpurun_out/r02f_bench_driver_args.json"):
    d = json.load(open(f))
    print(f, d["steps"], d["value"], d["sustained"]["value"], d["roofline"].get("shader_clock_mhz_timed_region"), d["cpu_baseline"]["value"])
    for o in d.get("other_configs", []):
        print("   ", o["config"][:60], o["value"], o.get("chip_frac"))
PY
