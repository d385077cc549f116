#!/bin/bash
# round-3 GPU call: full suite, default bench line (2048 steps) and driver-shaped line, counter passes incl. the random-stream mode,
# and the headline counter passes (pmc_valu) re-collected on this round's kernels
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $OUT/pytest.log; tail -12 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err || { echo "bench k20 failed"; exit 1; }
python - <<PY
import json
for f in ("$OUT/bench.json", "$OUT/bench_k20.json"):
    d = json.load(open(f))
    print(f.split("/")[-1], d["steps"], d["value"], "sustained", d["sustained"]["value"], "MHz", d["roofline"].get("shader_clock_mhz_timed_region"), "ttfm", d.get("time_to_first_match"))
for o in d.get("other_configs", []):
    print("   ", o["config"][:70], o.get("value"), o.get("chip_frac"), o.get("error"))
PY
bash tools/pmc_keys.sh $TAG/pmc "keys random p2tr" > $OUT/pmc_keys.log 2>&1
python - <<PY
import json
d = json.load(open("$OUT/pmc/pmc_keys.json"))
for m in ("keys", "random", "p2tr"):
    for k, e in d.get(m, {}).items():
        if "rocclr" in k: continue
        print(m, k[:34], e.get("lone_launch_us_under_pmc"), e.get("valu_instr_per_key"), e.get("valu_busy"), e.get("simd_cycles_per_valu_instr"))
print(d.get("failed_passes"))
PY
bash tools/pmc_valu.sh $TAG/pmc_valu > $OUT/pmc_valu.log 2>&1; tail -c 1500 $OUT/pmc_valu.log | head -40
