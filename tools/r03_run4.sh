#!/bin/bash
# random-stream mode after the fill-kernel change: parity tests of that mode, its counter passes, its bench leg
TAG=${1:-r03d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "random or width or keys or cli" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
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
cp $OUT/pmc/pmc_keys.json profiles/pmc_keys.json
timeout -k 10 300 python bench.py --steps 256 --warmup 16 --sustained-seconds 0.5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
for o in d.get("other_configs", []):
    if "scalar" in o["config"] or "random" in o["config"] or "P2TR" in o["config"]:
        print("   ", o["config"][:80], o.get("value"), o.get("chip_frac"), (o.get("roofline") or {}).get("valu_instr_per_key"))
PY
