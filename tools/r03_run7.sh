#!/bin/bash
# 24-bit default table: GPU suite, counter passes (default + 16/20/22/26), bench
TAG=${1:-r03i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $OUT/pytest.log; tail -6 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
bash tools/pmc_keys.sh $TAG/pmc "keys random random_endo p2tr" > $OUT/pmc_keys.log 2>&1
python - <<PY
import json
d = json.load(open("$OUT/pmc/pmc_keys.json"))
for m in ("keys", "random", "random_endo", "p2tr", "keys16", "keys20", "keys22", "keys26"):
    for k, e in d.get(m, {}).items():
        if "rocclr" in k or "rnd_fill" in k or "seq_inv" in k: continue
        print(m, k[:34], e.get("lone_launch_us_under_pmc"), e.get("valu_instr_per_key"), e.get("valu_busy"), e.get("simd_cycles_per_valu_instr"), e.get("l2_hit_rate"), e.get("hbm_side_gb_per_s"))
print(d.get("failed_passes"))
PY
cp $OUT/pmc/pmc_keys.json profiles/pmc_keys.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench_k20.json"))
print(d["value"], d["sustained"]["value"])
for o in d.get("other_configs", []):
    print("   ", o["config"][:90], o.get("value"), o.get("chip_frac"), o.get("error"))
PY
