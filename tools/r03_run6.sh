#!/bin/bash
# random keys on endomorphism contexts + full-size dump tests: GPU suite, bench legs
TAG=${1:-r03g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 256 --warmup 16 --sustained-seconds 0.5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(d["value"])
for o in d.get("other_configs", []):
    print("   ", o["config"][:90], o.get("value"), o.get("chip_frac"), o.get("error"))
PY
