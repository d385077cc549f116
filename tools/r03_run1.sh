#!/bin/bash
# round-3 first GPU call: the GPU suite, the driver-shaped bench line, kernel durations alone on the chip, counter passes
# of the scalar-multiplication paths.  Steps are joined so that a failed GPU step starts no further one.
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err || { echo "bench failed"; tail -5 $OUT/bench_k20.err; exit 1; }
python - <<PY
import json
d = json.load(open("$OUT/bench_k20.json"))
print("k20", d["value"], "sustained", d["sustained"]["value"], "MHz", d["roofline"].get("shader_clock_mhz_timed_region"), "lone", d["roofline"].get("lone_launch"))
print("ttfm", d.get("time_to_first_match"))
for o in d.get("other_configs", []):
    print("   ", o["config"][:70], o.get("value"), o.get("chip_frac"), o.get("error"))
PY
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-configs --sustained-seconds 0.5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --frames 1 --steps 256 --warmup 16 $Q > $OUT/trace_frames1_bench.json 2> $OUT/trace1.err || { echo "trace failed"; exit 1; }
cp $(find $OUT/trace1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_frames1.csv
cat $OUT/kernel_stats_frames1.csv | cut -c1-160
rm -rf $OUT/trace1
cd $GRAFT_REPO_ROOT && bash tools/pmc_keys.sh $TAG/pmc "keys p2tr" > $OUT/pmc_keys.log 2>&1
tail -c 3000 $OUT/pmc_keys.log
