#!/bin/bash
# round-3 second GPU call: full GPU suite (new: random dispatch, failing contexts, near-limit LDS), bench with the random-stream
# leg, A/B of 3 vs 4 waves per SIMD for the register-capped variants that spill
TAG=${1:-r03b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err || { echo "bench failed"; tail -5 $OUT/bench_k20.err; exit 1; }
python - <<PY
import json
d = json.load(open("$OUT/bench_k20.json"))
print("k20", d["value"], "sustained", d["sustained"]["value"], "MHz", d["roofline"].get("shader_clock_mhz_timed_region"))
for o in d.get("other_configs", []):
    print("   ", o["config"][:70], o.get("value"), o.get("chip_frac"), o.get("error"))
PY
{
echo "# 3 vs 4 waves per SIMD for the register-capped seq_bwd variants that carry spills at 4 (P2SH-P2WPKH, P2PKH-uncompressed, FULL = on-device DFA)"
echo "# A = in-tree (4 waves: 128 VGPRs, 12-80 B scratch), w3 = 3 waves (168 VGPRs, no scratch); Mkeys/s over 256 steps, 12 frames"
STEPS=256 bash tools/ab_fmt.sh w3 --format p2sh-p2wpkh --pattern '^3Cat' --no-other-configs --sustained-seconds 0
STEPS=256 bash tools/ab_fmt.sh w3 --format p2pkh-uncompressed --pattern '^1Cat' --no-other-configs --sustained-seconds 0
STEPS=256 bash tools/ab_fmt.sh w3 --format p2pkh --pattern '1[Oo]ri' --no-other-configs --sustained-seconds 0
STEPS=256 bash tools/ab_fmt.sh w3 --format p2sh-p2wpkh --pattern '^3Cat' --endo --no-other-configs --sustained-seconds 0
STEPS=256 bash tools/ab_fmt.sh w3 --format p2pkh-uncompressed --pattern '^1Cat' --endo --no-other-configs --sustained-seconds 0
} > $OUT/waves_ab.txt 2>&1
cat $OUT/waves_ab.txt
