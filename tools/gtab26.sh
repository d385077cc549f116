#!/bin/bash
# the 26-bit generator table (10 windows, 9 additions; 43 GB): parity, build time, lone-launch counters and rates of the
# random-key mode (one and six keys per draw) and of P2TR against the 22- and 24-bit tables
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_gtab26
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generator_table" 2>&1 | tail -3
for B in 22 24 26; do
  echo "== VGEN_GTAB_BITS=$B"
  VGEN_GTAB_BITS=$B VGEN_TRACE_CREATE=1 timeout -k 10 300 python tools/ttfm_formats.py 2>&1 | grep -E "generator tables|P2tr" | head -8
  VGEN_GTAB_BITS=$B timeout -k 10 200 python tools/rnd_frames.py 4 2>&1 | tail -1
  VGEN_GTAB_BITS=$B timeout -k 10 300 python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import bench, vgen_amd as vg
o = bench.keys_mode_config(vg, 1 << 20, 8, 0, 1.5, random_stream=True, endo=True)
print("random x6:", o["value"], "Mkeys/s")
o = bench.timed_config(vg, "p2tr", "^bc1pqqq", False, 1 << 20, 12, 0, 1.5, "p2tr")
print("p2tr:", o["value"], "Mkeys/s")
PY
done
cd /tmp && export TMPDIR=/tmp
export VGEN_GTAB_BITS=26
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/keys26_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_driver.py random > $OUT/keys26_sq.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/keys26_fetch -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_driver.py random > $OUT/keys26_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/keys26_tcc -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_driver.py random > $OUT/keys26_tcc.log 2>&1
unset VGEN_GTAB_BITS
python3 - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT/tools")
import pmc_keys_summarize as p, json
r = p.summarize("$OUT", "keys26")
for k, e in r.items():
    if "rocclr" in k or "gen_table" in k: continue
    print("keys26", k[:34], e.get("lone_launch_us_under_pmc"), e.get("valu_instr_per_key"), e.get("valu_busy"), e.get("simd_cycles_per_valu_instr"), e.get("l2_hit_rate"), e.get("hbm_side_gb_per_s"))
json.dump(r, open("$OUT/keys26.json", "w"), indent=1)
PY
