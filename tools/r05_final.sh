# Round-5 evidence at HEAD on one MI355X: GPU suite, the driver's bench command, the default bench, kernel stats of the driver's command
set -e
O=gpurun_out/r05j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -40 $O/gputests.txt; exit 1; }
tail -2 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for f in ("bench_driver_args","bench"):
    d=json.loads(open("gpurun_out/r05j/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], "sustained", d.get("sustained",{}).get("value"), "frac", d["roofline"]["frac"], d["roofline"].get("frac_sustained"))
    for o in d.get("other_configs",[]): print("   %-70s %s"%(o.get("workload","")[:70], o.get("value")))
PY
