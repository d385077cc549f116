# full GPU suite, bench summary, CLI generate with and without the endomorphism (run on the GPU box from the repo root)
(timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02o_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02o_pytest.log; tail -4 gpurun_out/r02o_pytest.log)
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02o_bench.json 2> gpurun_out/r02o_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02o_bench.json"))
print(d["value"], d["sustained"])
for o in d["other_configs"]:
    print("  ", o["config"][:70], o["value"], o.get("chip_frac"))
PY
cd vgen_amd && timeout 60 ./vgen-hip generate -p "^1Cats" -o json; timeout 60 ./vgen-hip generate -p "^1Cats" -o json --no-endo | grep -v wif
