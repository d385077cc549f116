# The scalar-multiplication paths (uploaded scalars, random keys, P2TR): in-tree library against vgen_amd/libvgen_hip.so.<tag>, interleaved, with a
# parity check of the in-tree build first.   usage (GPU box): bash tools/keys_ab.sh tag
T=$1
keys() { python tools/gpu_perf_keys.py 4 2>&1 | tail -1 | cut -c1-60; }
b() { python bench.py --no-cpu-baseline --no-other-configs --sustained-seconds 0 --multi-leg-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-9.1f %s' % (d['value'], d['config']['workload'][:50]))"; }
run() { keys; b --format p2tr --pattern '^bc1pqqq' --steps 96 --warmup 16; }
echo "== parity (KEYS / random / P2TR tests of the GPU suite)"; timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "keys or random or p2tr or table_width" 2>&1 | tail -1
cp vgen_amd/libvgen_hip.so /tmp/libA.so
for i in 1 2; do
  cp /tmp/libA.so vgen_amd/libvgen_hip.so; echo "== A"; run
  cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so; echo "== $T"; run
done
cp /tmp/libA.so vgen_amd/libvgen_hip.so
