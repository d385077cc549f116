# the driver's 20-step command, in-tree library against vgen_amd/libvgen_hip.so.<tag>, interleaved   usage: bash tools/k20_ab.sh tag [rounds]
T=$1; R=${2:-4}
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --sustained-seconds 0.3 --multi-leg-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing_region']; print('%-8.1f fill %.0f steady %.0f drain %.0f  completions %s' % (d['value'], t['fill_us'], t['steady_us'], t['drain_us'], [round(x) for x in t['completion_us'][:6]]))"; }
cp vgen_amd/libvgen_hip.so /tmp/libA.so
for i in $(seq 1 $R); do
  cp /tmp/libA.so vgen_amd/libvgen_hip.so; echo -n "A      "; run
  cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so; echo -n "$T  "; run
done
cp /tmp/libA.so vgen_amd/libvgen_hip.so
