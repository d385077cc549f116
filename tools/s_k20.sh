run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --sustained-seconds 1.0 --multi-leg-seconds 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing_region']; print('%-8.1f sustained %-8.1f fill %.0f steady %.0f drain %.0f' % (d['value'], d['sustained']['value'], t['fill_us'], t['steady_us'], t['drain_us']))"; }
for i in 1 2 3; do
  echo -n "S=8 F=12: "; VGEN_SEQ_S=8 run
  echo -n "S=4 F=12: "; VGEN_SEQ_S=4 run
  echo -n "S=4 F=8 : "; VGEN_SEQ_S=4 run --frames 8
  echo -n "S=4 F=6 : "; VGEN_SEQ_S=4 run --frames 6
  echo -n "S=16 F=12: "; VGEN_SEQ_S=16 run
done
