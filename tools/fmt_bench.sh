# Mkeys/s of the BASELINE configurations + the other formats, one line each (run on the GPU box).
run() { python bench.py --no-cpu-baseline --steps ${STEPS:-2048} --warmup 64 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10.1f %s' % (d['value'], d['config']['workload']))"; }
run --format p2pkh --pattern '^1Cat'
run --format p2wpkh --pattern 'dead$'
run --format ethereum --pattern '^0xdead' --ci
run --format p2sh-p2wpkh --pattern '^3Cat'
run --format p2pkh-uncompressed --pattern '^1Cat'
run --format p2pkh --pattern '1[Oo]ri'
run --format p2tr --pattern '^bc1pqqq' --steps 64
