# A/B of whole-format bench lines: in-tree lib vs vgen_amd/libvgen_hip.so.<tag>   usage: bash tools/ab_fmt.sh tag "bench args"
T=$1; shift
run() { python bench.py --no-cpu-baseline --steps ${STEPS:-256} --warmup 32 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10.1f %s' % (d['value'], d['config']['workload'][:60]))"; }
cp vgen_amd/libvgen_hip.so /tmp/libA.so
echo "== A"; run "$@"; run "$@"
cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so
echo "== $T"; run "$@"; run "$@"
cp /tmp/libA.so vgen_amd/libvgen_hip.so
echo "== A"; run "$@"
