# A/B on one box: the in-tree libvgen_hip.so against variants saved as vgen_amd/libvgen_hip.so.<tag>
# usage (on the GPU box): bash tools/ab.sh "<gpu_perf args>" tagB tagC ...
ARGS=${1:-"0 16,16"}; shift
cp vgen_amd/libvgen_hip.so /tmp/libA.so
echo "== A (in-tree)"; python tools/gpu_perf.py $ARGS 2>&1 | grep Mkeys | cut -c1-120
for T in "$@"; do
  cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so
  echo "== $T"; python tests/manual/gpu_smoke.py ${ARGS%% *} 32768 2>&1 | tail -1 | cut -c1-80; python tools/gpu_perf.py $ARGS 2>&1 | grep Mkeys | cut -c1-120
done
cp /tmp/libA.so vgen_amd/libvgen_hip.so
echo "== A again"; python tools/gpu_perf.py $ARGS 2>&1 | grep Mkeys | cut -c1-120
