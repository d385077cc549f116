# A/B of the in-tree libvgen_hip.so against variants saved as vgen_amd/libvgen_hip.so.<tag>, topo_sweep arguments in $1
ARGS=${1:-"--topos frame --frames 1,2,4,8 --fused 1,0 --steps 2048"}; shift
cp vgen_amd/libvgen_hip.so /tmp/libA.so
echo "== A (in-tree)"; python tools/topo_sweep.py $ARGS 2>&1 | cut -c20-120
for T in "$@"; do
  cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so
  echo "== $T"; python tests/manual/gpu_smoke.py 0 32768 2>&1 | tail -1 | cut -c1-80; python tools/topo_sweep.py $ARGS 2>&1 | cut -c20-120
done
cp /tmp/libA.so vgen_amd/libvgen_hip.so
