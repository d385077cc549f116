# A/B on one box with runs long enough for the clocks to settle (the 0.17 s runs of tools/ab.sh read ~8 % low and ±5 %):
# every variant vgen_amd/libvgen_hip.so.<tag> (and the in-tree library as "A") for VGEN_PERF_STEPS dispatches (default 40 000
# = ~3 s at 13 Gkeys/s) per frames value, ROUNDS times in interleaved order (A B C A B C ...), parity-checked once each.
# usage (on the GPU box): bash tools/ab_long.sh "<fmt> <frames,frames,...>" tagB tagC ...
ARGS=${1:-"0 12"}; shift
export VGEN_PERF_STEPS=${VGEN_PERF_STEPS:-40000}
ROUNDS=${ROUNDS:-2}
cp vgen_amd/libvgen_hip.so /tmp/libA.so
for T in "$@"; do
  cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so
  echo "== parity $T: $(VGEN_LONE_VARIANT=0 python tests/manual/gpu_smoke.py ${ARGS%% *} 32768 2>&1 | grep -c 'mismatches 0 /') of 4 starts clean"
done
for R in $(seq 1 $ROUNDS); do
  for T in A "$@"; do
    if [ $T = A ]; then cp /tmp/libA.so vgen_amd/libvgen_hip.so; else cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so; fi
    python tools/gpu_perf.py $ARGS 2>&1 | grep Mkeys | sed "s/^S=[^ ]* WG=256 PREG=- /round $R $T: /" | cut -c1-130
  done
done
cp /tmp/libA.so vgen_amd/libvgen_hip.so
