# Split form: keys per lane of seq_hash_kernel (VGEN_HASH_KPL) x frames in flight, and the point-arithmetic kernel's issue priority
# (library variants ecprio1 / ecprio2 = -DVG_EC_PRIO=1 / 2), against the fused kernel (VGEN_SPLIT=0), on one box.
FR=${1:-"2,4,8,12"}
export VGEN_PERF_STEPS=${VGEN_PERF_STEPS:-16000}
run() { python tools/gpu_perf.py 0 $FR 2>&1 | grep Mkeys | sed "s/^S=[^ ]* WG=256 PREG=- batch=2^20 //" | cut -c1-90; }
echo "== fused"; VGEN_SPLIT=0 run
for K in 1 2 4 8 16; do echo "== split kpl=$K"; VGEN_HASH_KPL=$K run; done
cp vgen_amd/libvgen_hip.so /tmp/libA.so
for T in "$@"; do
  [ "$T" = "$FR" ] && continue
  cp vgen_amd/libvgen_hip.so.$T vgen_amd/libvgen_hip.so
  for K in 2 4; do echo "== $T split kpl=$K"; VGEN_HASH_KPL=$K run; done
done
cp /tmp/libA.so vgen_amd/libvgen_hip.so
echo "== fused again"; VGEN_SPLIT=0 run
