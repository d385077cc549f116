# frames x GPU_MAX_HW_QUEUES sweep (single process), P2PKH ^1Cat 2^20 keys/dispatch
for Q in ${QUEUES:-12 16 24}; do
  for F in ${FRAMES:-8 10 12 16}; do
    echo -n "queues=$Q "
    GPU_MAX_HW_QUEUES=$Q python tools/gpu_perf.py 0 $F 2>&1 | grep Mkeys | sed 's/WG=256 PREG=- //; s/cand=.*//'
  done
done
