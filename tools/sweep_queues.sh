# Throughput by stream kind x frames x chain priority (run on the GPU box from the repo root); P2PKH ^1Cat, 2^20 keys per
# dispatch, 4096 dispatches per line.  GPU_MAX_HW_QUEUES is NOT set: "plain" therefore means the HIP default of 4 queues.
for PRIO in 1 0; do
  for K in priority plain cumask; do
    echo "== VGEN_STREAM_KIND=$K VGEN_CHAIN_PRIO=$PRIO"
    VGEN_CHAIN_PRIO=$PRIO VGEN_STREAM_KIND=$K python tools/topo_sweep.py --topos frame --frames ${FRAMES:-1,2,4,8,12,16,20} --fused 0 --steps 4096 2>&1 | grep Mkeys | cut -c45-112
  done
done
echo "== GPU_MAX_HW_QUEUES=24 VGEN_STREAM_KIND=plain VGEN_CHAIN_PRIO=1 (the round-1 configuration plus the priority fix)"
GPU_MAX_HW_QUEUES=24 VGEN_STREAM_KIND=plain python tools/topo_sweep.py --topos frame --frames ${FRAMES:-1,2,4,8,12,16,20} --fused 0 --steps 4096 2>&1 | grep Mkeys | cut -c45-112
