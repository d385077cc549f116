# S x frames sweep of the sequential path (bring-up tool; run on the GPU box)
for S in ${SVALS:-4 8}; do
  export VGEN_SEQ_S=$S
  python tests/manual/gpu_smoke.py 0 32768 2>&1 | grep -c "mismatches 0"
  python tools/gpu_perf.py 0 ${FRAMES:-12,16,20} 2>&1 | grep Mkeys
done
