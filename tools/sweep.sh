# S x frames sweep of the sequential path (bring-up tool; run on the GPU box)
for S in 2 4 8 16; do
  export VGEN_SEQ_S=$S
  python tools/gpu_smoke.py 0 32768 2>&1 | grep -c "mismatches 0"
  python tools/gpu_perf.py 0 1,2,3,4,6,8 2>&1 | grep Mkeys
done
