O=gpurun_out/r05l; mkdir -p $O
X="--no-other-configs --sustained-seconds 0 --multi-leg-seconds 0"
echo "== parity (fmt 5, 32768 keys x 4 starts)"; cp vgen_amd/libvgen_hip.so /tmp/libA.so; cp vgen_amd/libvgen_hip.so.kecc vgen_amd/libvgen_hip.so; python tests/manual/gpu_smoke.py 5 32768 2>&1 | grep -c 'mismatches 0 /'; cp /tmp/libA.so vgen_amd/libvgen_hip.so
STEPS=4096 bash tools/ab_fmt.sh kecc --format ethereum --pattern '^0xdead' --ci $X
STEPS=4096 bash tools/ab_fmt.sh kecc --format ethereum --pattern '^0xdead' --ci --endo $X
STEPS=4096 bash tools/ab_fmt.sh kecc --format ethereum --pattern 'dead.*beef' $X
