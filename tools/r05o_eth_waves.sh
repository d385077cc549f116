O=gpurun_out/r05o; mkdir -p $O
X="--no-other-configs --sustained-seconds 0 --multi-leg-seconds 0"
{ echo "# A = in-tree: Ethereum at four waves per SIMD, running inverse in LDS (no scratch); eth3 = three waves (round-4 occupancy), both with the Keccak block"
STEPS=4096 bash tools/ab_fmt.sh eth3 --format ethereum --pattern '^0xdead' --ci $X
STEPS=4096 bash tools/ab_fmt.sh eth3 --format ethereum --pattern '^0xdead' --ci --endo $X
STEPS=4096 bash tools/ab_fmt.sh eth3 --format ethereum --pattern 'dead.*beef' $X; } > $O/ab_eth3.txt 2>&1
cat $O/ab_eth3.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "eth or keccak or Eth or endo or dump_full or generated_patterns" > $O/gputests_eth.txt 2>&1; tail -3 $O/gputests_eth.txt
