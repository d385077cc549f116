O=gpurun_out/r05n; mkdir -p $O
X="--no-other-configs --sustained-seconds 0 --multi-leg-seconds 0"
{ STEPS=4096 bash tools/ab_fmt.sh eth4 --format ethereum --pattern '^0xdead' --ci $X
STEPS=4096 bash tools/ab_fmt.sh eth4 --format ethereum --pattern '^0xdead' --ci --endo $X
STEPS=4096 bash tools/ab_fmt.sh eth4 --format ethereum --pattern 'dead.*beef' $X; } > $O/ab_eth4.txt 2>&1
cat $O/ab_eth4.txt
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
