set -e
O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -30 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
X="--no-other-configs --sustained-seconds 0 --multi-leg-seconds 0"
STEPS=4096 bash tools/ab_fmt.sh noparki --format p2pkh-uncompressed --pattern '^1Cat' $X > $O/ab_uncompressed.txt 2>&1
cat $O/ab_uncompressed.txt
STEPS=4096 bash tools/ab_fmt.sh noparki --format p2pkh --pattern '1[Oo]ri' --endo $X > $O/ab_ori_endo.txt 2>&1
cat $O/ab_ori_endo.txt
STEPS=4096 bash tools/ab_fmt.sh noparki --format p2pkh-uncompressed --pattern '1[Oo]ri' $X > $O/ab_unc_ori.txt 2>&1
cat $O/ab_unc_ori.txt
