# Round-5 closing validation of the new kernels on one MI355X: full-mode GPU suite, randomised parity soak, the scan driver's random walk
O=gpurun_out/r05t; mkdir -p $O
VGEN_TEST_FULL=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_full.txt 2>&1; tail -2 $O/gputests_full.txt
timeout -k 10 700 python tests/manual/soak.py 5 160 > $O/soak.txt 2>&1; cat $O/soak.txt
for seed in 1 2 3; do timeout -k 10 200 tests/native/scan_driver_hip fuzz $seed 300 >> $O/scan_fuzz.txt 2>&1; done; tail -3 $O/scan_fuzz.txt
