# Regenerates the round's evidence under gpurun_out/prof_<tag>/ (run on the GPU box from the repo root):
#   bench.json                 default bench.py line (sustained leg, other configs, cpu_baseline)
#   bench_driver_args.json     the driver's invocation (--steps 20 --warmup 5)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of `bench.py --frames 1` (kernels alone on the chip: the
#                              average seq_bwd_kernel duration must agree with roofline.lone_launch.avg_launch_ms)
#   kernel_stats_overlapped.csv  the same for the default run (20 frames in flight; tracing slows it down)
#   pmc/pmc_valu.json          issue-side counters + HBM bytes (tools/pmc_valu.sh)
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-configs --sustained-seconds 0.5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --frames 1 --steps 256 --warmup 16 $Q > $OUT/trace_frames1_bench.json 2> $OUT/trace1.err
cp $(find $OUT/trace1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/trace_bench.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_overlapped.csv
cd $GRAFT_REPO_ROOT && bash tools/pmc_valu.sh prof_$TAG/pmc > $OUT/pmc.log 2>&1
find $OUT -name "*.csv" -size +1M -delete
rm -rf $OUT/trace/*/*.db $OUT/trace1/*/*.db
