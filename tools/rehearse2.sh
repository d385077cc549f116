#!/bin/bash
# two ranks sharing the box's one GPU (VGEN_BENCH_REHEARSE=1) with the driver's arguments: exercises bench.py's N>1 timing path
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_abi_and_sharding.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
VGEN_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/rehearse2_$i.json 2> gpurun_out/rehearse2_$i.err
python -c "
import json
d = json.load(open('gpurun_out/rehearse2_$i.json'))
print(d['n_gpus'], d['steps'], d['value'], d['sustained']['value'], d['timing'])
"
done
