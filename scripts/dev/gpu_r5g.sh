#!/bin/bash
# round 5: the multi-rank path of bench.py on the final tree, two ranks sharing the one device of the box (dry run: gloo collectives, no RCCL between devices)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5g2
mkdir -p $OUT
cd $R
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --length 100000 --no-extras --no-cpu-baseline > $OUT/bench_2rank_100k.json 2>$OUT/bench_2rank_100k.err; tail -c 900 $OUT/bench_2rank_100k.json; tail -3 $OUT/bench_2rank_100k.err | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_2rank_1M.json 2>$OUT/bench_2rank_1M.err; cut -c1-700 $OUT/bench_2rank_1M.json; tail -3 $OUT/bench_2rank_1M.err | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_merge_group.py -m gpu -q -x 2>&1 | tail -2
