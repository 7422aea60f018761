#!/bin/bash
# round 5: the multi-rank path of bench.py on the final tree, two ranks sharing the one device of the box (dry run: gloo collectives, no RCCL between devices)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5g3
mkdir -p $OUT
cd $R
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --length 100000 --no-extras --no-cpu-baseline > $OUT/bench_2rank_100k.json 2>$OUT/bench_2rank_100k.err; tail -c 1500 $OUT/bench_2rank_100k.json | cut -c1-1500; grep -i "error\|Traceback" $OUT/bench_2rank_100k.err | head -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5g3/bench_2rank_100k.json'))
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d.get('with_a_join_per_pass'), str(d.get('config',{}).get('stitch_sharding'))[:300])
PY
