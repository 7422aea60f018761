#!/bin/bash
# round 5, after the rank-order choice: the parity file, step A/B by rank order and by the register kernel's minimum sweep, the enqueue-cost microbench with large kernel arguments
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5x
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.txt 2>&1; tail -5 $OUT/pytest_parity.txt
for o in lifo auto; do
  for i in 1 2; do
    if [ $o = lifo ]; then export CL_RANK_ORDER=lifo; else unset CL_RANK_ORDER; fi
    timeout 200 python scripts/step_launches.py --steps 20 --warmup 3 --json /tmp/s.json > /dev/null 2>$OUT/step.err; echo "rank order $o: $(python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%d['ms_per_step'])")" | tee -a $OUT/ab.txt
  done
done
unset CL_RANK_ORDER
for m in 0 128 256 512; do
  CL_LANE_MIN_SWEEP=$m timeout 200 python scripts/step_launches.py --steps 20 --warmup 3 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; echo "CL_LANE_MIN_SWEEP=$m: $(python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%d['ms_per_step'])")" | tee -a $OUT/ab.txt
done
CL_LANE_MIN_SWEEP=0 timeout 300 python scripts/stress_set.py --json $OUT/stress_lane0.json 2>&1 | cut -c1-200 | tee $OUT/stress_lane0.txt
hipcc -O2 --offload-arch=gfx950 scripts/dev/graph_cost.cpp -o /tmp/graph_cost 2>/dev/null
for a in "" b; do timeout 120 /tmp/graph_cost 1200 $a 2>&1 | grep direct | tee -a $OUT/graph_cost_args.txt; done
timeout 400 python - <<'PY' 2>&1 | tee $OUT/far_forks_both.txt
import json, sys
sys.path.insert(0, '.')
import bench
from centrolign_amd import capi
print(json.dumps(bench.big_dag_section(None)["far_forks_in_both_graphs"], indent=1))
PY
