#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5i
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python scripts/stress_set.py --json $OUT/stress_set.json 2>&1 | cut -c1-200 | tee $OUT/stress.txt
for i in 1 2; do timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>&1; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step, %d launches'%(d['ms_per_step'], len([l for l in d['launches'] if l['n_problems']])))"; done
