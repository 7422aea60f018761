#!/bin/bash
# round 5: two chain pairs of 17-32 rows per wave — parity, the step with and without
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5n
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.txt 2>&1; tail -5 $OUT/pytest_parity.txt
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%(d['ms_per_step']))"; }
for i in 1 2 3; do
  echo "two pairs of 17-32 rows per wave (default): $(step)" | tee -a $OUT/ab.txt
  echo "CL_LINEAR_DUOS=0: $(CL_LINEAR_DUOS=0 step)" | tee -a $OUT/ab.txt
done
tail -3 $OUT/step.err
timeout 300 python scripts/stress_set.py --json $OUT/stress.json 2>&1 | grep "32^2" | cut -c1-200
