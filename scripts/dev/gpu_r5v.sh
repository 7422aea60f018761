#!/bin/bash
# round 5: rounds of three strips in popoa_lane_kernel against rounds of four — parity, the step, single pairs, the stress set
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5v
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.txt 2>&1; tail -5 $OUT/pytest_parity.txt
step() { timeout 200 python scripts/step_launches.py --steps 20 --warmup 3 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%d['ms_per_step'])"; }
for i in 1 2 3; do
  for w in 3 4; do echo "CL_LANE_WAVES=$w: $(CL_LANE_WAVES=$w step)" | tee -a $OUT/ab.txt; done
done
for i in 1 2; do
  for m in 256 384 512; do echo "CL_LANE_WAVES=3 CL_LANE_MIN_SWEEP=$m: $(CL_LANE_WAVES=3 CL_LANE_MIN_SWEEP=$m step)" | tee -a $OUT/ab.txt; done
done
timeout 300 python scripts/dev/lane_probe.py 2>&1 | cut -c1-220 | tee $OUT/lane_probe_w3.txt
CL_LANE_WAVES=4 timeout 300 python scripts/dev/lane_probe.py 2>&1 | cut -c1-220 | tee $OUT/lane_probe_w4.txt
timeout 300 python scripts/stress_set.py --json $OUT/stress.json 2>&1 | cut -c1-200 | tee $OUT/stress.txt
