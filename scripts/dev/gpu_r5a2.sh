#!/bin/bash
# round 5: routing thresholds of the register kernel with overlapping passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5a3
mkdir -p $OUT
cd $R
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>/tmp/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%(d['ms_per_step']))"; }
for i in 1 2 3; do
  for m in 160 208 256 320; do echo "CL_LANE_MIN_SWEEP=$m: $(CL_LANE_MIN_SWEEP=$m step)" | tee -a $OUT/ab.txt; done
done
