#!/bin/bash
# round 5: fewer launches per pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5k
mkdir -p $OUT
cd $R
export CL_STITCH_SCHED_LOG=1
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>/tmp/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step, %d launches'%(d['ms_per_step'], len([l for l in d['launches'] if l['n_problems']])))"; grep "stitch plan" /tmp/step.err | tail -1 | cut -c1-1100; }
for i in 1 2 3; do
  echo "A default: $(step)" | tee -a $OUT/ab.txt
  echo "B big=8: $(CL_LINEAR_WAVES_BIG=8 step)" | tee -a $OUT/ab.txt
  echo "C big=8, no quads: $(CL_LINEAR_WAVES_BIG=8 CL_NO_LINEAR_QUADS=1 step)" | tee -a $OUT/ab.txt
  echo "D big=8, one launch per systolic shape: $(CL_LINEAR_WAVES_BIG=8 CL_STITCH_MERGE_ALL=1 step)" | tee -a $OUT/ab.txt
  echo "E all three: $(CL_LINEAR_WAVES_BIG=8 CL_NO_LINEAR_QUADS=1 CL_STITCH_MERGE_ALL=1 step)" | tee -a $OUT/ab.txt
done
