#!/bin/bash
# round 5: LDS classes of the systolic kernel's launches in the step with overlapping passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5j
mkdir -p $OUT
cd $R
export CL_STITCH_SCHED_LOG=1
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>/tmp/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step, %d launches'%(d['ms_per_step'], len([l for l in d['launches'] if l['n_problems']])))"; grep "stitch plan" /tmp/step.err | tail -1 | cut -c1-1100; }
for i in 1 2 3; do
  echo "12,32,64,100 (default): $(step)" | tee -a $OUT/ab.txt
  echo "20,32,64,100: $(CL_STITCH_LDS_CLASSES=20,32,64,100 step)" | tee -a $OUT/ab.txt
  echo "12,24,64,100: $(CL_STITCH_LDS_CLASSES=12,24,64,100 step)" | tee -a $OUT/ab.txt
  echo "16,40,64,100: $(CL_STITCH_LDS_CLASSES=16,40,64,100 step)" | tee -a $OUT/ab.txt
  echo "12,20,32,64: $(CL_STITCH_LDS_CLASSES=12,20,32,64 step)" | tee -a $OUT/ab.txt
done
