#!/bin/bash
# round 5: run-to-run spread of the step with overlapping passes, with the deal of the launches over the streams logged
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5m
mkdir -p $OUT
cd $R
export CL_STITCH_SCHED_LOG=1
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>/tmp/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%(d['ms_per_step']))"; grep "stitch plan" /tmp/step.err | tail -1 | cut -c1-900; }
for i in 1 2 3 4 5; do
  echo "CL_LINEAR_DUOS=1: $(CL_LINEAR_DUOS=1 step)" | tee -a $OUT/ab.txt
  echo "CL_LINEAR_DUOS=0: $(CL_LINEAR_DUOS=0 step)" | tee -a $OUT/ab.txt
  echo "CL_STITCH_RECAL=0: $(CL_STITCH_RECAL=0 step)" | tee -a $OUT/ab.txt
done
