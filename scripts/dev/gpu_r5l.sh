#!/bin/bash
# round 5: wide launches of the systolic kernel dealt into parts; with and without the two-pairs-per-wave chain kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5l
mkdir -p $OUT
cd $R
export CL_STITCH_SCHED_LOG=1
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>/tmp/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%(d['ms_per_step']))"; grep "stitch plan" /tmp/step.err | tail -1 | cut -c1-1100; }
for i in 1 2 3; do
  echo "split 1400, no duos: $(CL_LINEAR_DUOS=0 step)" | tee -a $OUT/ab.txt
  echo "split 0, no duos: $(CL_STITCH_SPLIT=0 CL_LINEAR_DUOS=0 step)" | tee -a $OUT/ab.txt
  echo "split 900, no duos: $(CL_STITCH_SPLIT=900 CL_LINEAR_DUOS=0 step)" | tee -a $OUT/ab.txt
  echo "split 1400, duos: $(step)" | tee -a $OUT/ab.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
