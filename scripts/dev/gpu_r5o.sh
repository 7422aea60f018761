#!/bin/bash
# round 5: streams per plan and re-dealing rounds with overlapping passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5o
mkdir -p $OUT
cd $R
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%(d['ms_per_step']))"; }
for i in 1 2; do
  for n in 6 8 10 12; do
    for r in 1 2; do echo "CL_CTX_STREAMS=$n CL_STITCH_RECAL=$r: $(CL_CTX_STREAMS=$n CL_STITCH_RECAL=$r step)" | tee -a $OUT/ab.txt; done
  done
done
tail -3 $OUT/step.err
