#!/bin/bash
# round 5: high-priority streams for the longest launches of a stitch plan
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5f2
mkdir -p $OUT
cd $R
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 30 --warmup 5 --json /tmp/s.json > /dev/null 2>/tmp/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%(d['ms_per_step']))"; }
for i in 1 2 3; do
  for k in 0 1 2 3 4; do echo "CL_CTX_PRIO_STREAMS=$k: $(CL_CTX_PRIO_STREAMS=$k step)" | tee -a $OUT/ab.txt; done
done
