#!/bin/bash
# round 5: the step's gaps — host enqueue time per step, issue order (few-workgroup launches first), graph replay, the timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5q
mkdir -p $OUT
cd $R
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step, host enqueue %.3f ms per step'%(d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for i in 1 2 3; do
  echo "few-workgroup launches first (default): $(step)" | tee -a $OUT/ab.txt
  echo "CL_STITCH_ORDER=cost: $(CL_STITCH_ORDER=cost step)" | tee -a $OUT/ab.txt
  echo "CL_STITCH_GRAPH=1: $(CL_STITCH_GRAPH=1 step)" | tee -a $OUT/ab.txt
  echo "CL_STITCH_RECAL=0: $(CL_STITCH_RECAL=0 step)" | tee -a $OUT/ab.txt
done
tail -5 $OUT/step.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 8 --warmup 4 --json $OUT/tl.json > /dev/null 2>$OUT/tl.err
python3 $R/scripts/dev/step_timeline.py $OUT/tl 14 | tee $OUT/step_timeline.txt
rm -rf $OUT/tl
