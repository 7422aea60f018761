#!/bin/bash
# round 5: where the step's time outside its kernels goes — graph replay against direct enqueue, the gaps between steps in the trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5s
mkdir -p $OUT
cd $R
step() { timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "direct enqueue (default): $(step)" | tee -a $OUT/ab.txt
  echo "graph replay (CL_STITCH_GRAPH=1): $(CL_STITCH_GRAPH=1 step)" | tee -a $OUT/ab.txt
done
export TMPDIR=/tmp
cd /tmp
N=$(python3 -c "import json;print(len([l for l in json.load(open('$R/gpurun_out/r5t/tl.json'))['launches'] if l['n_problems']]))" 2>/dev/null || echo 14)
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 8 --warmup 4 --json $OUT/tl.json > /dev/null 2>$OUT/tl.err
python3 $R/scripts/dev/step_timeline.py $OUT/tl $N | tee $OUT/step_timeline_direct_default.txt
rm -rf $OUT/tl
CL_STITCH_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 8 --warmup 4 --json $OUT/tl2.json > /dev/null 2>$OUT/tl2.err
python3 $R/scripts/dev/step_timeline.py $OUT/tl $N | tee $OUT/step_timeline_graph.txt
rm -rf $OUT/tl
