#!/bin/bash
# round 5: second-stage launch scheduling (in-pass durations) A/B on the step, the timeline after, parity
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5t
mkdir -p $OUT
cd $R
step() { timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%d['ms_per_step'])"; }
for i in 1 2 3; do
  for r in 0 1 2; do echo "CL_STITCH_RECAL=$r: $(CL_STITCH_RECAL=$r step)" | tee -a $OUT/ab.txt; done
done
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 6 --warmup 4 --json $OUT/tl.json > /dev/null 2>$OUT/tl.err
cd $R
python3 scripts/dev/step_timeline.py $OUT/tl 13 | tee $OUT/step_timeline.txt
rm -rf $OUT/tl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stitch.py -m gpu -q -x 2>&1 | tail -3
