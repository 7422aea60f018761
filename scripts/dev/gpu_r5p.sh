#!/bin/bash
# round 5: passes of a resident plan that overlap (lazy join) against a join per pass; parity of everything that runs plans; the timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5p
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_host_seams.py -m gpu -q -x > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
step() { rm -f /tmp/s.json; timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step, host enqueue %.3f ms per step'%(d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for i in 1 2 3; do
  echo "passes overlap (default): $(step)" | tee -a $OUT/ab.txt
  echo "CL_STITCH_JOIN=eager: $(CL_STITCH_JOIN=eager step)" | tee -a $OUT/ab.txt
  echo "passes overlap, CL_STITCH_RECAL=0: $(CL_STITCH_RECAL=0 step)" | tee -a $OUT/ab.txt
  echo "passes overlap, CL_STITCH_ORDER=cost: $(CL_STITCH_ORDER=cost step)" | tee -a $OUT/ab.txt
done
tail -3 $OUT/step.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 8 --warmup 4 --json $OUT/tl.json > /dev/null 2>$OUT/tl.err
python3 $R/scripts/dev/step_timeline.py $OUT/tl 14 | tee $OUT/step_timeline.txt
rm -rf $OUT/tl
cd $R
timeout 600 python bench.py --no-extras --no-cpu-baseline 2>$OUT/bench.err | cut -c1-700
