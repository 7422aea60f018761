#!/bin/bash
# round 5: waves per workgroup of popoa_linear_kernel (rounds of W strips) in the shared step and alone; the step's launches on a time axis
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5u
mkdir -p $OUT
cd $R
for e in "3 3" "4 8" "3 4"; do set -- $e
  CL_LINEAR_WAVES_MID=$1 CL_LINEAR_WAVES_BIG=$2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity_$1_$2.txt 2>&1; echo "parity mid $1 big $2: $(tail -1 $OUT/pytest_parity_$1_$2.txt)"
done
step() { timeout 200 python scripts/step_launches.py --steps 20 --warmup 3 --json /tmp/s.json > /dev/null 2>>$OUT/step.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('%.3f ms per step'%d['ms_per_step'])"; }
for i in 1 2; do
  for e in "4 16" "4 8" "4 4" "4 3" "3 16" "3 4" "3 3"; do set -- $e
    echo "CL_LINEAR_WAVES_MID=$1 CL_LINEAR_WAVES_BIG=$2: $(CL_LINEAR_WAVES_MID=$1 CL_LINEAR_WAVES_BIG=$2 step)" | tee -a $OUT/ab.txt
  done
done
for e in "4 16" "4 4" "3 3" "4 8"; do set -- $e
  echo "# stress set, MID=$1 BIG=$2" | tee -a $OUT/stress.txt
  CL_LINEAR_WAVES_MID=$1 CL_LINEAR_WAVES_BIG=$2 timeout 300 python scripts/stress_set.py --json $OUT/stress_$1_$2.json 2>&1 | grep linear | cut -c1-200 | tee -a $OUT/stress.txt
done
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 6 --warmup 2 --json $OUT/tl.json > /dev/null 2>$OUT/tl.err
cd $R
python3 scripts/dev/step_timeline.py $OUT/tl 13 | tee $OUT/step_timeline.txt
rm -rf $OUT/tl
