#!/bin/bash
# round 5: the far pass beyond 2^32 arena words — the far-mode variants (two of them with the structures pushed past 2^32 / 2^33 words), then 50 x 1 Mbp whose root (625 combinations) now takes the far pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5y
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_far_modes.py -m gpu -q -x > $OUT/pytest_far.txt 2>&1; tail -5 $OUT/pytest_far.txt
timeout 1500 python scripts/configs4_walk.py 50 1000000 --workers 1 --json $OUT/c4_50x1M_far.json --log $OUT/c4_50x1M_far.log 2>$OUT/err.txt | tail -c 1500
grep -n "far pass\|device memory held" $OUT/c4_50x1M_far.log | tail -12 | cut -c1-260
grep -n "cl_anchor_chain\]   prep" $OUT/c4_50x1M_far.log | tail -4
rm -f $OUT/*.log
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- python3 $R/scripts/step_launches.py --steps 6 --warmup 2 --json $OUT/tl.json > /dev/null 2>$OUT/tl.err
cd $R
python3 scripts/dev/step_timeline.py $OUT/tl 13 | tee $OUT/step_timeline.txt
rm -rf $OUT/tl
