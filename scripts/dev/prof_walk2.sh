#!/bin/bash
# kernel trace of the 2 x 1 Mbp chaining DP (affine + gap-free) with the walk variant given by the environment; prints the per-macro-block timeline
# usage (GPU box, from the repo root): bash scripts/dev/prof_walk2.sh TAG   (after scripts/dev/walk2_ab.py has written /tmp/walk2_ab_input.npz)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
OUT=$R/gpurun_out/prof_walk2_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for kind in affine sparse; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$kind -o t -- python3 $R/tests/far_ab_child.py /tmp/walk2_ab_input.npz $kind > $OUT/$kind.txt 2> $OUT/$kind.err
  grep RESULT $OUT/$kind.txt | cut -c1-60,190-
  f=$(find $OUT/$kind -name "t_kernel_trace.csv" | head -1)
  python3 $R/scripts/dev/block_timeline.py $f
  s=$(find $OUT/$kind -name "t_kernel_stats.csv" | head -1)
  head -8 $s | cut -c1-60,100-200
  rm -f $f
done
