#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_far -o far -- python3 $R/scripts/far_check.py /tmp/x.npz $1 > $OUT/far_run.txt 2>$OUT/prof_far.err
tail -3 $OUT/far_run.txt
cut -c1-200 $OUT/prof_far/far_kernel_stats.csv | head -14
rm -f $OUT/prof_far/far_kernel_trace.csv
