#!/bin/bash
# rocprofv3 kernel stats of the whole configs[2] MSA (per-kernel totals are dominated by the root merge's chaining DP)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -o c3 -- python3 $R/scripts/c3_profile.py ${1:-10} ${2:-1000000} 1 > $OUT/c3_prof_run.txt 2>$OUT/prof_c3.err
tail -2 $OUT/c3_prof_run.txt | cut -c1-300
cut -c1-220 $OUT/prof_c3/c3_kernel_stats.csv | head -30
rm -f $OUT/prof_c3/c3_kernel_trace.csv
