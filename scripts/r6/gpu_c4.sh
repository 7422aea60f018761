#!/bin/bash
# round 6: BASELINE configs[4] at its stated size, 50 x 5 Mbp, without -c (one worker), one step of scripts/configs4_walk.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6c4
mkdir -p $OUT
cd $R
free -g | head -2 > $OUT/host_mem.txt; nproc >> $OUT/host_mem.txt
timeout ${1:-2300} python scripts/configs4_walk.py 50 5000000 --workers 1 --json $OUT/configs4_50x5M.json --log $OUT/configs4_50x5M.log > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc $?" >> $OUT/stdout.txt
tail -c 1500 $OUT/stdout.txt; tail -5 $OUT/stderr.txt | cut -c1-400
grep -c . $OUT/configs4_50x5M.log; gzip -f $OUT/configs4_50x5M.log
