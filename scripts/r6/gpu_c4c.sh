#!/bin/bash
# round 6: BASELINE configs[4]'s width with -c at 1 Mbp (the step of round 5 that took 547 s: five 150-kbp duplications, four workers) on the tree with the group-by-group chaining path
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6c4c
mkdir -p $OUT
cd $R
timeout 1500 python scripts/configs4_walk.py 50 1000000 --cyclize --dup 150000 --carriers 5 --workers 4 --json $OUT/configs4_50x1M_c.json --log $OUT/configs4_50x1M_c.log > $OUT/stdout.txt 2> $OUT/stderr.txt
echo "rc $?" >> $OUT/stdout.txt
tail -c 1200 $OUT/stdout.txt; tail -3 $OUT/stderr.txt | cut -c1-300
gzip -f $OUT/configs4_50x1M_c.log; ls -la $OUT
