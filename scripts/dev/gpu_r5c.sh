#!/bin/bash
# round 5: the far pass at the 625-combination root only where the context's DPs choose branch-and-bound — 50 x 100 kbp (they sweep: root without the far structures) and 50 x 1 Mbp (they prune: root with them)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5c4
mkdir -p $OUT
cd $R
timeout 900 python scripts/configs4_walk.py 50 100000 --workers 1 --json $OUT/c4_50x100k_final.json --log $OUT/a.log 2>$OUT/a.err | tail -c 300
grep -n "far pass not taken" $OUT/a.log | tail -3
timeout 900 python scripts/configs4_walk.py 50 1000000 --workers 1 --json $OUT/c4_50x1M_final.json --log $OUT/b.log 2>$OUT/b.err | tail -c 300
grep -n "far pass not taken" $OUT/b.log | tail -3
rm -f $OUT/*.log
