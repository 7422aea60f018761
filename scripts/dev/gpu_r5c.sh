#!/bin/bash
# round 5: 50 x 100 kbp again, with and without -c, now that the far pass reaches the 625-combination root: the GFAs must be the ones of the runs before
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5c2
mkdir -p $OUT
cd $R
timeout 900 python scripts/configs4_walk.py 50 100000 --workers 1 --json $OUT/c4_50x100k_far.json --log $OUT/a.log 2>$OUT/a.err | tail -c 600
timeout 1500 python scripts/configs4_walk.py 50 100000 --cyclize --dup 20000 --min-cyclizing-length 10000 --workers 1 --json $OUT/c4_50x100k_c_bonds_far.json --log $OUT/b.log 2>$OUT/b.err | tail -c 600
rm -f $OUT/*.log
