#!/bin/bash
# round 6, visit A: per-phase host/device timeline of cl_anchor_chain on a leaf merge and on the whole 10 x 1 Mbp MSA (one worker), baseline before any change
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6a
mkdir -p $OUT
cd $R
CL_CHAIN_TIMING=1 timeout 300 python scripts/dev/leaf_merge_timing.py > $OUT/leaf.out 2> $OUT/leaf.err; tail -3 $OUT/leaf.out
CL_CHAIN_TIMING=1 timeout 600 python scripts/c3_profile.py 10 1000000 1 > $OUT/c3_w1.out 2> $OUT/c3_w1.err; tail -2 $OUT/c3_w1.out | cut -c1-300
timeout 300 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_timeline.txt 2>&1; cat $OUT/msa_timeline.txt | cut -c1-300
