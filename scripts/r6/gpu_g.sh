#!/bin/bash
# round 6, visit G: own radix sort (cl_radix.h) against the library sort it replaced (rocPRIM called directly, the tree of commit 3ceed4a's three files), SAME box, alternating:
# the 10 x 1 Mbp MSA's wall-clock with four workers, and one leaf merge alone
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6g
mkdir -p $OUT
cd $R
L=centrolign_amd/lib
cp $L/libcentrolign_amd.so $L/ab/libcentrolign_amd_own.so
for rep in 1 2; do
  for v in own rocprim; do
    cp $L/ab/libcentrolign_amd_$v.so $L/libcentrolign_amd.so
    timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_${v}_$rep.txt 2>&1; echo "$v $rep"; grep workers $OUT/msa_${v}_$rep.txt | cut -c1-200
    timeout 100 python scripts/dev/leaf_merge_timing.py 2>/dev/null | tail -1
  done
done
cp $L/ab/libcentrolign_amd_own.so $L/libcentrolign_amd.so
