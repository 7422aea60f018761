#!/bin/bash
# round 6, visit H: compute units reserved for the chaining DP's serial stream (CL_CTX_CU_RESERVE=r per XCD) — A/B on the MSA's wall-clock and on one leaf merge alone
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6h
mkdir -p $OUT
cd $R
for v in 0 8 0 8 16 4; do
  echo "CL_CTX_CU_RESERVE=$v"
  CL_CTX_CU_RESERVE=$v timeout 120 python scripts/dev/leaf_merge_timing.py 2>/dev/null | tail -1
  CL_CTX_CU_RESERVE=$v timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_$v.txt 2>&1; grep -E "workers|align ms" $OUT/msa_$v.txt | cut -c1-220
done
