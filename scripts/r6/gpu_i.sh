#!/bin/bash
# round 6, visit I: the chaining DP's serial stream at the highest stream priority (CL_CTX_SERIAL_PRIO=1), A/B on the MSA
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6i
mkdir -p $OUT
cd $R
for v in 0 1 0 1; do
  echo "CL_CTX_SERIAL_PRIO=$v"
  CL_CTX_SERIAL_PRIO=$v timeout 120 python scripts/dev/leaf_merge_timing.py 2>/dev/null | tail -1
  CL_CTX_SERIAL_PRIO=$v timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_$v.txt 2>&1; grep -E "workers|align ms" $OUT/msa_$v.txt | cut -c1-220
done
