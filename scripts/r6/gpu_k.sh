#!/bin/bash
# round 6, visit K: the near sweep inside the walk's launch — parity (bit-identical dp), A/B on one leaf merge alone and on the four-worker MSA
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6k
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_far_modes.py tests/test_gpu_host_seams.py tests/test_c3_full.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
for v in 1 0 1 0; do
  echo "CL_CHAIN_NEAR_IN_WALK=$v"
  CL_CHAIN_NEAR_IN_WALK=$v CL_CHAIN_TIMING=1 timeout 120 python scripts/dev/leaf_merge_timing.py 2>$OUT/leaf_$v.err | tail -1; sed -n '/==== timed merge/,$p' $OUT/leaf_$v.err | grep "device" | grep "prep" | cut -c1-120
  CL_CHAIN_NEAR_IN_WALK=$v timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_$v.txt 2>&1; grep -E "workers" $OUT/msa_$v.txt | cut -c1-220
done
for p in 4 12; do echo "parts $p"; CL_CHAIN_NEAR_PARTS=$p timeout 120 python scripts/dev/leaf_merge_timing.py 2>/dev/null | tail -1; CL_CHAIN_NEAR_PARTS=$p timeout 200 python scripts/dev/msa_timeline.py 4 > $OUT/msa_p$p.txt 2>&1; grep -E "workers" $OUT/msa_p$p.txt | cut -c1-220; done
