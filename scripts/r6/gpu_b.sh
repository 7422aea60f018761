#!/bin/bash
# round 6, visit B: parity of the chaining host changes (select by radix order, shared graph order, tie trees prebuilt beside the DP) + the MSA's timeline, A/B of streams per context
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_far_modes.py tests/test_gpu_host_seams.py tests/test_msa.py tests/test_c3_full.py -m gpu -x -q > $OUT/pytest_chain.txt 2>&1; tail -3 $OUT/pytest_chain.txt
CL_CHAIN_TIMING=1 timeout 300 python scripts/dev/leaf_merge_timing.py > $OUT/leaf.out 2> $OUT/leaf.err; tail -1 $OUT/leaf.out
timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_timeline.txt 2>&1; grep workers $OUT/msa_timeline.txt | cut -c1-250
for s in 3 4 5; do
  CL_CTX_STREAMS=$s timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_timeline_s$s.txt 2>&1; echo "streams $s"; grep workers $OUT/msa_timeline_s$s.txt | cut -c1-250
done
timeout 200 python scripts/dev/msa_timeline.py 3 5 6 > $OUT/msa_timeline_w.txt 2>&1; grep workers $OUT/msa_timeline_w.txt | cut -c1-250
