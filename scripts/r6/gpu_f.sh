#!/bin/bash
# round 6, visit F: the hand-written radix sort / prefix sum (cl_radix.h) under the match finder, the chaining DP's value index and the far pass's set-up: parity + timings
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6f
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_match_finder.py tests/test_gpu_chain.py tests/test_gpu_far_modes.py tests/test_gpu_host_seams.py tests/test_msa.py tests/test_c3_full.py tests/test_cyclize_flow.py tests/test_restart.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
CL_CHAIN_TIMING=1 timeout 300 python scripts/dev/leaf_merge_timing.py > $OUT/leaf.out 2> $OUT/leaf.err; tail -1 $OUT/leaf.out; sed -n '/==== timed merge/,$p' $OUT/leaf.err | grep -E "suffix_array|find_matches|far pass setup|candidates" | cut -c1-200
timeout 200 python scripts/match_root_bench.py > $OUT/match_root.txt 2>&1; tail -6 $OUT/match_root.txt | cut -c1-250
timeout 200 python scripts/dev/msa_timeline.py 4 4 > $OUT/msa_timeline.txt 2>&1; grep workers $OUT/msa_timeline.txt | cut -c1-250
