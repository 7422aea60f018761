#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6n
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_cyclize_flow.py tests/test_gpu_far_modes.py tests/test_gpu_chain.py tests/test_msa.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt | cut -c1-300
timeout 400 python scripts/polish_profile.py cyclize_50x8k 1 | head -10 | cut -c1-200
CL_CHAIN_GROUP_PATH=0 timeout 400 python scripts/polish_profile.py cyclize_50x8k 1 | head -9 | cut -c1-200
