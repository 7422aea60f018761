#!/bin/bash
# round 6, visit J: popoa_linear_span_kernel — parity, stress set A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6j
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "span_several or chain_problems_all or lopsided or c2_pair" > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt | cut -c1-250
CL_LINEAR_SPAN=1 timeout 300 python scripts/stress_set.py --json $OUT/stress_span1.json > $OUT/stress_span1.txt 2>&1; grep linear $OUT/stress_span1.txt | cut -c1-200
CL_LINEAR_SPAN=0 timeout 300 python scripts/stress_set.py --json $OUT/stress_span0.json > $OUT/stress_span0.txt 2>&1; grep linear $OUT/stress_span0.txt | cut -c1-200
