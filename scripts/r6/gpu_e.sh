#!/bin/bash
# round 6, visit E: the 2-rank bench test, calibrations on contexts of their own (A/B), the bench line with the new roofline / in_situ / roofline_chain blocks
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6e
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_bench_two_ranks.py -m gpu -x -q > $OUT/pytest_two_ranks.txt 2>&1; tail -15 $OUT/pytest_two_ranks.txt | cut -c1-300
timeout 300 python scripts/dev/msa_calib_ab.py > $OUT/msa_calib_ab.txt 2>&1; tail -8 $OUT/msa_calib_ab.txt | cut -c1-250
cp profiles/r06_dominant_launches_before_sysr.json profiles/dominant_launches_latest.json
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-900; tail -5 $OUT/bench.err
