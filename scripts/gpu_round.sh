#!/bin/bash
# one GPU-box visit: parity tests, bench line, rocprof kernel trace + stats (run via gpurun from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
python bench.py --steps 50 --warmup 5 2>$OUT/bench.err | tee $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r01 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_prof.json 2>$OUT/prof.err
ls -R $OUT/prof | head -30
# match finding on the same pair: per-kernel times of the device suffix sort / LCP
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -o r01 -- python3 $R/scripts/match_bench.py > $OUT/match_prof.txt 2>$OUT/match_prof.err
tail -5 $OUT/match_prof.txt
