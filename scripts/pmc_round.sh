#!/bin/bash
# HBM traffic of the stitch kernels from the PMC counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run via gpurun from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o r01 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/pmc_$c.json 2>$OUT/pmc_$c.err
  ls $OUT/pmc_$c | head
done
cd $R
python3 scripts/pmc_summary.py $OUT | tee $OUT/pmc_summary.txt
