#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_chain -o r01 -- python3 $R/scripts/chain_bench.py > $OUT/chain_bench.txt 2>$OUT/prof_chain.err
cat $OUT/chain_bench.txt | tail -3
cat $OUT/prof_chain/r01_kernel_stats.csv | cut -c1-200
