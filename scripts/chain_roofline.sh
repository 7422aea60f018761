#!/bin/bash
# Hardware pricing of the chaining DP's kernels (round-5 verdict item 2, SURVEY §8(d) last bullet): four rocprofv3 passes over ONE command — the Anchorer::anchor_chain seam on
# the 2 x 1 Mbp pair (scripts/anchor_bench.py: both whole-graph DPs + fill-in) — kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ counters.  PMC passes carry --kernel-trace only.
# scripts/chain_roofline.py joins them per kernel -> gpurun_out/chainroof/chain_roofline.json -> profiles/chain_roofline_latest.json (bench.py: roofline_chain).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/chainroof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/scripts/anchor_bench.py > $OUT/trace.out 2>$OUT/trace.err
rm -f $OUT/trace/*kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/scripts/anchor_bench.py > $OUT/pmc_$c.out 2>$OUT/pmc_$c.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -o p -- python3 $R/scripts/anchor_bench.py > $OUT/pmc_SQ.out 2>$OUT/pmc_SQ.err
cd $R
python3 scripts/chain_roofline.py $OUT "rocprofv3 [--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY] --kernel-trace -- python3 scripts/anchor_bench.py (scripts/chain_roofline.sh)" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-200
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do rm -f $OUT/$d/*kernel_trace.csv $OUT/$d/*counter_collection.csv $OUT/$d/*agent_info.csv; done
du -sh $OUT
