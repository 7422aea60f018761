#!/bin/bash
# Round-4 evidence for profiles/: (1) rocprofv3 kernel stats of the bench command (durations), (2) HBM bytes per kernel launch from the PMC
# counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# (3) the VALU / LDS issue counters of the same kernels.  PMC passes carry --kernel-trace only (no other trace domain).  The summary
# (scripts/pmc_summary.py) joins the three on the kernel name and records this command line and the tree it was taken on.
# Run via gpurun from the repo root; the program itself follows "--" (no wrapper).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=20
cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-extras --workers 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o r04 -- python3 $R/bench.py $ARGS > $OUT/prof_bench.json 2>$OUT/prof_bench.err
rm -f $OUT/prof_bench/r04_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o r04 -- python3 $R/bench.py $ARGS > $OUT/pmc_$c.json 2>$OUT/pmc_$c.err
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -o r04 -- python3 $R/bench.py $ARGS > $OUT/pmc_SQ.json 2>$OUT/pmc_SQ.err
cd $R
python3 scripts/pmc_summary.py $OUT "rocprofv3 [--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY] --kernel-trace -- python3 bench.py $ARGS (scripts/pmc_round4.sh)" | tee $OUT/pmc_summary.txt | head -80
# the per-dispatch CSVs are hundreds of MB (gpurun copies back at most 64 MiB): only the summaries travel
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do rm -f $OUT/$d/*kernel_trace.csv $OUT/$d/*counter_collection.csv; done
du -sh $OUT
