#!/bin/bash
# One command, three profiler passes (kernel trace; FETCH_SIZE; WRITE_SIZE — they do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"; PMC passes carry
# --kernel-trace only) over scripts/step_launches.py = the timed stitch step of bench.py and nothing else; scripts/dominant_launches.py joins them per LAUNCH
# -> profiles/r06_dominant_launches.json (duration, HBM bytes read x 2 + written as the guide prescribes, algorithmic bytes) which bench.py's roofline.traffic reads.
# Run via gpurun from the repo root; the program itself follows "--" (no wrapper).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dom
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 10 --warmup 3 --evented 4 --alone"
python3 $R/scripts/step_launches.py $ARGS --json $OUT/plain.json > /dev/null 2>$OUT/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/scripts/step_launches.py $ARGS --json $OUT/trace.json > /dev/null 2>$OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/scripts/step_launches.py $ARGS --json $OUT/pmc_$c.json > /dev/null 2>$OUT/pmc_$c.err
done
cd $R
python3 scripts/dominant_launches.py $OUT "rocprofv3 [--kernel-trace --stats | --pmc FETCH_SIZE --kernel-trace | --pmc WRITE_SIZE --kernel-trace] -- python3 scripts/step_launches.py $ARGS (scripts/dominant_launches.sh)" > $OUT/summary.txt 2>&1
tail -30 $OUT/summary.txt
du -sh $OUT
