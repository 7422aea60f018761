#!/bin/bash
# Round-6 evidence in one GPU-box visit (run via gpurun from the repo root): the -m gpu suite, the smoke run, the per-launch profile of the timed step with the three clocks
# (kernel trace + FETCH_SIZE + WRITE_SIZE passes: scripts/dominant_launches.sh) and the chaining kernels' pricing (scripts/chain_roofline.sh) — both copied into profiles/ BEFORE
# the bench so that the bench line's roofline.traffic / frac_from_committed_profile / roofline_chain quote the launches of this very tree —, the bench line, rocprofv3 kernel
# stats of the bench command, the kernel-only stress set.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6z
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 bash scripts/dominant_launches.sh > $OUT/dominant.txt 2>&1; tail -18 $OUT/dominant.txt | cut -c1-200
cp gpurun_out/dom/dominant_launches.json profiles/dominant_launches_latest.json 2>/dev/null
cp gpurun_out/dom/dominant_launches.json $OUT/dominant_launches.json 2>/dev/null
cp gpurun_out/dom/trace/*kernel_stats.csv $OUT/step_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/dom/pmc_FETCH_SIZE gpurun_out/dom/pmc_WRITE_SIZE gpurun_out/dom/trace
timeout 900 bash scripts/chain_roofline.sh > $OUT/chain_roofline.txt 2>&1; tail -14 $OUT/chain_roofline.txt | cut -c1-220
cp gpurun_out/chainroof/chain_roofline.json profiles/chain_roofline_latest.json 2>/dev/null
cp gpurun_out/chainroof/chain_roofline.json $OUT/chain_roofline.json 2>/dev/null
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-600
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r06 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_prof.json 2>$OUT/prof.err
cd $R
cp $OUT/prof/*kernel_stats.csv $OUT/c3_msa_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
timeout 300 python scripts/stress_set.py --json $OUT/stress_set.json > $OUT/stress.txt 2>&1; cat $OUT/stress.txt | cut -c1-200
du -sh $OUT
