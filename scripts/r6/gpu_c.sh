#!/bin/bash
# round 6, visit C: full -m gpu suite on the tree, A/B of the 128-thread systolic launches on the timed step, per-launch profile with the three clocks, chain roofline, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6c
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for v in 1 0 1 0; do
  CL_SYS_BLOCK128=$v timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json $OUT/step_b128_$v.json > /dev/null 2>$OUT/step_b128_$v.err
  python - <<P
import json; d=json.load(open("$OUT/step_b128_$v.json")); print("CL_SYS_BLOCK128=$v: %.3f ms per step" % d["ms_per_step"]); [print("   %-28s n=%6d %8.1f us in pass" % (l["kernel"], l["n_problems"], l["in_pass_ms"]*1e3)) for l in d["launches"] if "sys" in l["kernel"]]
P
done
timeout 900 bash scripts/dominant_launches.sh > $OUT/dominant.txt 2>&1; tail -22 $OUT/dominant.txt | cut -c1-200
cp gpurun_out/dom/dominant_launches.json $OUT/dominant_launches.json 2>/dev/null
cp gpurun_out/dom/trace/*kernel_stats.csv $OUT/step_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/dom/pmc_FETCH_SIZE gpurun_out/dom/pmc_WRITE_SIZE gpurun_out/dom/trace
timeout 900 bash scripts/chain_roofline.sh > $OUT/chain_roofline.txt 2>&1; tail -14 $OUT/chain_roofline.txt | cut -c1-220
cp gpurun_out/chainroof/chain_roofline.json $OUT/chain_roofline.json 2>/dev/null
du -sh $OUT gpurun_out/chainroof gpurun_out/dom
