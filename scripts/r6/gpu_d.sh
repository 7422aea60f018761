#!/bin/bash
# round 6, visit D: parity of popoa_sysr_kernel (rows per lane), A/B on the timed step, stress set
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_parity.txt 2>&1; tail -5 $OUT/pytest_parity.txt
for v in 1 0 1 0; do
  CL_SYS_ROWS_PER_LANE=$v timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json $OUT/step_rpl_$v.json > /dev/null 2>$OUT/step_rpl_$v.err
  python - <<P
import json; d=json.load(open("$OUT/step_rpl_$v.json")); print("CL_SYS_ROWS_PER_LANE=$v: %.3f ms per step" % d["ms_per_step"]); [print("   %-28s n=%6d %8.1f us in pass" % (l["kernel"], l["n_problems"], l["in_pass_ms"]*1e3)) for l in d["launches"] if "sys" in l["kernel"]]
P
done
CL_SYS_ROWS_PER_LANE=1 timeout 300 python scripts/stress_set.py --json $OUT/stress_rpl1.json > $OUT/stress_rpl1.txt 2>&1; cat $OUT/stress_rpl1.txt | cut -c1-200
CL_SYS_ROWS_PER_LANE=0 timeout 300 python scripts/stress_set.py --json $OUT/stress_rpl0.json > $OUT/stress_rpl0.txt 2>&1; cat $OUT/stress_rpl0.txt | cut -c1-200
