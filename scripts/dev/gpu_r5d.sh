#!/bin/bash
# round 5: the opt-in graph replay of a plan's pass after the lazy join (it failed inside the capture before), and the whole -m gpu suite on the final library (ABI 10)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5d2
mkdir -p $OUT
cd $R
for g in 0 1; do
  CL_STITCH_GRAPH=$g timeout 200 python scripts/step_launches.py --steps 20 --warmup 4 --json /tmp/s.json > /dev/null 2>$OUT/graph_$g.err; python -c "import json;d=json.load(open('/tmp/s.json'));print('CL_STITCH_GRAPH=$g: %.3f ms per step'%(d['ms_per_step']))"; tail -2 $OUT/graph_$g.err | cut -c1-300
done
CL_STITCH_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
