#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5h
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "busy_device or overlap" 2>&1 | tail -8 | tee gpurun_out/r5h/out.txt
python -c "
from centrolign_amd import capi
print(capi.fallback_counters())"
