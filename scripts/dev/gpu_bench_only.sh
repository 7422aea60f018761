#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5b2
timeout 900 python bench.py 2>gpurun_out/r5b2/bench.err | tee gpurun_out/r5b2/bench.json | cut -c1-500
