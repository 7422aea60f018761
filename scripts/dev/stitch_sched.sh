#!/bin/bash
# ms per step of the nine resident stitch plans under stream / graph / queue settings (each setting a bench run of its own)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  env "$@" python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s ms/step %.3f  value %.3g  longest launch %.3f ms  msa %.2f' % ('$*', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['msa_wall_s']))
"
}
run CL_STITCH_STREAMS=6
run CL_STITCH_STREAMS=1
run CL_STITCH_STREAMS=2
run CL_STITCH_STREAMS=3
run CL_STITCH_STREAMS=12 CL_CTX_STREAMS=12
run CL_STITCH_STREAMS=2 CL_NO_GRAPH=1
run CL_STITCH_STREAMS=6 CL_NO_GRAPH=1
run CL_STITCH_STREAMS=2 GPU_MAX_HW_QUEUES=24
