#!/bin/bash
# ms per step of the resident stitch plans under layout / graph / stream settings (each setting a bench run of its own)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  lay=$1; shift
  env "$@" python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 --plans $lay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s %-50s ms/step %.3f  value %.3g  other layout %.3f' % ('$lay', '$*', d['ms_per_step'], d['value'], d['config']['other_plan_layout']['ms_per_step']))
"
}
run one CL_NO_GRAPH=1
run one CL_NO_GRAPH=1 CL_STITCH_STREAMS=12 CL_CTX_STREAMS=12
run one CL_NO_GRAPH=1 CL_STITCH_STREAMS=3
run one CL_NO_GRAPH=0
