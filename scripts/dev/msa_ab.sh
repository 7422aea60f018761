#!/bin/bash
# per-merge chaining times of the 10 x 1 Mbp MSA on ONE worker context, under the environment given (A/B of walk variants etc.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py --workers ${WORKERS:-1} --no-extras --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('msa %.2f s  ms/step %.2f  sha %s' % (d['msa_wall_s'], d['ms_per_step'], d['config']['gfa_sha256'][:10]))
for m in d['msa']['per_merge']: print('   %-3d chain_ms %6.0f dev %6.0f match %5.0f  align %6.0f' % (m['chain_combinations'], m['chain_ms'], m['chain_device_ms'], m['match_ms'], m['align_ms']))
"
