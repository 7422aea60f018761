"""random large branching pairs through whatever kernel the planner picks (mostly popoa_strip_kernel) against the CPU oracle.
usage: python scripts/dev/strip_fuzz.py [cases] [seed]"""
import os
import sys
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from centrolign_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = capi.Context(0)
seen = collections.Counter()
bad = 0
for case in range(n_cases):
    k = int(rng.integers(1, 4))
    sizes = []
    for _ in range(k):
        a, b = int(rng.integers(192, 2600)), int(rng.integers(300, 5000))
        if rng.random() < 0.15:
            a, b = int(rng.integers(4096, 4600)), int(rng.integers(4096, 5200))
        sizes.append((a, b) if rng.random() < 0.5 else (b, a))
    kw = dict(extra_edge_p=float(rng.choice([0.0, 0.02, 0.1, 0.3])), skip_max=int(rng.choice([1, 2, 4, 8, 20, 40, 62])), n_alt=int(rng.integers(0, 4)),
              alphabet=int(rng.integers(2, 5)))
    if rng.random() < 0.35:   # far forks on the longer side: saved columns
        sizes = [(min(a, b2), max(a, b2) + 1) if rng.random() < 0.6 else (max(a, b2) + 1, min(a, b2)) for a, b2 in sizes]
        kw = dict(n_far=int(rng.integers(1, 10)), far_min=int(rng.choice([70, 200, 600])), far_max=2000)
        b = synth.far_fork_batch(sizes, seed=int(rng.integers(0, 1 << 30)), **kw)
    else:
        b = synth.sized_dag_batch(sizes, seed=int(rng.integers(0, 1 << 30)), **kw)
    npw = int(rng.integers(0, 4))
    f = None if npw == 0 else np.full(b.n_problems, npw, np.uint8)
    plan = ctx.plan(b, force_num_pw=f)
    for li in plan.launches():
        seen[li["kernel"].split(" x ")[0]] += 1
    plan.execute(); plan.sync()
    got = plan.collect()
    plan.destroy()
    want = po.oracle_stitch_batch(b, force_num_pw=f)
    diff = got.same_as(want)
    if diff is not None:
        bad += 1
        print("MISMATCH case %d sizes %s kw %s npw %d: %s" % (case, sizes, kw, npw, diff), flush=True)
print("%d cases, %d mismatches; kernels: %s" % (n_cases, bad, dict(seen)))
sys.exit(1 if bad else 0)
