"""developer measurement: device time of the stitch batches of MSA merges (graph x graph subproblems -> general kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centrolign_amd import capi
from tests import helpers as H

ctx = capi.Context(0)
for name in sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("msa4_") or f.startswith("stitch4_")):
    z = np.load(os.path.join(H.GOLDEN, name))
    b = H.load_batch(z)
    plan = ctx.plan(b)
    st = plan.stats()
    for _ in range(3):
        plan.execute(); plan.sync()
    ms = []
    for _ in range(10):
        plan.execute(); ms.append(plan.sync())
    ms = float(np.median(ms))
    print("%-26s problems %5d (linear %5d) cells %9d  %.3f ms  %.2f Gcells/s" % (name, st["n_problems"], st["n_linear"], st["dp_cells"], ms, st["dp_cells"] / ms / 1e6), flush=True)
    plan.execute_profiled()
    plan.sync()
    for li in plan.launches():
        print("     %-40s problems %5d cells %8d  %.3f ms" % (li["kernel"], li["n_problems"], li["dp_cells"], li["ms"]))
    n1 = np.diff(b.side[0].node_off.astype(np.int64)); n2 = np.diff(b.side[1].node_off.astype(np.int64))
    print("     sizes: max n1 %d max n2 %d, max cells %d" % (n1.max(), n2.max(), ((n1 + 1) * (n2 + 1)).max()))
    plan.destroy()
