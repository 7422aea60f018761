"""developer experiment: device time of the C2 stitch batch by problem class (which part bounds the pass?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centrolign_amd import capi
import bench

ctx = capi.Context(0)
b = bench.build_workload()
n1 = np.diff(b.side[0].node_off.astype(np.int64)); n2 = np.diff(b.side[1].node_off.astype(np.int64))
short, long_ = np.minimum(n1, n2), np.maximum(n1, n2)
cells = (n1 + 1) * (n2 + 1)
po = (n1 > 0) & (n2 > 0)


def timeit(idx, label):
    if len(idx) == 0:
        print(label, "empty"); return
    sb = b.subset(idx)
    plan = ctx.plan(sb)
    st = plan.stats()
    for _ in range(3):
        plan.execute(); plan.sync()
    ms = []
    for _ in range(10):
        plan.execute(); ms.append(plan.sync())
    ms = float(np.median(ms))
    print("%-34s n=%6d cells=%10d  %.3f ms  %.1f Gcells/s" % (label, len(idx), st["dp_cells"], ms, st["dp_cells"] / ms / 1e6), flush=True)
    plan.destroy()


order = np.argsort(-cells)
timeit(order[:1], "largest 1 (%dx%d)" % (n1[order[0]], n2[order[0]]))
timeit(order[:8], "largest 8")
timeit(np.nonzero(po & (short > 256))[0], "W16 class (short>256)")
timeit(np.nonzero(po & (short > 128) & (short <= 256) | po & (short > 64) & (short <= 128) & (long_ >= 300))[0], "W4 class")
timeit(np.nonzero(po & (short > 64) & (short <= 128) & (long_ < 300))[0], "W1 R2 class")
timeit(np.nonzero(po & (short <= 64))[0], "W1 R1 class (short<=64)")
timeit(np.nonzero(po & (short <= 64) & (long_ <= 64))[0], "  of which long<=64")
timeit(np.nonzero(po & (short <= 64) & (long_ > 64))[0], "  of which long>64")
timeit(np.arange(b.n_problems), "all")
for q in (0.5, 0.9, 0.99, 1.0):
    print("quantile %.2f: short %d long %d" % (q, np.quantile(short[po], q), np.quantile(long_[po], q)))
