"""which kernel the large branching pairs of tests/test_gpu_parity.py::test_lopsided_and_large_dag_pairs take, and how long a launch lasts
(in-kernel clocks of the profiled pass).  usage: python scripts/dev/big_dag_time.py [n1xn2 ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centrolign_amd import capi, synth  # noqa: E402

sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1000, 1200), (1024, 1500), (2000, 2000), (1000, 10000), (3000, 10000), (5500, 5500)]
ctx = capi.Context(0)
for sz in sizes:
    for kw in (dict(), dict(extra_edge_p=0.02, skip_max=2)):
        b = synth.sized_dag_batch([sz], seed=5, **kw)
        plan = ctx.plan(b)
        for _ in range(3):
            plan.execute(); plan.sync()
        plan.execute_profiled(); plan.sync()
        t0 = time.perf_counter()
        plan.execute(); ms_sync = plan.sync()
        wall = (time.perf_counter() - t0) * 1e3
        for li in plan.launches():
            print("%5d x %5d %-28s %-34s %9.3f ms alone, %7.1f M cells -> %6.2f G cells/s, lds %6d B, sweep %d; execute+sync %.2f ms" %
                  (sz[0], sz[1], str(kw) if kw else "default bubbles", li["kernel"], li["ms"], li["dp_cells"] / 1e6, li["dp_cells"] / max(li["ms"], 1e-6) / 1e6, li["lds_bytes"], li["max_sweep"], wall), flush=True)
        plan.destroy()
