"""per-problem time of the graph x graph stitch kernels on single-problem batches of chosen shapes (lopsided / square / large), with
bubbles (sized_dag_batch) — CL_NO_SYS=1 / CL_FORCE_GENERAL=1 / CL_NO_RING=1 select the older kernels for A/B comparisons, CL_DEBUG_SKIP_TRACEBACK=1 times the fill alone"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from centrolign_amd import capi, synth


def main():
    ctx = capi.Context(0)
    shapes = [(7, 2051), (30, 2087), (83, 4398), (200, 2500), (255, 256), (500, 500), (1000, 1200), (2000, 2000)]
    for p in (0.0, 0.15):
        for shp in shapes:
            b = synth.sized_dag_batch([shp], seed=3, extra_edge_p=p, n_alt=0 if p == 0.0 else 2)
            plan = ctx.plan(b)
            for _ in range(2):
                plan.execute(); plan.sync()
            ms = []
            for _ in range(5):
                plan.execute_profiled(); plan.sync()
                ms.append(sum(l["ms"] for l in plan.launches()))
            li = plan.launches()[0]
            cells = (shp[0] + 1) * (shp[1] + 1)
            print("edge_p %.2f  %5d x %5d  %-28s %8.3f ms  %7.3f us/step  %8.1f M cells/s" %
                  (p, shp[0], shp[1], li["kernel"], min(ms), min(ms) * 1e3 / (shp[0] + shp[1] + 1), cells / (min(ms) * 1e-3) / 1e6), flush=True)
            plan.destroy()


if __name__ == "__main__":
    main()
