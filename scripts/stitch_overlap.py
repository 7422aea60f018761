"""developer experiment: throughput of the C2 stitch batch with D independent passes in flight (D contexts, one plan each)"""
import os, sys, time
if len(sys.argv) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centrolign_amd import capi
import bench

b = bench.build_workload()
cells = None
for D in (1, 2, 3, 4, 6, 8):
    ctxs = [capi.Context(0) for _ in range(D)]
    plans = [c.plan(b) for c in ctxs]
    cells = plans[0].stats()["dp_cells"]
    for p in plans:
        p.execute(); p.sync()
    K = 48
    t0 = time.perf_counter()
    for k in range(K):
        plans[k % D].execute()
    for p in plans:
        p.sync()
    dt = time.perf_counter() - t0
    print("HWQ=%s D=%d: %.3f ms/step, %.1f Gcells/s" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), D, dt / K * 1e3, cells * K / dt / 1e9), flush=True)
    for p in plans:
        p.destroy()
    for c in ctxs:
        c.close()
