"""where a long near-chain sweep's time goes: single subproblems of the 10 x 1 Mbp batches (bench_data/c3_batches.npz) alone on the device, through the lane kernel and
(CL_NO_LANE=1) the systolic kernel, with and without plane stores / traceback (CL_DEBUG_SKIP_TRACEBACK = 0 / 1 / 3): the kernel's own clock per variant.
usage: python scripts/dev/lane_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from centrolign_amd import capi  # noqa: E402
from step_launches import load_batch  # noqa: E402

batch = load_batch(os.path.join(ROOT, "bench_data", "c3_batches.npz"))
n1, n2 = batch.sizes()
ctx = capi.Context(0)
for want in ((2225, 165), (2130, 35), (441, 433), (259, 255), (68, 64)):
    k = [i for i in range(batch.n_problems) if {int(n1[i]), int(n2[i])} == set(want)]
    if not k:
        continue
    sub = batch.subset(np.array(k[:1]))
    for no_lane in ("0", "1"):
        os.environ["CL_NO_LANE"] = no_lane
        row = []
        for skip in ("0", "1", "3"):
            os.environ["CL_DEBUG_SKIP_TRACEBACK"] = skip
            plan = ctx.plan(sub)
            for _ in range(3):
                plan.execute(); plan.sync()
            best = 1e9
            for _ in range(5):
                plan.execute(); plan.sync()
                best = min(best, max(li["in_pass_ms"] for li in plan.launches()))
            kern = [li["kernel"] for li in plan.launches() if li["n_problems"]][0]
            plan.destroy()
            row.append(best)
        steps = want[0] + want[1]
        print("%-22s %-26s full %.3f ms | no traceback %.3f | no stores, no traceback %.3f ms = %.3f us per step of %d" % (str(want), kern, row[0], row[1], row[2], row[2] * 1e3 / steps, steps), flush=True)
os.environ.pop("CL_DEBUG_SKIP_TRACEBACK", None)
os.environ.pop("CL_NO_LANE", None)
