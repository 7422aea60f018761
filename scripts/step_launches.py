"""The timed stitch step of bench.py alone — the stitch subproblems of the nine merges of BASELINE configs[2] as ONE resident plan, W warm-up passes, K timed passes — for the
profiler: run under `rocprofv3 --kernel-trace` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (scripts/dominant_launches.sh) its dispatches are the step's launches and nothing
else, so that a per-LAUNCH table (duration, HBM bytes, algorithmic bytes) can be made from one command (profiles/r05_dominant_launches.json).
The batches come from bench_data/c3_batches.npz (scripts/dev/dump_c3_batches.py; made here by running the MSA when the file is absent).
usage: python3 scripts/step_launches.py [--steps K] [--warmup W] [--json OUT]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centrolign_amd import capi  # noqa: E402


def load_batch(path):
    z = np.load(path)
    sides = [capi.GraphSide(**{k: (z["side%d.%s" % (si, k)] if ("side%d.%s" % (si, k)) in z.files else None) for k in capi._SIDE_DTYPES}) for si in (0, 1)]
    return capi.StitchBatch(sides[0], sides[1], z["only_deletion_alns"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--evented", type=int, default=0, help="after the timed passes: N rounds of [plain pass, pass with HIP events round every launch on its stream, plain pass] (cl_stitch_plan_execute_evented)")
    ap.add_argument("--alone", action="store_true", help="at the very end: every launch ALONE on the device (cl_stitch_plan_execute_profiled: each launched twice, the second one timed)")
    args = ap.parse_args()
    path = os.path.join(ROOT, "bench_data", "c3_batches.npz")
    if not os.path.exists(path):
        import subprocess
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "dev", "dump_c3_batches.py"), path])
    batch = load_batch(path)
    os.environ.setdefault("CL_CTX_STREAMS", "8")      # as bench.py's one-plan layout
    ctx = capi.Context(0)
    plan = ctx.plan(batch)
    for _ in range(args.warmup):
        plan.execute(); plan.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps - 1):
        plan.execute()
    plan.execute()
    enqueued = time.perf_counter() - t0     # the host is done handing the steps to the runtime: if this is the step time, the step is bound by the host
    plan.sync()
    elapsed = time.perf_counter() - t0
    st = plan.stats()
    launches = plan.launches()          # in_pass_ms: the launches' own clocks in the last timed pass
    if args.evented:
        acc = {}
        for _ in range(args.evented):
            plan.execute(); plan.execute_evented(); plan.execute(); plan.sync()
            for i, li in enumerate(plan.launches()):
                acc[i] = acc.get(i, 0.0) + li["event_ms"]
        for i, li in enumerate(launches):
            li["event_ms"] = acc.get(i, 0.0) / args.evented
    if args.alone:
        plan.execute_profiled(); plan.sync()
        for li, la in zip(launches, plan.launches()):
            li["alone_ms"] = la["ms"]
    out = dict(steps=args.steps, warmup=args.warmup, evented_rounds=args.evented, alone=bool(args.alone), ms_per_step=elapsed / args.steps * 1e3, host_enqueue_ms_per_step=enqueued / args.steps * 1e3, dp_cells=int(st["dp_cells"]), cells_per_s=st["dp_cells"] * args.steps / elapsed,
               launches=launches)
    text = json.dumps(out)
    if args.json:
        with open(args.json, "w") as f:
            f.write(text + "\n")
    print(text)
    plan.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
