"""Join the three profiler passes of scripts/dominant_launches.sh into ONE per-launch table of the timed stitch step:
for every launch of the step (kernel, workgroups = subproblems, cells, algorithmic bytes — from the plan itself: cl_stitch_plan_launch_info) the duration from
`rocprofv3 --kernel-trace`, the HBM bytes from `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes; FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side, the
read side doubled on gfx950 as MI355X_MICROARCH.md "HBM" prescribes), each a mean over the timed passes of that one launch — matched by kernel name and grid size, which
identify a launch inside a step.  Writes OUT/dominant_launches.json (copied to profiles/r05_dominant_launches.json); bench.py takes roofline.traffic from the rows of its
dominant kernel.   usage: python3 scripts/dominant_launches.py OUT_DIR "command line as recorded" """
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

out = sys.argv[1]
command = sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    n = name.split("::")[-1]
    n = n.split("(")[0].strip()
    # the plan names the lane kernel's launches "<W>" / "<W, wide>": the profiler prints the template arguments as they are
    return n.replace(", false>", ">").replace(", true>", ", wide>")


def rows_of(pattern):
    files = glob.glob(os.path.join(out, pattern), recursive=True)
    return list(csv.DictReader(open(files[0]))) if files else []


plan = json.load(open(os.path.join(out, "trace.json")))
steps, warmup = plan["steps"], plan["warmup"]
# what the plan says about each launch of a step
want = {}
for li in plan["launches"]:
    if li["n_problems"]:
        want[(li["kernel"].split(" x ")[0], li["n_problems"])] = li

# ---- durations: every dispatch of the kernel trace, keyed by (kernel, workgroups) ----
dur = defaultdict(list)
for r in rows_of("trace/**/*kernel_trace.csv") or rows_of("trace/*kernel_trace.csv"):
    k = short(r["Kernel_Name"])
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    dur[(k, grid // max(1, wg))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
ctr = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    for r in rows_of("pmc_%s/**/*counter_collection.csv" % c) or rows_of("pmc_%s/*counter_collection.csv" % c):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        acc[(k, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))].append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    ctr[c] = acc


# what follows the timed passes in the command (scripts/step_launches.py --evented N --alone): N rounds of [plain, evented, plain] and, at the very end, every launch alone (twice)
n_ev, n_alone = int(plan.get("evented_rounds", 0) or 0), 2 if plan.get("alone") else 0
tail = 3 * n_ev + n_alone


def timed(vals):
    """the dispatches of the timed passes: `steps` dispatches in front of the tail (evented rounds, alone launches) — warm-up and the plan's own calibration pass come before them"""
    vals = sorted(vals)
    vals = vals[:len(vals) - tail] if tail else vals
    return [v for _, v in vals[-steps:]]


def evented(vals):
    vals = sorted(vals)
    vals = vals[len(vals) - tail:len(vals) - n_alone] if n_ev else []
    return [v for i, (_, v) in enumerate(vals) if i % 3 == 1]


def alone(vals):
    vals = sorted(vals)
    return [vals[-1][1]] if n_alone and vals else []


table = []
for (kernel, n_prob), li in sorted(want.items(), key=lambda kv: -kv[1]["dp_bytes"]):
    # strips: a launch's workgroups are strips, not subproblems — matched by kernel name alone when the name is unique in the step
    key = (kernel, n_prob)
    cand = [k for k in dur if k[0] == kernel]
    if key not in dur and len(cand) == 1:
        key = cand[0]
    d = timed(dur.get(key, []))
    f = timed(ctr["FETCH_SIZE"].get(key, []))
    w = timed(ctr["WRITE_SIZE"].get(key, []))
    row = dict(kernel=kernel, subproblems=n_prob, dp_cells=li["dp_cells"], algorithmic_bytes=li["dp_bytes"], longest_sweep=li["max_sweep"], lds_bytes=li["lds_bytes"],
               dispatches_matched=dict(trace=len(d), fetch=len(f), write=len(w)))
    if d:
        row["duration_us_mean"] = sum(d) / len(d) / 1e3
        row["duration_us_min_max"] = [min(d) / 1e3, max(d) / 1e3]
    # the other clocks of the SAME process (the run under --kernel-trace): the kernel's own clock in the last timed pass, HIP events on the launch's stream in the evented
    # passes with the trace's duration of those very dispatches, the launch alone by both clocks — what explains the gap between trace and own clock (round-5 verdict, item 2)
    if li.get("in_pass_ms"):
        row["own_clock_in_pass_us"] = li["in_pass_ms"] * 1e3
    if li.get("event_ms"):
        row["hip_events_in_pass_us"] = li["event_ms"] * 1e3
        ev = evented(dur.get(key, []))
        if ev:
            row["trace_us_of_the_evented_dispatches"] = sum(ev) / len(ev) / 1e3
    if li.get("alone_ms"):
        row["own_clock_alone_us"] = li["alone_ms"] * 1e3
        al = alone(dur.get(key, []))
        if al:
            row["trace_us_alone"] = al[0] / 1e3
    if f and w:
        fetch, write = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
        row.update(fetch_bytes_x2=fetch, write_bytes=write, hbm_bytes=fetch + write, hbm_over_algorithmic=(fetch + write) / max(1, li["dp_bytes"]))
        if d:
            row["hbm_GBps"] = (fetch + write) / (sum(d) / len(d))
            row["algorithmic_GBps"] = li["dp_bytes"] / (sum(d) / len(d))
            row["algorithmic_frac_of_8TBps"] = row["algorithmic_GBps"] / 8000.0
    table.append(row)

by_kernel = defaultdict(lambda: dict(us=0.0, alg=0, hbm=0.0, launches=0))
for r in table:
    if "duration_us_mean" in r:
        b = by_kernel[r["kernel"]]
        b["us"] += r["duration_us_mean"]; b["alg"] += r["algorithmic_bytes"]; b["hbm"] += r.get("hbm_bytes", 0.0); b["launches"] += 1
dominant = max(by_kernel, key=lambda k: by_kernel[k]["us"]) if by_kernel else None
try:
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:   # noqa: BLE001
    head = ""
res = dict(command=command, tree=head or os.environ.get("CL_TREE", "working tree (no .git on the GPU box)"),
           step=dict(steps=steps, warmup=warmup, ms_per_step_unprofiled=json.load(open(os.path.join(out, "plain.json")))["ms_per_step"] if os.path.exists(os.path.join(out, "plain.json")) else None,
                     ms_per_step_under_kernel_trace=plan["ms_per_step"], dp_cells=plan["dp_cells"]),
           dominant_kernel=dominant,
           dominant=None if dominant is None else dict(launches_per_step=by_kernel[dominant]["launches"], duration_us_sum=by_kernel[dominant]["us"],
                                                       algorithmic_bytes=by_kernel[dominant]["alg"], hbm_bytes=by_kernel[dominant]["hbm"],
                                                       algorithmic_GBps=by_kernel[dominant]["alg"] / (by_kernel[dominant]["us"] * 1e3) if by_kernel[dominant]["us"] else None,
                                                       frac_of_8TBps=by_kernel[dominant]["alg"] / (by_kernel[dominant]["us"] * 1e3) / 8000.0 if by_kernel[dominant]["us"] else None,
                                                       hbm_over_algorithmic=by_kernel[dominant]["hbm"] / max(1, by_kernel[dominant]["alg"])),
           launches=table,
           note="one row per launch of a timed step; durations and counters are means over the timed passes of THAT launch (matched by kernel name and grid size); "
                "FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes (MI355X_MICROARCH.md, HBM); algorithmic bytes = cells x 4 x (1 + 2 NumPW) of the launch's subproblems")
json.dump(res, open(os.path.join(out, "dominant_launches.json"), "w"), indent=1)
print(json.dumps(dict(dominant_kernel=dominant, dominant=res["dominant"], step=res["step"]), indent=1))
for r in table[:12]:
    print("%-28s %5d problems %10d cells  %8.1f us  alg %7.1f MB  hbm %7.1f MB  (x %.2f)" % (r["kernel"], r["subproblems"], r["dp_cells"], r.get("duration_us_mean", float("nan")),
                                                                                          r["algorithmic_bytes"] / 1e6, r.get("hbm_bytes", float("nan")) / 1e6, r.get("hbm_over_algorithmic", float("nan"))))
