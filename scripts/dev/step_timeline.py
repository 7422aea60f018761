"""The launches of ONE timed stitch step on a time axis: from a rocprofv3 kernel trace of scripts/step_launches.py (the last step of the run), per dispatch its kernel, workgroups,
start and end relative to the step's first dispatch, and the queue it ran on — what bounds the step when the launches overlap (the longest launch? two long ones on one stream? a tail?).
usage: python scripts/dev/step_timeline.py TRACE_DIR N_LAUNCHES_PER_STEP"""
import csv
import glob
import os
import sys

d, per = sys.argv[1], int(sys.argv[2])
f = (glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) or [None])[0]
rows = [r for r in csv.DictReader(open(f)) if "popoa" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the steps are enqueued back to back: split the dispatches into steps by dispatch order per kernel identity — simpler: the last `per` popoa dispatches by START are
# not exactly one step when steps overlap, so take the dispatches whose Dispatch_Id is among the last `per`
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
last = rows[-per:]
t0 = min(int(r["Start_Timestamp"]) for r in last)
last.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in last:
    name = r["Kernel_Name"].split("::")[-1].split("(")[0]
    wg = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) // max(1, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1))
    print("%-34s %6d wg  queue %-4s  start %8.1f us  end %8.1f us  (%7.1f)" % (name[:34], wg, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                                         (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("step span %.1f us" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e3))
# the last steps as intervals: what lies between the end of one step's last kernel and the start of the next step's first one is not kernel time
prev_end = None
for k in range(min(5, len(rows) // per), 0, -1):
    chunk = rows[len(rows) - k * per: len(rows) - (k - 1) * per]
    a, b = min(int(r["Start_Timestamp"]) for r in chunk), max(int(r["End_Timestamp"]) for r in chunk)
    print("step -%d: span %8.1f us%s" % (k, (b - a) / 1e3, "" if prev_end is None else "   gap since the previous step's last kernel %7.1f us" % ((a - prev_end) / 1e3)))
    prev_end = b
