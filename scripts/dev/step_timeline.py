"""timeline of ONE timed stitch step from a rocprofv3 kernel trace of bench.py (popoa_* kernels only): the last complete step = the last burst of popoa launches
separated from the one before by a gap; prints every launch with start / end relative to the burst's first start, its queue, grid size"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "popoa_" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("::")[-1].split("(")[0], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Stream_Id", "?")) for r in rows)
# bursts: a new burst starts when a launch starts more than 300 us after every earlier launch has ended
bursts, cur, cur_end = [], [], 0
for e in ev:
    if cur and e[0] > cur_end + 300_000:
        bursts.append(cur); cur = []
    cur.append(e); cur_end = max(cur_end, e[1])
if cur: bursts.append(cur)
sizes = [len(b) for b in bursts]
print("%d bursts; launches per burst (last 12): %s" % (len(bursts), sizes[-12:]))
want = int(sys.argv[2]) if len(sys.argv) > 2 else -3
b = bursts[want]
t0 = min(e[0] for e in b)
print("burst %d: %d launches, %.3f ms from first start to last end; sum of durations %.3f ms" % (want, len(b), (max(e[1] for e in b) - t0) / 1e6, sum(e[1] - e[0] for e in b) / 1e6))
for e in sorted(b, key=lambda e: e[0]):
    print("  start %7.3f end %7.3f dur %6.3f ms  queue %-4s grid %-8s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[4], e[2]))
# the timed steps of the nine-plan layout run back to back: take the largest burst, cut it into steps of equal launch counts
if len(sys.argv) > 3:
    n_steps = int(sys.argv[3])
    big = max(bursts, key=len)
    per = len(big) // n_steps
    big.sort(key=lambda e: e[0])
    stp = big[per * (n_steps - 2):per * (n_steps - 1)]          # the last but one step
    t0 = min(e[0] for e in stp)
    print("largest burst: %d launches = %d steps of %d; step shown: %.3f ms from first start to last end, sum of durations %.3f ms, queues used %d" %
          (len(big), n_steps, per, (max(e[1] for e in stp) - t0) / 1e6, sum(e[1] - e[0] for e in stp) / 1e6, len(set(e[3] for e in stp))))
    byq = {}
    for e in stp:
        byq.setdefault(e[3], []).append(e)
    for q, es in sorted(byq.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
        print("  queue %-4s %3d launches busy %.3f ms  first start %.3f last end %.3f; longest %.3f ms (%s)" %
              (q, len(es), sum(e[1] - e[0] for e in es) / 1e6, (min(e[0] for e in es) - t0) / 1e6, (max(e[1] for e in es) - t0) / 1e6,
               max(e[1] - e[0] for e in es) / 1e6, max(es, key=lambda e: e[1] - e[0])[2]))
