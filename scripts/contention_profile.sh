#!/bin/bash
# Why does every chaining kernel take 2-3 x longer when four merges run side by side?  (round 6: DESIGN.md section 11, item 2.)  The 10 x 1 Mbp MSA with ONE and with FOUR worker
# contexts under rocprofv3: kernel stats (mean duration per kernel) and the HBM read counter per kernel (FETCH_SIZE: if the search structures of four DPs push one another out of
# the L2 / Infinity Cache, a far-pass launch fetches more from HBM beside three others than alone).  PMC passes carry --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/contention
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in 1 4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_w$w -o t -- python3 $R/scripts/c3_profile.py 10 1000000 $w > $OUT/trace_w$w.out 2>$OUT/trace_w$w.err
  rm -f $OUT/trace_w$w/*kernel_trace.csv
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w$w -o p -- python3 $R/scripts/c3_profile.py 10 1000000 $w > $OUT/pmc_w$w.out 2>$OUT/pmc_w$w.err
done
cd $R
python3 - <<P
import csv, glob, json, os
from collections import defaultdict
out = "$OUT"
def short(n): return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
res = {}
for w in (1, 4):
    st = {}
    for f in glob.glob(os.path.join(out, "trace_w%d" % w, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            st[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, total_ms=float(r["TotalDurationNs"]) / 1e6)
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_w%d" % w, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k in st:
            st[k]["fetch_MB_per_call_x2"] = sum(v) / len(v) * 1024 * 2 / 1e6
    res[w] = st
rows = []
for k in sorted(res[4], key=lambda k: -res[4][k]["total_ms"])[:14]:
    a, b = res[1].get(k, {}), res[4][k]
    rows.append(dict(kernel=k, calls=b["calls"], avg_us_one_worker=a.get("avg_us"), avg_us_four_workers=b["avg_us"], ratio=(b["avg_us"] / a["avg_us"]) if a.get("avg_us") else None,
                     fetch_MB_one_worker=a.get("fetch_MB_per_call_x2"), fetch_MB_four_workers=b.get("fetch_MB_per_call_x2")))
    print("%-34s calls %6d  avg %7.1f -> %7.1f us (x %.2f)   HBM read per call %8.2f -> %8.2f MB" % (k[:34], b["calls"], a.get("avg_us") or 0, b["avg_us"], rows[-1]["ratio"] or 0, a.get("fetch_MB_per_call_x2") or 0, b.get("fetch_MB_per_call_x2") or 0))
json.dump(dict(what="10 x 1 Mbp MSA under rocprofv3 with one and with four worker contexts: mean kernel duration and HBM bytes read per call (FETCH_SIZE x 2, KiB -> bytes)", rows=rows), open(os.path.join(out, "contention.json"), "w"), indent=1)
P
for d in pmc_w1 pmc_w4; do rm -f $OUT/$d/*kernel_trace.csv $OUT/$d/*counter_collection.csv; done
grep -h "^MSA" $OUT/trace_w1.out $OUT/trace_w4.out | cut -c1-60
du -sh $OUT
