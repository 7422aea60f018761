#!/bin/bash
# kernel trace (start / end of every launch) of the 2 x 1 Mbp chaining DP: gpurun_out/far_trace.csv, trimmed to name, start, end
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_far -o far -- python3 $R/scripts/far_check.py /tmp/x.npz $1 > $OUT/far_run.txt 2>$OUT/prof_far.err
tail -3 $OUT/far_run.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/prof_far/far_kernel_trace.csv")))
with open("$OUT/far_trace.csv", "w") as f:
    for r in rows:
        n = r["Kernel_Name"]
        short = "far" if "far_prune" in n else "walk" if "chain_walk" in n else "seal" if "far_seal" in n else "inter" if "chain_inter" in n else "other"
        f.write("%s,%s,%s,%s\n" % (short, r["Start_Timestamp"], r["End_Timestamp"], r.get("Stream_Id", r.get("Queue_Id", ""))))
print(len(rows), "kernels")
PY
rm -f $OUT/prof_far/far_kernel_trace.csv
