"""mean of one PMC counter per launch of the kernels whose name contains a pattern, from a rocprofv3 --pmc counter_collection CSV.
usage: python scripts/dev/pmc_kernel_mean.py counter_collection.csv COUNTER pattern"""
import csv
import sys
from collections import defaultdict

path, counter, pat = sys.argv[1], sys.argv[2], sys.argv[3]
per = defaultdict(float)     # (dispatch id) -> sum over the counter's instances
name = {}
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter or pat not in r["Kernel_Name"]:
        continue
    k = r["Dispatch_Id"]
    per[k] += float(r["Counter_Value"])
    name[k] = r["Kernel_Name"].split("(")[0][-60:]
by = defaultdict(list)
for k, v in per.items():
    by[name[k]].append(v)
for n, vs in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s %6d launches, mean %.1f, max %.1f" % (n, len(vs), sum(vs) / len(vs), max(vs)))
