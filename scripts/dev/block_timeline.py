"""per-macro-block timeline of the chaining DPs from a rocprofv3 kernel trace (csv): for every walk launch, the time since the previous walk ended, split into
gap -> near launch -> gap -> walk, and whether a far launch ended inside the second gap (the walk was waiting for the far pass).
usage: python scripts/dev/block_timeline.py kernel_trace.csv"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
def kind(n):
    return "far" if "far_prune" in n else "walk" if "chain_walk" in n else "seal" if "far_seal" in n else "near" if "chain_inter" in n else None
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"]), "<false" in r["Kernel_Name"] or "chain_inter_kernel" in r["Kernel_Name"]) for r in rows if kind(r["Kernel_Name"])]
ev.sort()
for affine in (False, True):
    walks = [e for e in ev if e[2] == "walk" and ("<false>" in "" or True) and e[3] == affine]
    nears = [e for e in ev if e[2] == "near" and e[3] == affine]
    fars = sorted(e[1] for e in ev if e[2] == "far" and e[3] == affine)
    if len(walks) < 50:
        continue
    fars = np.array(fars)
    near_by_start = sorted(nears)
    ns = np.array([e[0] for e in near_by_start])
    out = []
    for i in range(1, len(walks)):
        w0, w1 = walks[i - 1], walks[i]
        if w1[0] - w0[1] > 2_000_000:      # another DP / host gap
            continue
        j = np.searchsorted(ns, w0[1] - 1000)
        n = near_by_start[j] if j < len(ns) and ns[j] < w1[0] else None
        if n is None:
            continue
        gap1, near, gap2, walk = n[0] - w0[1], n[1] - n[0], w1[0] - n[1], w1[1] - w1[0]
        k = np.searchsorted(fars, w1[0]) - 1
        far_wait = k >= 0 and fars[k] > n[1] and w1[0] - fars[k] < 30_000   # a far launch ended after the near launch and just before the walk started
        out.append((gap1, near, gap2, walk, far_wait, fars[k] - n[1] if k >= 0 else 0, w1[0] - fars[k] if k >= 0 else 0))
    a = np.array([o[:4] for o in out], float) / 1e3
    fw = np.array([o[4] for o in out])
    print("%s DPs: %d macro-blocks; per block mean us: gap %.1f | near %.1f | gap %.1f | walk %.1f | total %.1f" % ("affine" if affine else "sparse", len(out), *a.mean(0), a.sum(1).mean()))
    print("   walk waited for a far launch in %.0f %% of the blocks; second gap when it did: %.1f us, when not: %.1f us" % (100 * fw.mean(), a[fw, 2].mean() if fw.any() else 0, a[~fw, 2].mean() if (~fw).any() else 0))
    print("   percentiles of the second gap (10/50/90): %s" % np.percentile(a[:, 2], [10, 50, 90]).round(1))
    wake = np.array([o[6] for o in out], float)[fw] / 1e3
    if len(wake):
        print("   when the walk waited for a far launch: far end -> walk start (10/50/90 us): %s" % np.percentile(wake, [10, 50, 90]).round(1))
