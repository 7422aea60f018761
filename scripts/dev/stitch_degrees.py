"""in-degree structure of the longest stitch subproblems of the 10 x 1 Mbp MSA: how many rows (shorter side) / columns have a predecessor count
(a source's boundary counted as one) outside {1, 2} — those cells leave popoa_sys_kernel's straight-line path"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth
from bench import stitch_batches
ctx = capi.Context(0)
names, seqs, tree = synth.c3_workload(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000)
r = msa.progressive_msa(ctx, seqs, tree, workers=4, keep_merges=True)
rows = []
for m, b in stitch_batches(r["stats"]["kept"]):
    n1, n2 = b.sizes()
    for k in np.argsort(-(n1 + n2))[:6]:
        if n1[k] == 0 or n2[k] == 0:
            continue
        out = []
        for s in b.side:
            lo, hi = int(s.node_off[k]), int(s.node_off[k + 1])
            deg = np.diff(s.prev_off[lo:hi + 1].astype(np.int64))
            src = np.zeros(hi - lo, np.int64)
            src[s.src_idx[int(s.src_off[k]):int(s.src_off[k + 1])].astype(np.int64)] = 1
            nq = deg + src
            out.append((hi - lo, int((nq > 2).sum()), int((nq == 0).sum()), int(deg.max())))
        rows.append((int(n1[k] + n2[k]), m[:24], out))
rows.sort(key=lambda x: -x[0])
for sweep, m, out in rows[:25]:
    print("sweep %5d  %-24s  graph1: %5d nodes, %3d with > 2 preds (max in-degree %d) | graph2: %5d nodes, %3d with > 2 preds (max in-degree %d)" %
          (sweep, m, out[0][0], out[0][1], out[0][3], out[1][0], out[1][1], out[1][3]))
