"""predecessor structure of the branching subproblems of a stitch batch in TOPOLOGICAL-RANK space (the order the device packer uses: Kahn, LIFO stack
seeded in ascending id order): per problem the distances (in ranks) from a node to its predecessors, per side — what a register / DPP systolic kernel
has to cover.  usage: python scripts/dev/batch_structure.py BATCH.npz [min_sweep]   (golden format g1./g2. or scripts/dev/dump_c3_batches.py's side0./side1.)"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def topo_rank(n, prev_off, prev_idx, base):
    nxt = [[] for _ in range(n)]
    indeg = np.zeros(n, np.int64)
    for v in range(n):
        for e in range(int(prev_off[base + v]), int(prev_off[base + v + 1])):
            p = int(prev_idx[e])
            nxt[p].append(v)
            indeg[v] += 1
    # next lists in the order the packer derives them: by ascending successor id per predecessor (counting sort over v)
    stack = [v for v in range(n) if indeg[v] == 0]
    order = []
    while stack:
        v = stack.pop()
        order.append(v)
        for w in nxt[v]:
            indeg[w] -= 1
            if indeg[w] == 0:
                stack.append(w)
    rank = np.zeros(n, np.int64)
    rank[order] = np.arange(n)
    return rank


def side_structure(z, pre, k):
    no, po, pi = z[pre + "node_off"], z[pre + "prev_off"], z[pre + "prev_idx"]
    b, e = int(no[k]), int(no[k + 1])
    n = e - b
    rank = topo_rank(n, po, pi, b)
    so, si = z[pre + "src_off"], z[pre + "src_idx"]
    src = set(int(x) for x in si[int(so[k]):int(so[k + 1])])
    dist = Counter()
    degs = Counter()
    far = 0
    for v in range(n):
        ps = [int(pi[x]) for x in range(int(po[b + v]), int(po[b + v + 1]))]
        degs[len(ps) + (1 if v in src else 0)] += 1
        for p in ps:
            dist[int(rank[v] - rank[p])] += 1
        if v in src and rank[v] > 0:
            dist[("src", int(rank[v]) + 1)] += 1
    return n, dist, degs


def main():
    z = np.load(sys.argv[1])
    min_sweep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    pres = ("g1.", "g2.") if "g1.node_off" in z.files else ("side0.", "side1.")
    n1 = np.diff(z[pres[0] + "node_off"].astype(np.int64))
    n2 = np.diff(z[pres[1] + "node_off"].astype(np.int64))
    rows = []
    for k in np.argsort(-(n1 + n2)):
        if n1[k] == 0 or n2[k] == 0 or n1[k] + n2[k] < min_sweep:
            continue
        a = side_structure(z, pres[0], k)
        b = side_structure(z, pres[1], k)
        lin = all(set(d.keys()) <= {1} for d in (a[1], b[1]))
        if lin:
            continue
        rows.append((int(n1[k] + n2[k]), a, b))
    tot_row, tot_col = Counter(), Counter()
    for sweep, a, b in rows[:40]:
        short, long_ = (a, b) if a[0] <= b[0] else (b, a)
        fmt = lambda d: " ".join("%s:%d" % (kk, vv) for kk, vv in sorted(d.items(), key=lambda x: (isinstance(x[0], tuple), x[0])))
        print("sweep %5d | rows %4d dist{%s} deg{%s} | cols %5d dist{%s} deg{%s}" % (sweep, short[0], fmt(short[1]), fmt(short[2]), long_[0], fmt(long_[1]), fmt(long_[2])))
    for sweep, a, b in rows:
        short, long_ = (a, b) if a[0] <= b[0] else (b, a)
        for kk, vv in short[1].items():
            tot_row[kk if not isinstance(kk, tuple) else "src"] += vv
        for kk, vv in long_[1].items():
            tot_col[kk if not isinstance(kk, tuple) else "src"] += vv
    print("branching problems with sweep >= %d: %d" % (min_sweep, len(rows)))
    print("row distances:", sorted(tot_row.items(), key=lambda x: str(x[0]))[:30])
    print("col distances:", sorted(tot_col.items(), key=lambda x: str(x[0]))[:40])


if __name__ == "__main__":
    main()
