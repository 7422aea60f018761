"""Synthetic inputs for the stitch path (no real centromeres are available).

* `hor_sequences`      — HOR-like tandem arrays: 171-bp monomer x 12-mer HOR, tiled and mutated (the family
                         BASELINE.md's CPU numbers were measured on).
* `random_dag_batch`   — random graph pairs with bubbles / multiple sources and sinks, for parity tests.
* `linear_batch`       — pairwise (chain x chain) stitch subproblems of chosen sizes.
* `hor_stitch_batch`   — the between-anchor subproblem batch of a synthetic HOR pair, derived from the known
                         true alignment of the simulated pair: maximal exact-match runs >= `min_anchor` play
                         the anchors, the stretches between them are the stitch subproblems.  It reproduces
                         the shape of the batch the reference extracts for the same pair (thousands of small
                         chain x chain matrices, a few hundred large ones) without needing the
                         reference's match finder and chainer on the GPU box.
"""
import random

import numpy as np

from .capi import GraphSide, StitchBatch

_B = "ACGT"
_ENC = {"A": 1, "C": 2, "G": 3, "T": 4}  # any injective code works: the DP only compares labels


def encode(seq):
    lut = np.zeros(256, dtype=np.uint8)
    for c, v in _ENC.items():
        lut[ord(c)] = v
    return lut[np.frombuffer(seq.encode(), dtype=np.uint8)]


def _mut_track(rng, s, r):
    """mutate s at rate r (80% substitution, 10% deletion, 10% insertion); also returns, for every output
    position, the index in s it descends from (or -1 for inserted bases)"""
    out, src = [], []
    for i, c in enumerate(s):
        x = rng.random()
        if x < r * 0.8:
            out.append(rng.choice(_B)); src.append(i)
        elif x < r * 0.9:
            continue
        elif x < r:
            out.append(c); src.append(i)
            out.append(rng.choice(_B)); src.append(-1)
        else:
            out.append(c); src.append(i)
    return out, src


def hor_sequences(seed, total_len, n_seq, mono_len=171, hor_n=12, mono_div=0.25, hor_div=0.02, seq_div=0.005,
                  indel_hor=2, track=False):
    """n_seq HOR-array sequences descending from one ancestor array.  With track=True also returns per
    sequence the ancestor coordinate of every base (-1 = inserted), which gives the true alignment."""
    rng = random.Random(seed)
    base = [rng.choice(_B) for _ in range(mono_len)]
    monos = [_mut_track(rng, base, mono_div)[0] for _ in range(hor_n)]
    hor = [c for m in monos for c in m]
    n_hor = total_len // len(hor) + 1
    anc = []
    for _ in range(n_hor):
        anc.extend(_mut_track(rng, hor, hor_div)[0])
    anc = anc[:total_len]
    seqs, srcs = [], []
    for _ in range(n_seq):
        s, src = _mut_track(rng, anc, seq_div)
        for _ in range(indel_hor):
            p = rng.randrange(0, max(1, len(s) - len(hor)))
            p -= p % len(hor)
            if rng.random() < 0.5:
                s = s[:p] + s[p + len(hor):]
                src = src[:p] + src[p + len(hor):]
            else:
                s = s[:p] + s[p:p + len(hor)] + s[p:]
                src = src[:p] + [-1] * len(hor) + src[p:]
        seqs.append("".join(s))
        srcs.append(np.array(src, dtype=np.int64))
    return (seqs, srcs) if track else seqs


def tandem_dup_sequences(seed, total_len, n_seq, dup_len, dup_at=None, dup_div=0.003, carriers=None, **kw):
    """hor_sequences in which some sequences carry a recent tandem duplication: a copy of [dup_at, dup_at + dup_len), mutated at rate
    dup_div, inserted right behind the original (inputs for the cyclisation path, the CLI's -c)"""
    seqs = hor_sequences(seed, total_len, n_seq, **kw)
    rng = random.Random(seed * 7919 + 13)
    carriers = range(n_seq) if carriers is None else carriers
    out = list(seqs)
    for i in carriers:
        s = out[i]
        a = (len(s) // 3 if dup_at is None else dup_at)
        a = max(0, min(a, len(s) - dup_len))
        copy = "".join(_mut_track(rng, list(s[a:a + dup_len]), dup_div)[0])
        out[i] = s[:a + dup_len] + copy + s[a + dup_len:]
    return out


def write_fasta(path, seqs, names=None):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">%s\n" % (names[i] if names else "seq%d" % i))
            for j in range(0, len(s), 80):
                f.write(s[j:j + 80] + "\n")


class _SideBuilder:
    def __init__(self):
        self.node_off = [0]
        self.label, self.prev_off, self.prev_idx = [], [0], []
        self.next_off, self.next_idx = [0], []
        self.src_off, self.src_idx, self.snk_off, self.snk_idx = [0], [], [0], []
        self.back = []
        self._ne = 0
        self._np = 0

    def add_chain(self, labels, back0):
        """a simple path graph: node i -> i+1; source = node 0, sink = last node"""
        n = len(labels)
        self.label.append(np.asarray(labels, dtype=np.uint8))
        ne = max(n - 1, 0)
        self.prev_idx.append(np.arange(0, ne, dtype=np.uint32))
        self.next_idx.append(np.arange(1, n, dtype=np.uint32))
        # cumulative list lengths after each node: prev -> v, next -> min(v+1, n-1)
        self.prev_off.append(self._np + np.arange(0, n, dtype=np.int64))
        self.next_off.append(self._ne + np.minimum(np.arange(1, n + 1, dtype=np.int64), ne))
        self._np += ne
        self._ne += ne
        self.node_off.append(self.node_off[-1] + n)
        if n:
            self.src_idx.append(0); self.snk_idx.append(n - 1)
        self.src_off.append(len(self.src_idx)); self.snk_off.append(len(self.snk_idx))
        self.back.append(np.arange(back0, back0 + n, dtype=np.uint64))

    def add_graph(self, labels, edges, sources, sinks, back=None):
        """general DAG: edges is the edge-insertion sequence [(u, v)...] (defines previous()/next() order)"""
        n = len(labels)
        prevs = [[] for _ in range(n)]
        nexts = [[] for _ in range(n)]
        for u, v in edges:
            prevs[v].append(u); nexts[u].append(v)
        self.label.append(np.asarray(labels, dtype=np.uint8))
        po, no = [], []
        for v in range(n):
            self._np += len(prevs[v]); po.append(self._np)
            self._ne += len(nexts[v]); no.append(self._ne)
        self.prev_idx.append(np.array([u for l in prevs for u in l], dtype=np.uint32))
        self.next_idx.append(np.array([u for l in nexts for u in l], dtype=np.uint32))
        self.prev_off.append(np.array(po, dtype=np.int64)); self.next_off.append(np.array(no, dtype=np.int64))
        self.node_off.append(self.node_off[-1] + n)
        self.src_idx.extend(sources); self.snk_idx.extend(sinks)
        self.src_off.append(len(self.src_idx)); self.snk_off.append(len(self.snk_idx))
        self.back.append(np.arange(n, dtype=np.uint64) if back is None else np.asarray(back, dtype=np.uint64))

    def finish(self):
        cat = lambda parts, dt: (np.concatenate([np.asarray(p, dtype=dt).ravel() for p in parts])
                                 if parts else np.zeros(0, dt))
        return GraphSide(
            node_off=np.array(self.node_off, np.uint64), label=cat(self.label, np.uint8),
            prev_off=np.concatenate([[0], cat(self.prev_off[1:], np.int64)]).astype(np.uint64),
            prev_idx=cat(self.prev_idx, np.uint32),
            next_off=np.concatenate([[0], cat(self.next_off[1:], np.int64)]).astype(np.uint64),
            next_idx=cat(self.next_idx, np.uint32),
            src_off=np.array(self.src_off, np.uint64), src_idx=np.array(self.src_idx, np.uint32),
            snk_off=np.array(self.snk_off, np.uint64), snk_idx=np.array(self.snk_idx, np.uint32),
            back_translation=cat(self.back, np.uint64))


def linear_batch(sizes, seed=0, divergence=0.1):
    """chain x chain problems; sizes = [(n1, n2), ...]; second sequence is a mutated copy of the first"""
    rng = np.random.default_rng(seed)
    b1, b2 = _SideBuilder(), _SideBuilder()
    for n1, n2 in sizes:
        a = rng.integers(1, 5, size=n1, dtype=np.uint8)
        if n2 <= n1:
            b = a[:n2].copy()
        else:
            b = np.concatenate([a, rng.integers(1, 5, size=n2 - n1, dtype=np.uint8)])
        mut = rng.random(n2) < divergence
        b[mut] = rng.integers(1, 5, size=int(mut.sum()), dtype=np.uint8)
        b1.add_chain(a, 0); b2.add_chain(b, 0)
    return StitchBatch(b1.finish(), b2.finish(), np.zeros(len(sizes), np.uint8))


def _random_dag(rng, n, extra_edge_p, skip_max, n_alt_src, n_alt_snk, alphabet):
    """a connected DAG on n nodes given in a random (non-topological) id order.  Built as a backbone chain plus
    forward skip edges, then ids are shuffled; the edge insertion order is shuffled too, so previous() orders
    are arbitrary.  Every node lies on a source->sink walk, as in extracted stitch graphs
    (include/centrolign/subgraph_extraction.hpp:85-91)."""
    topo_edges = [(i, i + 1) for i in range(n - 1)]
    for i in range(n):
        if rng.random() < extra_edge_p:
            j = i + 2 + int(rng.integers(0, skip_max))
            if j < n:
                topo_edges.append((i, j))
    topo_edges = list(dict.fromkeys(topo_edges))
    perm = rng.permutation(n)  # topo position -> node id
    order = rng.permutation(len(topo_edges))
    edges = [(int(perm[topo_edges[k][0]]), int(perm[topo_edges[k][1]])) for k in order]
    sources = [int(perm[0])]
    sinks = [int(perm[n - 1])]
    # extra sources/sinks: like nodes adjacent to the anchor end through a bubble
    for _ in range(n_alt_src):
        c = int(perm[int(rng.integers(0, max(1, min(n, 4))))])
        if c not in sources:
            sources.append(c)
    for _ in range(n_alt_snk):
        c = int(perm[n - 1 - int(rng.integers(0, max(1, min(n, 4))))])
        if c not in sinks:
            sinks.append(c)
    rng.shuffle(sources); rng.shuffle(sinks)
    labels = rng.integers(1, 1 + alphabet, size=n, dtype=np.uint8)
    return labels, edges, [int(s) for s in sources], [int(s) for s in sinks]


def random_dag_batch(n_problems, seed=0, max_n=40, extra_edge_p=0.3, skip_max=4, alphabet=4, allow_empty=True,
                     related=True):
    """random graph pairs; graph2 is (usually) a relabelled/perturbed sibling of graph1 so alignments have
    long diagonals and plenty of score ties"""
    rng = np.random.default_rng(seed)
    b1, b2 = _SideBuilder(), _SideBuilder()
    for k in range(n_problems):
        n1 = int(rng.integers(0 if allow_empty else 1, max_n + 1))
        n2 = int(rng.integers(0 if allow_empty else 1, max_n + 1))
        if allow_empty and rng.random() < 0.05:
            n2 = 0
        for bld, n in ((b1, n1), (b2, n2)):
            if n == 0:
                bld.add_graph(np.zeros(0, np.uint8), [], [], [], None)
                continue
            lab, edges, src, snk = _random_dag(rng, n, extra_edge_p, skip_max, int(rng.integers(0, 3)),
                                               int(rng.integers(0, 3)), alphabet)
            bld.add_graph(lab, edges, src, snk, rng.integers(0, 1 << 40, size=n, dtype=np.uint64))
    return StitchBatch(b1.finish(), b2.finish(), (rng.random(n_problems) < 0.2).astype(np.uint8))


def sized_dag_batch(sizes, seed=0, extra_edge_p=0.15, skip_max=4, alphabet=4, n_alt=2):
    """graph pairs with bubbles of the given (n1, n2) sizes — lopsided and large matrices for the graph x graph kernels"""
    rng = np.random.default_rng(seed)
    b1, b2 = _SideBuilder(), _SideBuilder()
    for n1, n2 in sizes:
        for bld, n in ((b1, n1), (b2, n2)):
            lab, edges, src, snk = _random_dag(rng, n, extra_edge_p, skip_max, int(rng.integers(0, n_alt + 1)), int(rng.integers(0, n_alt + 1)), alphabet)
            bld.add_graph(lab, edges, src, snk, rng.integers(0, 1 << 40, size=n, dtype=np.uint64))
    return StitchBatch(b1.finish(), b2.finish(), np.zeros(len(sizes), np.uint8))


def far_fork_batch(sizes, seed=0, n_far=3, far_min=200, far_max=1500, alphabet=4):
    """graph pairs (n1 < n2 expected) whose SECOND graph has a few long-range skip edges — the fork in front of a bubble whose branches differ by a whole repeat
    unit — on top of short bubbles in both: what the saved columns of the systolic and the strip kernel are for"""
    rng = np.random.default_rng(seed)
    b1, b2 = _SideBuilder(), _SideBuilder()
    for n1, n2 in sizes:
        lab, edges, src, snk = _random_dag(rng, n1, 0.05, 3, 0, 0, alphabet)
        b1.add_graph(lab, edges, src, snk, rng.integers(0, 1 << 40, size=n1, dtype=np.uint64))
        # second graph: built in topological positions, then shuffled like _random_dag does
        topo = [(i, i + 1) for i in range(n2 - 1)]
        for i in range(n2):
            if rng.random() < 0.05:
                j = i + 2 + int(rng.integers(0, 3))
                if j < n2:
                    topo.append((i, j))
        for _ in range(n_far):
            i = int(rng.integers(0, max(1, n2 - far_min - 2)))
            j = min(n2 - 1, i + int(rng.integers(far_min, far_max)))
            topo.append((i, j))
        topo = list(dict.fromkeys(topo))
        perm = rng.permutation(n2)
        order = rng.permutation(len(topo))
        edges2 = [(int(perm[topo[k][0]]), int(perm[topo[k][1]])) for k in order]
        lab2 = rng.integers(1, 1 + alphabet, size=n2, dtype=np.uint8)
        b2.add_graph(lab2, edges2, [int(perm[0])], [int(perm[n2 - 1])], rng.integers(0, 1 << 40, size=n2, dtype=np.uint64))
    return StitchBatch(b1.finish(), b2.finish(), np.zeros(len(sizes), np.uint8))


def _bubble_graph_edges(rng, n, p_snp, p_del, n_long, long_min, long_max, alphabet, long_other=None):
    """a DAG of about n nodes as a progressive MSA makes them: a chain interrupted by SNP bubbles (fork -> one of two single nodes -> join), short deletion bubbles (a stretch
    of 1-3 nodes that an edge jumps over) and n_long LONG bubbles (two branches of long_min .. long_max nodes: the whole-repeat-unit indel, whose fork is read from a repeat
    unit further on).  Built in topological positions; returns (labels, edges, sources, sinks) with ids shuffled as _random_dag does"""
    edges, tails, k = [], None, 0            # tails: nodes the next node hangs under
    def node():
        nonlocal k
        k += 1
        return k - 1
    long_at = sorted(int(x) for x in rng.integers(0, max(1, n), n_long))
    while k < n or tails is None:
        here = k
        if long_at and here >= long_at[0]:
            long_at.pop(0)
            la, lb = int(rng.integers(long_min, long_max + 1)), int(rng.integers(0, 4))   # one branch a repeat unit long, the other short or empty (an indel)
            if long_other is not None:   # ... or BOTH long (two diverged copies of the unit): the other branch differs by up to long_other nodes
                lb = max(1, la + int(rng.integers(-long_other, long_other + 1)))
            fork = tails
            ends = []
            for length in (la, lb):
                prev = fork
                for _ in range(length):
                    v = node()
                    for u in (prev or []):
                        edges.append((u, v))
                    prev = [v]
                ends += prev if length else (fork or [])
            tails = ends if fork is not None or ends else None
            if tails is None:
                continue
            continue
        x = rng.random()
        if x < p_snp and tails is not None:
            a, b = node(), node()
            for u in tails:
                edges.append((u, a)); edges.append((u, b))
            tails = [a, b]
        elif x < p_snp + p_del and tails is not None:
            skip_from = tails
            prev = tails
            for _ in range(int(rng.integers(1, 4))):
                v = node()
                for u in prev:
                    edges.append((u, v))
                prev = [v]
            tails = list(dict.fromkeys(prev + skip_from))
        else:
            v = node()
            for u in (tails or []):
                edges.append((u, v))
            tails = [v]
    v = node()                                # one closing node: a single sink
    for u in tails:
        edges.append((u, v))
    n_nodes = k
    indeg = np.zeros(n_nodes, np.int64)
    for _, w in edges:
        indeg[w] += 1
    sources = [i for i in range(n_nodes) if indeg[i] == 0]
    perm = rng.permutation(n_nodes)
    order = rng.permutation(len(edges))
    edges = list(dict.fromkeys((int(perm[edges[i][0]]), int(perm[edges[i][1]])) for i in order))
    labels = rng.integers(1, 1 + alphabet, size=n_nodes, dtype=np.uint8)
    return labels, edges, [int(perm[i]) for i in sources], [int(perm[n_nodes - 1])]


def near_chain_batch(sizes, seed=0, p_snp=0.04, p_del=0.03, n_long=(0, 1), long_min=150, long_max=400, alphabet=4, related=True, long_other=None):
    """graph pairs shaped like the long stitch subproblems of a progressive MSA over HOR arrays (scripts/dev/batch_structure.py on the 10 x 1 Mbp batches): chains with
    SNP and short deletion bubbles (predecessors 1-3 ranks back) and, in the second graph of a pair, n_long[1] long bubbles; what popoa_lane_kernel is for"""
    rng = np.random.default_rng(seed)
    b1, b2 = _SideBuilder(), _SideBuilder()
    for n1, n2 in sizes:
        for bld, n, nl in ((b1, n1, n_long[0]), (b2, n2, n_long[1])):
            lab, edges, src, snk = _bubble_graph_edges(rng, n, p_snp, p_del, nl, long_min, long_max, alphabet, long_other)
            bld.add_graph(lab, edges, src, snk, rng.integers(0, 1 << 40, size=len(lab), dtype=np.uint64))
    return StitchBatch(b1.finish(), b2.finish(), np.zeros(len(sizes), np.uint8))


def hor_stitch_batch(seed, total_len, min_anchor=20, seq_div=0.005, hor_div=0.02, indel_hor=2, max_cells=40000000):
    """between-anchor subproblems of a simulated HOR pair (see module docstring).  Returns (batch, info)."""
    (s1, s2), (a1, a2) = hor_sequences(seed, total_len, 2, seq_div=seq_div, hor_div=hor_div, indel_hor=indel_hor,
                                       track=True)
    e1, e2 = encode(s1), encode(s2)
    # true alignment: positions of the two sequences that descend from the same ancestor base
    pos1 = np.flatnonzero(a1 >= 0); pos2 = np.flatnonzero(a2 >= 0)
    anc1, anc2 = a1[pos1], a2[pos2]
    # a HOR duplication keeps ancestry only for one copy, so ancestor coordinates are strictly increasing
    common, i1, i2 = np.intersect1d(anc1, anc2, assume_unique=True, return_indices=True)
    p1, p2 = pos1[i1], pos2[i2]
    good = e1[p1] == e2[p2]
    # maximal runs of consecutive (p1+1, p2+1) matching pairs
    brk = np.ones(len(p1), dtype=bool)
    brk[1:] = (np.diff(p1) != 1) | (np.diff(p2) != 1) | ~good[1:] | ~good[:-1]
    run_id = np.cumsum(brk) - 1
    run_start = np.flatnonzero(brk)
    run_len = np.diff(np.concatenate([run_start, [len(p1)]]))
    keep = (run_len >= min_anchor) & good[run_start]
    anchors = [(int(p1[s]), int(p2[s]), int(l)) for s, l, k in zip(run_start, run_len, keep) if k]
    b1, b2 = _SideBuilder(), _SideBuilder()
    only_del = []
    prev1, prev2 = 0, 0
    n_skipped = 0
    gaps = []
    for (x1, x2, l) in anchors + [(len(e1), len(e2), 0)]:
        g1, g2 = (prev1, x1), (prev2, x2)
        gaps.append((g1, g2))
        prev1, prev2 = x1 + l, x2 + l
    for idx, (g1, g2) in enumerate(gaps):
        n1, n2 = g1[1] - g1[0], g2[1] - g2[0]
        if (n1 + 1) * (n2 + 1) > max_cells:
            n_skipped += 1
            continue
        b1.add_chain(e1[g1[0]:g1[1]], g1[0]); b2.add_chain(e2[g2[0]:g2[1]], g2[0])
        only_del.append(1 if idx in (0, len(gaps) - 1) else 0)
    batch = StitchBatch(b1.finish(), b2.finish(), np.array(only_del, np.uint8))
    info = dict(n_anchors=len(anchors), len1=len(e1), len2=len(e2), n_skipped=n_skipped)
    return batch, info


def batch_from_intervals(seq1, seq2, intervals, only_del=None):
    """chain x chain stitch batch from (start1, len1, start2, len2) rows over two sequences; this is how the
    between-anchor subproblems of a PAIRWISE alignment look (every subgraph is a stretch of one sequence)."""
    e1 = encode(seq1) if isinstance(seq1, str) else np.asarray(seq1, np.uint8)
    e2 = encode(seq2) if isinstance(seq2, str) else np.asarray(seq2, np.uint8)
    iv = np.asarray(intervals, dtype=np.int64)
    sides = []
    for e, st, ln in ((e1, iv[:, 0], iv[:, 1]), (e2, iv[:, 2], iv[:, 3])):
        n = len(ln)
        node_off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        tot = int(node_off[-1])
        within = np.arange(tot, dtype=np.int64) - np.repeat(node_off[:-1], ln)
        pos = np.repeat(st, ln) + within
        ne = np.maximum(ln - 1, 0)
        edge_off = np.concatenate([[0], np.cumsum(ne)]).astype(np.int64)
        etot = int(edge_off[-1])
        ewithin = np.arange(etot, dtype=np.int64) - np.repeat(edge_off[:-1], ne)
        # node v of problem k: prev list = [v-1] for v >= 1 -> cumulative count after node v is edge_off[k] + v
        prev_off = np.concatenate([[0], np.repeat(edge_off[:-1], ln) + within]).astype(np.uint64)
        next_off = np.concatenate([[0], np.repeat(edge_off[:-1], ln) + np.minimum(within + 1, np.repeat(ne, ln))]).astype(np.uint64)
        has = (ln > 0).astype(np.int64)
        one_off = np.concatenate([[0], np.cumsum(has)]).astype(np.uint64)
        sides.append(GraphSide(
            node_off=node_off.astype(np.uint64), label=e[pos] if tot else np.zeros(0, np.uint8),
            prev_off=prev_off, prev_idx=ewithin.astype(np.uint32),
            next_off=next_off, next_idx=(ewithin + 1).astype(np.uint32),
            src_off=one_off, src_idx=np.zeros(int(has.sum()), np.uint32),
            snk_off=one_off, snk_idx=(ln[ln > 0] - 1).astype(np.uint32),
            back_translation=pos.astype(np.uint64)))
    od = np.zeros(len(iv), np.uint8) if only_del is None else np.asarray(only_del, np.uint8)
    return StitchBatch(sides[0], sides[1], od)


def base_graph_from_sequence(seq, sentinels=(5, 6)):
    """capi.BaseGraph of one sequence as the reference builds a leaf subproblem: one node per base in a chain, one path over
    them, plus the two sentinel nodes of add_sentinels (source before the first base, sink after the last; labels 5 and 6,
    or 7 and 8 for the second graph of a merge)"""
    from . import capi
    e = encode(seq) if isinstance(seq, str) else np.asarray(seq, np.uint8)
    n = len(e)
    src, snk = n, n + 1
    label = np.concatenate([e, list(sentinels)]).astype(np.uint8)
    nxt = [[i + 1] for i in range(n - 1)] + ([[snk]] if n else []) + [[0] if n else [snk], []]
    prv = ([[src]] if n else []) + [[i - 1] for i in range(1, n)] + [[], [n - 1] if n else [src]]
    def csr(lists):
        off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
        idx = np.array([v for x in lists for v in x], np.uint32)
        return off, idx
    no, ni = csr(nxt)
    po_, pi = csr(prv)
    return capi.BaseGraph(label, no, ni, po_, pi, np.array([0, n], np.uint64), np.arange(n, dtype=np.uint32), src, snk)


def bubble_graph(ancestor, n_paths, seed=0, alt_p=0.05, skip_p=0.02, sentinels=(5, 6)):
    """capi.BaseGraph shaped like a merged subproblem: n_paths embedded paths over a DAG made of the ancestor's chain plus
    bubbles.  Each path follows the ancestor, at some positions stepping through an alternative node (one per position and
    base, shared by the paths that take it) or skipping the position; edges are the path adjacencies, the sentinels hang
    before / after the path ends."""
    from . import capi
    rng = np.random.default_rng(seed)
    anc = encode(ancestor) if isinstance(ancestor, str) else np.asarray(ancestor, np.uint8)
    n = len(anc)
    labels = list(anc)
    alt = {}
    paths = []
    for _ in range(n_paths):
        p = []
        for i in range(n):
            r = rng.random()
            if r < skip_p and 0 < i < n - 1:
                continue
            if r < skip_p + alt_p:
                b = int((anc[i] + 1 + rng.integers(0, 3)) % 4)
                if (i, b) not in alt:
                    alt[(i, b)] = len(labels)
                    labels.append(b)
                p.append(alt[(i, b)])
            else:
                p.append(i)
        paths.append(p)
    # keep node ids topologically ordered: ancestor position first, alternatives after the position they shadow
    key = {v: (v, 0) for v in range(n)}
    for (i, b), v in alt.items():
        key[v] = (i, 1 + b)
    used = sorted({v for p in paths for v in p}, key=lambda v: key[v])
    remap = {v: k for k, v in enumerate(used)}
    m = len(used)
    src, snk = m, m + 1
    label = np.array([labels[v] for v in used] + list(sentinels), np.uint8)
    nxt = [[] for _ in range(m + 2)]
    prv = [[] for _ in range(m + 2)]
    def edge(a, b):
        if b not in nxt[a]:
            nxt[a].append(b)
            prv[b].append(a)
    path_nodes, path_off = [], [0]
    for p in paths:
        q = [remap[v] for v in p]
        edge(src, q[0])
        for a, b in zip(q[:-1], q[1:]):
            edge(a, b)
        edge(q[-1], snk)
        path_nodes.extend(q)
        path_off.append(len(path_nodes))
    def csr(lists):
        off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
        idx = np.array([v for x in lists for v in x], np.uint32)
        return off, idx
    no, ni = csr(nxt)
    po_, pi = csr(prv)
    return capi.BaseGraph(label, no, ni, po_, pi, np.array(path_off, np.uint64), np.array(path_nodes, np.uint32), src, snk)


def exact_matches(seq1, seq2, k=12, max_sets=None):
    """capi.MatchSets of the shared k-mers of two sequences (one set per k-mer, every occurrence a walk): a simple stand-in
    for the match finder, enough to drive the chaining seam in small tests"""
    from . import capi
    e1 = encode(seq1) if isinstance(seq1, str) else np.asarray(seq1, np.uint8)
    e2 = encode(seq2) if isinstance(seq2, str) else np.asarray(seq2, np.uint8)
    def index(e):
        d = {}
        for i in range(len(e) - k + 1):
            d.setdefault(e[i:i + k].tobytes(), []).append(i)
        return d
    d1, d2 = index(e1), index(e2)
    so1, wo1, n1, so2, wo2, n2, c1, c2, fl = [0], [0], [], [0], [0], [], [], [], []
    for key in sorted(set(d1) & set(d2)):
        if max_sets is not None and len(c1) >= max_sets:
            break
        for occ, so, wo, nd in ((d1[key], so1, wo1, n1), (d2[key], so2, wo2, n2)):
            for i in occ:
                nd.extend(range(i, i + k))
                wo.append(len(nd))
            so.append(len(wo) - 1)
        c1.append(len(d1[key])); c2.append(len(d2[key])); fl.append(k)
    return capi.MatchSets(set_off1=so1, walk_off1=wo1, nodes1=n1, set_off2=so2, walk_off2=wo2, nodes2=n2, count1=c1, count2=c2,
                          full_length=fl)


# BASELINE configs[2] / [3]: 10 sequences x 1 Mbp (seed 7, the generator above with its defaults), guide tree of SURVEY.md §8(d)
C3_NEWICK = "((((s0,s1),(s2,s3)),s4),(((s5,s6),(s7,s8)),s9));"
C3_TREE = (((("s0", "s1"), ("s2", "s3")), "s4"), ((("s5", "s6"), ("s7", "s8")), "s9"))


def c3_workload(length=1000000, n_seq=10, seed=7):
    """(names, {name: sequence}, guide tree as nested 2-tuples) of BASELINE configs[2]; other n_seq get a balanced tree"""
    seqs = hor_sequences(seed, length, n_seq)
    names = ["s%d" % i for i in range(n_seq)]
    if n_seq == 10:
        tree = C3_TREE
    else:
        def bal(nm):
            return nm[0] if len(nm) == 1 else (bal(nm[:len(nm) // 2]), bal(nm[len(nm) // 2:]))
        tree = bal(names)
    return names, dict(zip(names, seqs)), tree
