"""shared helpers for the parity tests"""
import hashlib
import os

import numpy as np

from centrolign_amd import capi, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIDE_KEYS = ("node_off", "label", "prev_off", "prev_idx", "next_off", "next_idx", "src_off", "src_idx", "snk_off",
             "snk_idx", "back_translation")


def load_batch(z, prefix=""):
    sides = [capi.GraphSide(**{k: z["%sg%d.%s" % (prefix, si, k)] for k in SIDE_KEYS}) for si in (1, 2)]
    return capi.StitchBatch(sides[0], sides[1], z[prefix + "only_deletion_alns"])


def load_result(z, prefix):
    n = len(z[prefix + "aln_off"]) - 1
    get = lambda k, dt: z[prefix + k] if (prefix + k) in z.files else np.zeros(n, dt)
    return capi.StitchResult(z[prefix + "aln_off"], z[prefix + "pairs"].reshape(-1, 2), get("score", np.int64),
                             get("route", np.uint8), get("num_pw", np.uint8))


def result_digest(aln_off, pairs):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(aln_off, dtype=np.uint64).tobytes())
    h.update(np.ascontiguousarray(pairs, dtype=np.uint64).tobytes())
    return h.hexdigest()


def tie_params():
    tp = capi.default_stitch_params()
    tp.alignment_params.match = 1
    tp.alignment_params.mismatch = 1
    tp.alignment_params.gap_open[:] = [1, 2, 3]
    tp.alignment_params.gap_extend[:] = [3, 2, 1]
    return tp


_c2 = {}


def c2_batch():
    """the reference's stitch batch for the 2 x 1 Mbp HOR pair, rebuilt from the seed + dumped intervals"""
    if "b" not in _c2:
        z = np.load(os.path.join(GOLDEN, "c2_pair_seed7_intervals.npz"))
        seqs = synth.hor_sequences(7, 1000000, 2)
        _c2["b"] = synth.batch_from_intervals(seqs[0], seqs[1], z["intervals"], z["only_del"])
        _c2["z"] = z
    return _c2["b"], _c2["z"]


def check_alignment_valid(batch, res, k):
    """structural validity of one alignment in LOCAL ids (back_translation removed by the caller):
    every node appears at most once per side, ids increase along graph edges is not checked here"""
    aln = res.alignment(k)
    for col in (0, 1):
        ids = aln[:, col]
        ids = ids[ids != capi.CL_GAP]
        assert len(np.unique(ids)) == len(ids)


# the reference's own fixed expectations, transcribed as data -------------------------------------------------
def _graph(labels, edges, sources, sinks):
    return dict(labels=[ord(c) for c in labels], edges=edges, sources=sources, sinks=sinks)


def known_answer_cases():
    """(graph1, graph2, num_pw, params, expected pairs) from src/test/test_alignment.cpp:684-773.
    -1 marks a gap."""
    bubbles = [(0, 1), (0, 2), (1, 3), (2, 3), (3, 4), (3, 5), (4, 6), (5, 6)]
    unit = capi.make_align_params(1, 1, [1], [1])
    g1 = _graph("ACGTGCA", bubbles, [0], [6])
    g2 = _graph("AGTTTGA", bubbles, [0], [6])
    cases = [(g1, g2, 1, unit, [(0, 0), (2, 1), (3, 3), (4, 5), (6, 6)])]
    g1t = _graph("ACGTGCAT", bubbles + [(7, 0)], [7], [6])
    g2t = _graph("AGTTTGAT", bubbles + [(6, 7)], [0], [7])
    cases.append((g1t, g2t, 1, unit, [(7, -1), (0, 0), (2, 1), (3, 3), (4, 5), (6, 6), (-1, 7)]))
    cases.append((g2t, g1t, 1, unit, [(-1, 7), (0, 0), (1, 2), (3, 3), (5, 4), (6, 6), (7, -1)]))
    return cases


def stitcher_known_answer():
    """src/test/test_stitcher.cpp:321-438: the four between-anchor subproblems Stitcher::stitch extracts for that
    fixture (derived by hand from extract_connecting_graph, include/centrolign/subgraph_extraction.hpp:52-125),
    the anchors copied between them, and the expected stitched alignment.  Stitcher class defaults
    (src/stitcher.cpp:13-22, stitcher.hpp:48-64)."""
    s1, s2 = "ACCAGTCGTTGA", "GATCGTGAACTATGC"
    L = lambda s, ids: [ord(s[i]) for i in ids]
    subs = [
        # (graph1 nodes, edges, sources, sinks, back), (graph2 ...), only_deletion_alns
        (dict(labels=[], edges=[], sources=[], sinks=[], back=[]),
         dict(labels=L(s2, [0]), edges=[], sources=[0], sinks=[0], back=[0]), 1),
        (dict(labels=L(s1, [3]), edges=[], sources=[0], sinks=[0], back=[3]),
         dict(labels=[], edges=[], sources=[], sinks=[], back=[]), 0),
        (dict(labels=L(s1, [6, 7, 8]), edges=[(1, 2), (0, 2)], sources=[0, 1], sinks=[2], back=[6, 7, 8]),
         dict(labels=L(s2, [7, 8, 9, 10]), edges=[(0, 1), (0, 2), (2, 3), (1, 3)], sources=[0], sinks=[3],
              back=[7, 8, 9, 10]), 0),
        (dict(labels=L(s1, [11]), edges=[], sources=[0], sinks=[0], back=[11]),
         dict(labels=L(s2, [14]), edges=[], sources=[0], sinks=[0], back=[14]), 1),
    ]
    anchors = [[(0, 1), (1, 3)], [(4, 4), (5, 5)], [(9, 12), (10, 13)]]
    expected = [(-1, 0), (0, 1), (1, 3), (3, -1), (4, 4), (5, 5), (-1, 7), (6, 9), (8, 10), (9, 12), (10, 13), (11, 14)]
    sp = capi.default_stitch_params()
    sp.max_trivial_size, sp.min_wfa_size, sp.max_wfa_size = 30000, 10000000, 50000000
    sp.deletion_alignment_ratio, sp.deletion_alignment_short_max_size, sp.deletion_alignment_long_min_size = 4, 4000, 2000
    return subs, anchors, expected, sp


def batch_from_graphs(pairs_of_graphs, only_del=None):
    b1, b2 = synth._SideBuilder(), synth._SideBuilder()
    for g1, g2 in pairs_of_graphs:
        for bld, g in ((b1, g1), (b2, g2)):
            bld.add_graph(np.array(g["labels"], np.uint8), g["edges"], g["sources"], g["sinks"], g.get("back"))
    n = len(pairs_of_graphs)
    return capi.StitchBatch(b1.finish(), b2.finish(), np.zeros(n, np.uint8) if only_del is None else only_del)


def as_signed_pairs(pairs):
    return [(-1 if a == capi.CL_GAP else int(a), -1 if b == capi.CL_GAP else int(b)) for a, b in pairs]


class host_route_batches:
    """the batches of tests/golden/host_routes.npz: every host route of Stitcher::do_alignment (stitcher.hpp:268-360)"""

    @staticmethod
    def make_sequences():
        rng = np.random.default_rng(9)
        a = rng.integers(0, 4, 12000).astype(np.uint8)
        b = a.copy()
        idx = rng.choice(len(a), 150, replace=False)
        b[idx] = (b[idx] + rng.integers(1, 4, 150)) % 4
        return a, b

    @staticmethod
    def build(a, b):
        from centrolign_amd import capi, synth
        out = {}
        # "ad1/ad2": lopsided unalignable gaps (short side <= 1500, long side >= 2000 and >= 8x), CLI thresholds
        rows = np.array([(100, 250, 100, 2100), (3000, 2400, 3000, 200), (500, 100, 400, 2500), (6000, 2200, 6100, 150)], np.int64)
        out["ad_linear"] = (synth.batch_from_intervals(a, b, rows, np.ones(len(rows), np.uint8)), capi.default_stitch_params())
        # "w": near-diagonal gaps above min_wfa_size (shrunk so that small problems take the route)
        sp = capi.default_stitch_params()
        sp.min_wfa_size, sp.max_wfa_size = 1000, 10 ** 9
        rows = np.array([(100, 300, 100, 300), (1000, 500, 1000, 510), (2000, 800, 2000, 790), (4000, 200, 4000, 205),
                         (7000, 1500, 7000, 1500), (9000, 400, 3000, 400)], np.int64)
        out["w_linear"] = (synth.batch_from_intervals(a, b, rows, np.zeros(len(rows), np.uint8)), sp)
        sp = capi.default_stitch_params()
        sp.min_wfa_size, sp.max_wfa_size, sp.max_wfa_ratio = 50, 10 ** 9, 100.0
        db = synth.random_dag_batch(60, seed=21, max_n=60)
        db.only_deletion_alns[:] = 0
        out["w_dags"] = (db, sp)
        # a mixed batch: random DAG pairs flagged unalignable, thresholds shrunk so that every route occurs
        sp = capi.default_stitch_params()
        sp.max_trivial_size, sp.deletion_alignment_ratio = 10, 3
        sp.deletion_alignment_short_max_size, sp.deletion_alignment_long_min_size = 100, 30
        db = synth.random_dag_batch(300, seed=23, max_n=150)
        db.only_deletion_alns[:] = 1
        out["mixed_dags"] = (db, sp)
        return out


def match_cases():
    """seeded inputs of the match-finding parity tests: (name, graph1, graph2, max_count).  Tandem-repeat ancestors at
    several divergences; leaf pairs (one chain, one path each) and merged-subproblem-like graphs (several paths over a DAG
    with bubbles); the second graph usually carries its own sentinel characters (7, 8) as in a real merge."""
    cases = []
    for seed in range(24):
        rng = np.random.default_rng(1000 + seed)
        L = int(rng.integers(20, 3000))
        unit = rng.integers(0, 4, int(rng.integers(3, 60))).astype(np.uint8)
        anc = np.tile(unit, L // len(unit) + 1)[:L].copy()
        mut = rng.random(L) < rng.choice([0.0, 0.01, 0.05, 0.3])
        anc[mut] = rng.integers(0, 4, int(mut.sum()))
        if seed % 3 == 0:
            a2 = anc.copy()
            m2 = rng.random(L) < 0.02
            a2[m2] = rng.integers(0, 4, int(m2.sum()))
            g1 = synth.base_graph_from_sequence(anc)
            g2 = synth.base_graph_from_sequence(a2[: L - int(rng.integers(0, 10))], sentinels=(7, 8))
        else:
            g1 = synth.bubble_graph(anc, int(rng.integers(1, 5)), seed=seed)
            g2 = synth.bubble_graph(anc, int(rng.integers(1, 5)), seed=seed + 1000, sentinels=(7, 8) if seed % 2 else (5, 6))
        cases.append(("rand%02d" % seed, g1, g2, int((1, 4, 50, 3000)[seed % 4])))
    # degenerate: a one-base pair, identical sequences, a homopolymer against itself, nothing in common
    one = np.array([2], np.uint8)
    same = np.random.default_rng(7).integers(0, 4, 500).astype(np.uint8)
    cases.append(("one_base", synth.base_graph_from_sequence(one), synth.base_graph_from_sequence(one, sentinels=(7, 8)), 3000))
    cases.append(("identical", synth.base_graph_from_sequence(same), synth.base_graph_from_sequence(same, sentinels=(7, 8)), 3000))
    cases.append(("homopolymer", synth.base_graph_from_sequence(np.zeros(300, np.uint8)),
                  synth.base_graph_from_sequence(np.zeros(280, np.uint8), sentinels=(7, 8)), 3000))
    cases.append(("disjoint", synth.base_graph_from_sequence(np.zeros(50, np.uint8)),
                  synth.base_graph_from_sequence(np.ones(50, np.uint8), sentinels=(7, 8)), 3000))
    return cases


def match_sets_digest(ms):
    """sha256 over every array of a MatchSets (for fixtures too large to commit in full)"""
    h = hashlib.sha256()
    for k in capi.MatchSets._DT:
        h.update(np.ascontiguousarray(getattr(ms, k)).tobytes())
    return h.hexdigest()


GAP = np.uint64(2 ** 64 - 1)


def fuse_cases():
    """seeded inputs of the fuse parity tests: (name, dest graph, source graph, alignment): bubble graphs over related
    ancestors with an alignment that walks their first paths in lockstep with random gaps (matches, mismatches ->
    substitution edges, insertions, deletions; leading / trailing gaps)"""
    cases = []
    for seed in range(24):
        rng = np.random.default_rng(300 + seed)
        L = int(rng.integers(1, 400))
        anc = rng.integers(0, 4, L).astype(np.uint8)
        a2 = anc.copy()
        mm = rng.random(L) < 0.1
        a2[mm] = rng.integers(0, 4, int(mm.sum()))
        g1 = synth.bubble_graph(anc, int(rng.integers(1, 4)), seed=seed)
        g2 = synth.bubble_graph(a2, int(rng.integers(1, 4)), seed=seed + 500, sentinels=(7, 8))
        p1 = g1.path_nodes[int(g1.path_off[0]):int(g1.path_off[1])]
        p2 = g2.path_nodes[int(g2.path_off[0]):int(g2.path_off[1])]
        i = j = 0
        out = []
        while i < len(p1) or j < len(p2):
            r = rng.random()
            if i < len(p1) and j < len(p2) and r < 0.8:
                out.append((p1[i], p2[j])); i += 1; j += 1
            elif i < len(p1) and (r < 0.9 or j >= len(p2)):
                out.append((p1[i], GAP)); i += 1
            else:
                out.append((GAP, p2[j])); j += 1
        cases.append(("fuse%02d" % seed, g1, g2, np.array(out, np.uint64).reshape(-1, 2)))
    return cases


def graph_digest(g):
    h = hashlib.sha256()
    for k in capi.GRAPH_KEYS:
        h.update(np.ascontiguousarray(getattr(g, k)).tobytes())
    h.update(np.array([g.src_id, g.snk_id], np.uint64).tobytes())
    return h.hexdigest()


def calibration_leaves():
    """seeded leaf graphs of the calibration parity tests: (name, graph, max_num_match_pairs)"""
    out = []
    for k, (length, budget, kw) in enumerate([(8000, 1250000, {}), (20000, 5000, {}), (12000, 1250000, dict(hor_div=0.1)),
                                             (3000, 1250000, dict(mono_len=31, hor_n=4)), (40000, 30000, dict(seq_div=0.02))]):
        seq = synth.hor_sequences(40 + k, length, 1, **kw)[0]
        out.append(("leaf%d" % k, synth.base_graph_from_sequence(seq), budget))
    rng = np.random.default_rng(9)
    out.append(("random", synth.base_graph_from_sequence(rng.integers(0, 4, 5000).astype(np.uint8)), 1250000))
    return out


def msa_cases():
    """(name, n sequences, length, seed, max_num_match_pairs) of the end-to-end fixtures (balanced guide tree over seq0..)"""
    return [("pair_10k", 2, 10000, 21, 1250000), ("pair_60k", 2, 60000, 3, 100000), ("msa4_8k", 4, 8000, 13, 20000), ("msa3_6k", 3, 6000, 5, 1250000),
            ("msa5_5k", 5, 5000, 8, 8000)]
