"""Cyclisation's graph bookkeeping against the compiled reference on random cyclic graphs (host only, live: needs oracle/_ref): simplify_bubbles
(which the product recognises without a snarl decomposition, cl_cyclize_api.cpp) and InconsistencyIdentifier::identify_inconsistencies (snarl tree
of a cyclic graph through a cactus graph whose three-edge-connected components are found by hashing here, by Tsin's algorithm in the reference,
snarl_tree.hpp).  Graphs: bubble graphs over a tandem-repeat ancestor, made cyclic by fusing a stretch of every path with its own copy one
repeat unit further on (cl_internal_fuse, pinned elsewhere), with some nodes cloned so that bubbles with identical alleles exist."""
import numpy as np
import pytest

from centrolign_amd import capi, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (the compiled reference) is not on this machine")


def lists_of(g):
    n = len(g.label)
    nxt = [list(g.next_idx[int(g.next_off[v]):int(g.next_off[v + 1])]) for v in range(n)]
    paths = [list(g.path_nodes[int(g.path_off[p]):int(g.path_off[p + 1])]) for p in range(len(g.path_off) - 1)]
    return list(g.label), nxt, paths


def graph_from(labels, paths, src, snk):
    """BaseGraph whose edges are the path adjacencies plus source -> first and last -> sink, in first-seen order"""
    n = len(labels)
    nxt, prv = [[] for _ in range(n)], [[] for _ in range(n)]

    def edge(a, b):
        if b not in nxt[a]:
            nxt[a].append(b); prv[b].append(a)
    for p in paths:
        edge(src, p[0])
        for a, b in zip(p, p[1:]):
            edge(a, b)
        edge(p[-1], snk)
    off = lambda ls: np.cumsum([0] + [len(x) for x in ls]).astype(np.uint64)
    flat = lambda ls: np.array([v for x in ls for v in x], np.uint32)
    return capi.BaseGraph(np.array(labels, np.uint8), off(nxt), flat(nxt), off(prv), flat(prv), off(paths), flat(paths), src, snk)


def random_cyclic_graph(seed):
    rng = np.random.default_rng(seed)
    unit = rng.integers(0, 4, size=int(rng.integers(12, 40)))
    copies = int(rng.integers(4, 9))
    anc = np.concatenate([np.where(rng.random(len(unit)) < 0.08, rng.integers(0, 4, size=len(unit)), unit) for _ in range(copies)]).astype(np.uint8)
    g = synth.bubble_graph(anc, int(rng.integers(2, 5)), seed=seed, alt_p=0.06, skip_p=0.03)
    labels, nxt, paths = lists_of(g)
    src, snk = g.src_id, g.snk_id
    # clone some inner nodes for some of the paths through them: bubbles with identical alleles
    for _ in range(int(rng.integers(0, 8))):
        p = int(rng.integers(0, len(paths)))
        i = int(rng.integers(1, len(paths[p]) - 1))
        labels.append(labels[paths[p][i]])
        paths[p][i] = len(labels) - 1
    g = graph_from(labels, paths, src, snk)
    # tandem bonds: a stretch of every path fused with the stretch one repeat unit later, base by base where the labels agree
    pairs = []
    for p in paths:
        if len(p) < 3 * len(unit):
            continue
        a = int(rng.integers(0, len(p) - 2 * len(unit)))
        length = int(rng.integers(len(unit) // 2, len(p) - a - len(unit)))
        for k in range(length):
            x, y = p[a + k], p[a + k + len(unit)]
            if labels[x] == labels[y] and rng.random() < 0.9:
                pairs.append((x, y))
    if not pairs:
        return g
    fused, _ = capi.internal_fuse(g, np.array(pairs, np.uint64))
    return fused


SETTINGS = [(10000, 100, 8, 50, 1000, 10000), (10000, 100, 2, 4, 30, 200), (40, 6, 1, 2, 10, 60)]


@pytest.mark.parametrize("block", range(4))
def test_simplify_bubbles_and_inconsistencies_match_the_reference_on_random_cyclic_graphs(block):
    n_simplified = n_regions = 0
    for seed in range(block * 40, block * 40 + 40):
        g = random_cyclic_graph(1000 + seed)
        want = po.ref_simplify_bubbles(g)
        got = capi.simplify_bubbles(g)
        assert capi.graphs_equal(got, want), seed
        n_simplified += len(got.label) < len(g.label)
        for st in SETTINGS:
            a = capi.identify_inconsistencies(got, capi.polish_params(**dict(zip(("max_tight_cycle_size", "max_bond_inconsistency_window",
                                              "min_inconsistency_disjoint_length", "min_inconsistency_total_length", "padding_target_min_length",
                                              "padding_max_length_limit"), st))))
            b = po.ref_inconsistencies(got, st)
            assert np.array_equal(a, b), (seed, st, a.tolist(), b.tolist())
            n_regions += len(a)
    assert n_regions > 0
