"""fuse (include/centrolign/fuse.hpp:46-152) and the loop body of Core::do_execution (core.hpp:268-392).

CPU suite: cl_fuse (host code of the C ABI) against the golden outputs of the compiled reference (tests/golden/fuse.npz), against
the graphs the reference's own MSA run carried from merge to merge (the parents of the root merge ARE the fused results of
the two leaf merges), and live against oracle/_ref.  GPU suite: cl_merge — find_matches, Core::align, fuse — replays that
4-sequence progressive MSA from its leaf graphs."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

Z = np.load(os.path.join(H.GOLDEN, "fuse.npz"))
CASES = {name: (g1, g2, pairs) for name, g1, g2, pairs in H.fuse_cases()}


def _golden(name, like):
    return capi.BaseGraph(*[Z["%s.%s" % (name, k)] for k in capi.GRAPH_KEYS], like.src_id, like.snk_id)


def _same_up_to_sentinel_labels(a, b):
    """the reference relabels the sentinels before every merge (reassign_sentinels, core.hpp:283-284)"""
    la, lb = a.label.copy(), b.label.copy()
    for g, l in ((a, la), (b, lb)):
        l[g.src_id] = l[g.snk_id] = 0
    return (a.src_id, a.snk_id) == (b.src_id, b.snk_id) and np.array_equal(la, lb) and \
        all(np.array_equal(getattr(a, k), getattr(b, k)) for k in capi.GRAPH_KEYS if k != "label")


@pytest.mark.parametrize("name", list(CASES))
def test_fuse_matches_reference_golden(name):
    g1, g2, pairs = CASES[name]
    got = capi.fuse(g1, g2, pairs)
    assert capi.graphs_equal(got, _golden(name, g1))
    # structure: every edge is mirrored, paths are walks of the fused graph
    n = len(got.label)
    fwd = sorted((v, int(w)) for v in range(n) for w in got.next_idx[int(got.next_off[v]):int(got.next_off[v + 1])])
    bwd = sorted((int(u), v) for v in range(n) for u in got.prev_idx[int(got.prev_off[v]):int(got.prev_off[v + 1])])
    assert fwd == bwd
    edges = set(fwd)
    for p in range(len(got.path_off) - 1):
        walk = got.path_nodes[int(got.path_off[p]):int(got.path_off[p + 1])]
        assert all((int(a), int(b)) in edges for a, b in zip(walk[:-1], walk[1:]))
    assert len(got.path_off) - 1 == (len(g1.path_off) - 1) + (len(g2.path_off) - 1)


@pytest.mark.parametrize("m", [0, 1, 2])
def test_fuse_msa_merges_match_reference_golden(m):
    z, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
    got = capi.fuse(graphs[0], graphs[1], z["stitched"].reshape(-1, 2))
    assert [len(got.label), len(got.next_idx), len(got.path_off) - 1] == Z["merge%d.sizes" % m].tolist()
    assert H.graph_digest(got) == str(Z["merge%d.digest" % m][0])


def test_fused_leaf_merges_are_the_root_merges_parents():
    """the reference's own run: the subproblem graphs of the root merge (dumped before its match finding) are the fused
    results of the two leaf merges"""
    _, root, _ = load_stitch_case("stitch4_30k_merge2.npz")
    by_size = {len(g.label): g for g in root}
    for m in (0, 1):
        z, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        got = capi.fuse(graphs[0], graphs[1], z["stitched"].reshape(-1, 2))
        assert _same_up_to_sentinel_labels(got, by_size[len(got.label)])


def test_fuse_rejects_malformed_input():
    g1, g2, pairs = CASES["fuse03"]
    bad = pairs.copy()
    bad[0, 0] = len(g1.label) + 5
    with pytest.raises(capi.ClError):
        capi.fuse(g1, g2, bad)
    empty = capi.fuse(g1, g2, np.zeros((0, 2), np.uint64))     # nothing aligned: the graphs share only the sentinels
    assert len(empty.label) == len(g1.label) + len(g2.label) - 2


@pytest.mark.skipif(not po.have_ref(), reason="needs oracle/_ref (build container only)")
def test_fuse_live_reference_agreement():
    for name, (g1, g2, pairs) in CASES.items():
        rng = np.random.default_rng(len(pairs))
        keep = rng.random(len(pairs)) < 0.7                      # a different alignment than the committed one
        sub = pairs[keep]
        assert capi.graphs_equal(capi.fuse(g1, g2, sub), po.ref_fuse(g1, g2, sub)), name
        # not an alignment at all: arbitrary pairs, nodes aligned several times — fuse is defined for any pair list
        n1, n2, k = len(g1.label) - 2, len(g2.label) - 2, 2 * len(pairs)
        a, b = rng.integers(0, n1, k).astype(np.uint64), rng.integers(0, n2, k).astype(np.uint64)
        a[rng.random(k) < 0.15] = H.GAP
        b[rng.random(k) < 0.15] = H.GAP
        wild = np.stack([a, b], 1)
        assert capi.graphs_equal(capi.fuse(g1, g2, wild), po.ref_fuse(g1, g2, wild)), name


@pytest.mark.gpu
def test_gpu_merge_replays_the_reference_msa(gpu_ctx):
    """the 4 x 30 kbp progressive MSA from its four leaf graphs, three cl_merge calls: every alignment and every fused graph
    equals the reference's run (goldens stitch4_30k_merge*.npz, fuse.npz)"""
    scale = float(np.load(os.path.join(H.GOLDEN, "align4_30k_merge2.npz"))["score_scale"][0])
    fused = {}
    for m in (0, 1):
        z, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        r = gpu_ctx.merge(graphs[0], graphs[1], score_scale=scale, max_num_match_pairs=40000)
        assert np.array_equal(r["alignment"].reshape(-1), z["stitched"]), m
        assert H.graph_digest(r["fused"]) == str(Z["merge%d.digest" % m][0])
        fused[len(r["fused"].label)] = r["fused"]
    z, root, _ = load_stitch_case("stitch4_30k_merge2.npz")
    r = gpu_ctx.merge(fused[len(root[0].label)], fused[len(root[1].label)], score_scale=scale, max_num_match_pairs=40000)
    assert np.array_equal(r["alignment"].reshape(-1), z["stitched"])
    assert H.graph_digest(r["fused"]) == str(Z["merge2.digest"][0])
    assert len(r["fused"].path_off) - 1 == 4 and r["n_match_sets"] == 15492
