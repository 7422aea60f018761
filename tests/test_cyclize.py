"""The tandem-duplication rounds of cyclisation (src/core.cpp:196-296), the parts that lie on the hot path's seams:
Anchorer::anchor_chain with masked matches and an overriding scale (seam S3's last two arguments), Core::generate_diagonal_mask /
update_mask, and Stitcher::internal_stitch (seam S2's second entry).  Expected values: the compiled reference
(tests/golden/make_golden.py cyclize -> cyclize_rounds.npz)."""
import os

import numpy as np
import pytest

from centrolign_amd import capi, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    z = np.load(os.path.join(HERE, "golden", "cyclize_rounds.npz"))
    for name in z["names"]:
        d = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
        seq = bytes(d["seq"]).decode()
        ms = capi.MatchSets(**{f: d["ms." + f] for f in capi.MatchSets._DT})
        yield str(name), synth.base_graph_from_sequence(seq, (5, 6)), ms, d


def to_original(mask, set_order):
    """a mask in the reordered indexing (position k holds original set set_order[k]) -> the caller's indexing, sorted"""
    m = np.asarray(mask, np.uint64).reshape(-1, 3).copy()
    if len(m):
        m[:, 0] = np.asarray(set_order, np.uint64)[m[:, 0].astype(np.int64)]
        m = m[np.lexsort((m[:, 2], m[:, 1], m[:, 0]))]
    return m


def test_diagonal_mask_and_update_mask_match_the_reference():
    """host only: Core::generate_diagonal_mask and Core::update_mask (mask_reciprocal = true, as src/core.cpp:290 calls it)"""
    for name, g, ms, d in cases():
        assert np.array_equal(capi.generate_diagonal_mask(ms), d["mask0"]), name
        cur, mask = ms, d["mask0"]
        for rnd in (1, 2):
            pre = "r%d." % rnd
            cur = cur.reordered(d[pre + "set_order"])
            got = capi.update_mask(cur, d[pre + "walk1"], d[pre + "walk2"], d[pre + "mask_after_chain"], mask_reciprocal=True)
            assert np.array_equal(got, d[pre + "mask_after_update"]), (name, rnd)
        # without the mirror image fewer pairs are masked, and the result always contains what went in
        one_way = capi.update_mask(ms, d["r1.walk1"], d["r1.walk2"], d["mask0"], mask_reciprocal=False)
        both = capi.update_mask(ms, d["r1.walk1"], d["r1.walk2"], d["mask0"], mask_reciprocal=True)
        assert len(d["mask0"]) <= len(one_way) <= len(both), name
        assert set(map(tuple, d["mask0"])) <= set(map(tuple, one_way)) <= set(map(tuple, both)), name


def test_match_sets_reordered_is_a_permutation():
    for name, g, ms, d in cases():
        order = d["r1.set_order"]
        r = ms.reordered(order)
        assert r.n_sets == ms.n_sets and r.n_pairs() == ms.n_pairs()
        inv = np.argsort(order)
        back = r.reordered(inv)
        for f in capi.MatchSets._DT:
            assert np.array_equal(getattr(back, f), getattr(ms, f)), (name, f)


@pytest.mark.gpu
def test_masked_anchor_chain_rounds_match_the_reference(gpu_ctx):
    """two rounds as Core runs them: chain with the mask and the leaf's scale, reorder the sets as the reference does in place,
    update the mask, chain again"""
    for name, g, ms, d in cases():
        scale, budget = float(d["scale"][0]), int(d["budget"][0])
        cur, mask = ms, d["mask0"]
        for rnd in (1, 2):
            pre = "r%d." % rnd
            got = gpu_ctx.anchor_chain_masked(g, g, cur, mask, override_scale=scale, max_num_match_pairs=budget, score_scale=scale)
            assert np.array_equal(got["set_order"], d[pre + "set_order"]), (name, rnd)
            assert np.array_equal(got["chain"], d[pre + "chain"]), (name, rnd, len(got["chain"]), len(d[pre + "chain"]))
            for f in ("walk_off", "walk1", "walk2"):
                assert np.array_equal(got[f], d[pre + f]), (name, rnd, f)
            assert np.array_equal(got["score"], d[pre + "score"]), (name, rnd)
            assert got["scale"] == scale
            # the reference re-indexes the mask to the new order in place; here it stays in the caller's indexing
            assert np.array_equal(to_original(d[pre + "mask_after_chain"], got["set_order"]), np.asarray(mask, np.uint64).reshape(-1, 3)), (name, rnd)
            cur = cur.reordered(got["set_order"])
            mask = capi.update_mask(cur, got["walk1"], got["walk2"], d[pre + "mask_after_chain"], mask_reciprocal=True)
            assert np.array_equal(mask, d[pre + "mask_after_update"]), (name, rnd)


@pytest.mark.gpu
def test_mask_changes_the_chain_and_an_empty_mask_does_not(gpu_ctx):
    for name, g, ms, d in cases():
        scale, budget = float(d["scale"][0]), int(d["budget"][0])
        plain = gpu_ctx.anchor_chain(g, g, ms, max_num_match_pairs=budget, score_scale=scale)
        empty = gpu_ctx.anchor_chain_masked(g, g, ms, np.zeros((0, 3), np.uint64), max_num_match_pairs=budget, score_scale=scale)
        for f in ("chain", "walk1", "walk2", "score", "set_order"):
            assert np.array_equal(plain[f], empty[f]), (name, f)
        assert plain["scale"] == empty["scale"]
        # unmasked, a sequence against itself chains along the main diagonal; masked, no anchor may pair a node with itself
        assert len(plain["walk1"]) and np.array_equal(plain["walk1"], plain["walk2"]), name
        masked = gpu_ctx.anchor_chain_masked(g, g, ms, d["mask0"], override_scale=scale, max_num_match_pairs=budget, score_scale=scale)
        assert not np.any(masked["walk1"] == masked["walk2"]), name


@pytest.mark.gpu
def test_internal_stitch_matches_the_reference(gpu_ctx):
    for name, g, ms, d in cases():
        n = int(d["stitch.n_anchors"][0])
        wo = d["r1.walk_off"][:n + 1]
        w1, w2 = d["r1.walk1"][:int(wo[-1])], d["r1.walk2"][:int(wo[-1])]
        got = gpu_ctx.internal_stitch(g, wo, w1, w2)
        assert np.array_equal(got, d["stitch.pairs"]), (name, len(got), len(d["stitch.pairs"]))
        if n:
            # a single anchor has no gap: its pairs come back as they are
            one = gpu_ctx.internal_stitch(g, wo[:2], w1[:int(wo[1])], w2[:int(wo[1])])
            assert np.array_equal(one, np.stack([w1[:int(wo[1])], w2[:int(wo[1])]], 1).astype(np.uint64)), name
    assert len(gpu_ctx.internal_stitch(g, np.zeros(1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32))) == 0


def _graph(z, pre):
    ids = z[pre + "ids"]
    return capi.BaseGraph(*[z[pre + k] for k in capi.GRAPH_KEYS], int(ids[0]), int(ids[1]))


def test_internal_fuse_matches_the_reference():
    """host only: internal_fuse (fuse.hpp:144-247) — node numbering by union-find group and label, adjacency lists in first-met order,
    translated paths and sentinels, and the old -> new translation — on the tandem-duplication alignments of the leaves and on random
    pairs over multi-path graphs (transitive merges, mixed labels, cycles); live against the compiled reference where it is built"""
    z = np.load(os.path.join(HERE, "golden", "internal_fuse.npz"))
    for name in z["names"]:
        g, want = _graph(z, "%s.g." % name), _graph(z, "%s.f." % name)
        got, trans = capi.internal_fuse(g, z["%s.pairs" % name])
        assert capi.graphs_equal(got, want), name
        assert np.array_equal(trans, z["%s.trans" % name]), name
        assert (got.src_id, got.snk_id) == (int(trans[g.src_id]), int(trans[g.snk_id]))
        # merged nodes share their label; an alignment with nothing but gaps changes nothing but (possibly) the numbering
        assert np.array_equal(got.label[trans.astype(np.int64)], g.label)
    name = str(z["names"][0])
    g = _graph(z, "%s.g." % name)
    same, trans = capi.internal_fuse(g, np.zeros((0, 2), np.uint64))
    assert len(same.label) == len(g.label) and np.array_equal(trans, np.arange(len(g.label), dtype=np.uint64)) and capi.graphs_equal(same, g)
    try:
        from oracle import pyoracle as po
        po.ref_lib()
    except Exception:
        return
    rng = np.random.default_rng(77)
    for k in range(4):
        g = synth.bubble_graph("".join("ACGT"[b] for b in rng.integers(0, 4, 120)), 3, seed=50 + k)
        n = len(g.label)
        pairs = np.stack([rng.integers(0, n, 2 * n), rng.integers(0, n, 2 * n)], 1).astype(np.uint64)
        got, tg = capi.internal_fuse(g, pairs)
        want, tw = po.ref_internal_fuse(g, pairs)
        assert capi.graphs_equal(got, want) and np.array_equal(tg, tw), k
