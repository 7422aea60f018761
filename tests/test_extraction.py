"""CPU suite: the product's host-side extraction (PathMerge reachability + extract_connecting_graph +
extract_graphs_between, centrolign_amd/csrc/stitch_host.hpp) against the reference's own extraction, dumped by
oracle/ref_driver.cpp from a 4-sequence MSA: every array of every SubGraphInfo must be identical (node numbering,
previous()/next() order, sources, sinks, back_translation)."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from tests import helpers as H

FILES = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("stitch4_"))


def load_stitch_case(name):
    z = np.load(os.path.join(H.GOLDEN, name))
    graphs = []
    for side in ("parent1.", "parent2."):
        t = z[side + "tableau"]
        graphs.append(capi.BaseGraph(z[side + "label"], z[side + "next_off"], z[side + "next_idx"], z[side + "prev_off"],
                                     z[side + "prev_idx"], z[side + "path_off"], z[side + "path_nodes"], t[0], t[1]))
    segs = capi.AnchorSegments(z["seg_off"], z["walk_off"], z["walk1"], z["walk2"])
    return z, graphs, segs


@pytest.mark.parametrize("name", FILES)
def test_extraction_matches_reference(name):
    z, graphs, segs = load_stitch_case(name)
    got = capi.extract_stitch_batch(graphs[0], graphs[1], segs)
    want = H.load_batch(z)
    assert got.n_problems == want.n_problems
    assert np.array_equal(got.only_deletion_alns, want.only_deletion_alns)
    for si in range(2):
        for k in H.SIDE_KEYS:
            assert np.array_equal(getattr(got.side[si], k), getattr(want.side[si], k)), (si, k)


def test_extraction_rejects_cycles():
    g = capi.BaseGraph([1, 2, 3, 4], [0, 1, 2, 3, 3], [1, 2, 1], [0, 0, 2, 3, 4], [0, 2, 1, 2][:3] + [2], [0, 2], [1, 2], 0, 3)
    segs = capi.AnchorSegments([0], [0], [], [])
    with pytest.raises(capi.ClError) as e:
        capi.extract_stitch_batch(g, g, segs)
    assert e.value.code in (-7, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_stitch_matches_reference(gpu_ctx, name):
    """Stitcher::stitch through cl_stitch: extraction on the host, every subalign on the GPU, anchors interleaved"""
    z, graphs, segs = load_stitch_case(name)
    got = gpu_ctx.stitch(graphs[0], graphs[1], segs)
    assert np.array_equal(got, z["stitched"].reshape(-1, 2))


def test_concatenated_batches_hold_the_same_problems():
    """capi.concat_stitch_batches: the subproblems of several batches as one batch (host only) — every problem keeps its nodes, edges, sources,
    sinks, back-translation and only-deletion flag"""
    cases = [load_stitch_case("stitch4_30k_merge%d.npz" % m) for m in (0, 2)]
    batches = []
    for _, graphs, seg in cases:
        batches.append(capi.extract_stitch_batch(graphs[0], graphs[1], seg))
    one = capi.concat_stitch_batches(batches)
    assert one.n_problems == sum(b.n_problems for b in batches) and one.dp_cells() == sum(b.dp_cells() for b in batches)
    at = 0
    for b in batches:
        for k in list(range(0, b.n_problems, max(1, b.n_problems // 40))) + [b.n_problems - 1]:
            for si in (0, 1):
                assert one.side[si].problem(at + k) == b.side[si].problem(k)
                lo, hi = int(b.side[si].node_off[k]), int(b.side[si].node_off[k + 1])
                lo1 = int(one.side[si].node_off[at + k])
                assert np.array_equal(one.side[si].back_translation[lo1:lo1 + hi - lo], b.side[si].back_translation[lo:hi])
            assert one.only_deletion_alns[at + k] == b.only_deletion_alns[k]
        at += b.n_problems
