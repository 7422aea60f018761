"""exhaustive_chain_dp + AnchorGraph::heaviest_weight_path (include/centrolign/anchorer.hpp:1342-1509, src/anchorer.cpp:68-133; SURVEY.md §8
row a23) behind cl_chain_exhaustive: the identical chain — the same anchors, not only the same weight — as the compiled reference on
budgeted subsets of the match sets of a 4-sequence MSA (tests/golden/exhaustive_chains.npz), and live where oracle/_ref is present."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

Z = np.load(os.path.join(H.GOLDEN, "exhaustive_chains.npz"))


@pytest.mark.parametrize("m", [0, 1, 2])
def test_exhaustive_chain_equals_reference(m):
    z = np.load(os.path.join(H.GOLDEN, "chain4_30k_merge%d.npz" % m))
    _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
    full = po.MatchSets(**{k: z["a.ms." + k] for k in po.MatchSets._DT})
    for seed, budget in ((1, 1500), (2, 3000), (3, 600)):
        sub = po.budget_subset(full, budget, seed=seed)
        for glob in (True, False):
            got = capi.chain_exhaustive(graphs[0], graphs[1], sub, params=capi.default_chain_params(global_anchoring=glob))
            assert np.array_equal(got, Z["m%d.%d.%d.%s" % (m, seed, budget, "g" if glob else "l")]), (m, seed, budget, glob)


@pytest.mark.ref
@pytest.mark.skipif(not po.have_ref(), reason="needs the compiled reference (build container only)")
def test_exhaustive_chain_live():
    z = np.load(os.path.join(H.GOLDEN, "chain4_30k_merge2.npz"))
    _, graphs, _ = load_stitch_case("stitch4_30k_merge2.npz")
    full = po.MatchSets(**{k: z["a.ms." + k] for k in po.MatchSets._DT})
    for seed, budget in ((11, 900), (12, 2000)):
        sub = po.budget_subset(full, budget, seed=seed)
        want, _ = po.ref_chain("exhaustive", graphs[0], graphs[1], sub, global_anchoring=True)
        assert np.array_equal(capi.chain_exhaustive(graphs[0], graphs[1], sub), want)
