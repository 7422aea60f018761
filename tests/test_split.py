"""CPU suite: cl_split_branching_matches (host code of the C ABI) == Anchorer::split_branching_matches of the reference
(include/centrolign/anchorer.hpp:800-956): golden outputs of the compiled reference, and a live comparison where
oracle/_ref is present."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

FILES = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("split4_"))


def _inputs(name):
    z = np.load(os.path.join(H.GOLDEN, name.replace("split4_", "anchor4_")))
    _, graphs, _ = load_stitch_case(name.replace("split4_", "stitch4_"))
    return graphs, capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})


@pytest.mark.parametrize("name", FILES)
def test_split_matches_reference_golden(name):
    z = np.load(os.path.join(H.GOLDEN, name))
    graphs, ms = _inputs(name)
    grew = False
    for tag in ("a", "b"):
        got = capi.split_branching_matches(graphs[0], graphs[1], ms, *[int(x) for x in z[tag + ".params"]])
        for k in capi.MatchSets._DT:
            assert np.array_equal(getattr(got, k), z["%s.%s" % (tag, k)]), (tag, k)
        grew |= got.n_sets > ms.n_sets
    assert grew


def test_split_disabled_is_identity():
    graphs, ms = _inputs(FILES[0])
    got = capi.split_branching_matches(graphs[0], graphs[1], ms, anchor_split_limit=0)
    for k in capi.MatchSets._DT:
        assert np.array_equal(getattr(got, k), getattr(ms, k))


@pytest.mark.ref
@pytest.mark.skipif(not po.have_ref(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("prm", [(5, 128, 50, 16), (2, 20, 3, 8), (7, 15, 0, 1000)])
def test_split_vs_compiled_reference_live(prm):
    for name in FILES:
        graphs, ms = _inputs(name)
        want = po.ref_split_branching_matches(graphs[0], graphs[1], ms, *prm)
        got = capi.split_branching_matches(graphs[0], graphs[1], ms, *prm)
        for k in capi.MatchSets._DT:
            assert np.array_equal(getattr(got, k), getattr(want, k)), (name, k)
