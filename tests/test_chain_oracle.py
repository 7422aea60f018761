"""CPU suite: pins the chaining oracle (oracle/chain_oracle.cpp: PathMerge, PostSwitchDistances, ForwardEdges,
MatchBank, MaxSearchTree, OrthogonalMaxSearchTree, sparse_chain_dp, sparse_affine_chain_dp, traceback) to the
reference: identical chains — the same (match set, idx1, idx2) triples in the same order — on the merges of a
4-sequence MSA (1-path x 1-path leaf merges and the 2-path x 2-path root merge)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

FILES = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("chain4_"))


@pytest.mark.parametrize("name", FILES)
def test_chain_oracle_matches_golden(name):
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    for tag in ("a", "b"):
        ms = po.MatchSets(**{k: z["%s.ms.%s" % (tag, k)] for k in po.MatchSets._DT})
        got = po.oracle_chain("sparse", graphs[0], graphs[1], ms)
        assert np.array_equal(got, z[tag + ".chain_sparse"])
        got = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=float(z[tag + ".scale"][0]))
        assert np.array_equal(got, z[tag + ".chain_affine"])
        assert len(got) > 50
        # Anchorer::global_anchoring (the CLI default): lead / final indels to the graph's ends are part of the score
        got = po.oracle_chain("sparse", graphs[0], graphs[1], ms, global_anchoring=True)
        assert np.array_equal(got, z[tag + ".chain_sparse_global"])
        got = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=float(z[tag + ".scale"][0]), global_anchoring=True)
        assert np.array_equal(got, z[tag + ".chain_affine_global"])


def test_chain_is_collinear():
    """structural property: consecutive anchors of a chain advance in both graphs"""
    z = np.load(os.path.join(H.GOLDEN, FILES[0]))
    _, graphs, _ = load_stitch_case(FILES[0].replace("chain4_", "stitch4_"))
    ms = po.MatchSets(**{k: z["a.ms." + k] for k in po.MatchSets._DT})
    chain, dp = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=0.7, want_dp=True)
    so1, wo1, so2, wo2 = (x.astype(np.int64) for x in (ms.set_off1, ms.walk_off1, ms.set_off2, ms.walk_off2))
    ends1 = [ms.nodes1[wo1[so1[s] + i + 1] - 1] for s, i, _ in chain]
    starts1 = [ms.nodes1[wo1[so1[s] + i]] for s, i, _ in chain]
    ends2 = [ms.nodes2[wo2[so2[s] + j + 1] - 1] for s, _, j in chain]
    starts2 = [ms.nodes2[wo2[so2[s] + j]] for s, _, j in chain]
    # leaf graphs: node id == sequence position
    assert all(e < s for e, s in zip(ends1[:-1], starts1[1:]))
    assert all(e < s for e, s in zip(ends2[:-1], starts2[1:]))
    assert np.isfinite(dp).all() and dp.max() > 0


@pytest.mark.ref
@pytest.mark.skipif(not po.have_ref(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("name", FILES)
def test_chain_oracle_vs_compiled_reference_live(name):
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    full = po.MatchSets(**{k: z["a.ms." + k] for k in po.MatchSets._DT})
    for seed, budget, scale in ((1, 3000, 1.0), (2, 8000, 0.1)):
        ms = po.budget_subset(full, budget, seed=seed)
        for algo in ("sparse", "affine"):
            for glob in (False, True):
                ref, _ = po.ref_chain(algo, graphs[0], graphs[1], ms, scale=scale, global_anchoring=glob)
                assert np.array_equal(po.oracle_chain(algo, graphs[0], graphs[1], ms, scale=scale, global_anchoring=glob), ref)


def chain_weight(ms, chain):
    tot = 0.0
    for s in chain[:, 0]:
        c = float(ms.count1[s] * ms.count2[s])
        w0 = int(ms.set_off1[s])
        ln, fl = float(ms.walk_off1[w0 + 1] - ms.walk_off1[w0]), float(ms.full_length[s])
        tot += (ln / fl) * (ln / c ** 0.5 - (ln / 2250.0) ** 2 * 2250.0)
    return tot


@pytest.mark.parametrize("name", FILES)
def test_sparse_chain_is_optimal(name):
    """independent check: the gap-free chain reaches the total weight that the reference's exhaustive_chain_dp
    (anchorer.hpp:1342-1509, O(M^2), the "-g 0" algorithm) finds on the same matches"""
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    full = po.MatchSets(**{k: z["a.ms." + k] for k in po.MatchSets._DT})
    for seed, budget in ((1, 1500), (2, 3000)):
        ms = po.budget_subset(full, budget, seed=seed)
        chain = po.oracle_chain("sparse", graphs[0], graphs[1], ms, global_anchoring=True)
        assert abs(chain_weight(ms, chain) - float(z["exhaustive.%d.%d" % (seed, budget)][0])) < 1e-6
