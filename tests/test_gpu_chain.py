"""GPU suite: the Anchorer's sparse affine chaining DP on the GPU (cl_chain_sparse_affine) against the chaining
oracle and the golden chains of the compiled reference: every DP value bit-identical (float), identical chain."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

pytestmark = pytest.mark.gpu

FILES = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("chain4_"))


@pytest.mark.parametrize("name", FILES)
def test_chain_matches_reference_golden(gpu_ctx, name):
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    for tag in ("a", "b"):
        ms = capi.MatchSets(**{k: z["%s.ms.%s" % (tag, k)] for k in capi.MatchSets._DT})
        scale = float(z[tag + ".scale"][0])
        got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=scale, want_dp=True)
        want_chain, want_dp = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=scale, want_dp=True)
        assert np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32)), "DP values differ"
        assert np.array_equal(got["chain"], z[tag + ".chain_affine"])
        assert np.array_equal(got["chain"], want_chain)


@pytest.mark.parametrize("seed,budget,scale", [(3, 1500, 1.0), (4, 12000, 0.05), (5, 30000, 0.6)])
def test_chain_vs_oracle_other_subsets(gpu_ctx, seed, budget, scale):
    name = FILES[seed % len(FILES)]
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    full = capi.MatchSets(**{k: z["a.ms." + k] for k in capi.MatchSets._DT})
    ms = po.budget_subset(full, budget, seed=seed)
    got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=scale, want_dp=True)
    want_chain, want_dp = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=scale, want_dp=True)
    assert np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32))
    assert np.array_equal(got["chain"], want_chain)


@pytest.mark.parametrize("name", FILES)
def test_sparse_chain_matches_reference_golden(gpu_ctx, name):
    """sparse_chain_dp (the gap-free chaining of estimate_score_scale / leaf calibration) on the GPU"""
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    for tag in ("a", "b"):
        ms = capi.MatchSets(**{k: z["%s.ms.%s" % (tag, k)] for k in capi.MatchSets._DT})
        got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, want_dp=True, sparse=True)
        want_chain, want_dp = po.oracle_chain("sparse", graphs[0], graphs[1], ms, want_dp=True)
        assert np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32)), "DP values differ"
        assert np.array_equal(got["chain"], z[tag + ".chain_sparse"])
        assert np.array_equal(got["chain"], want_chain)
