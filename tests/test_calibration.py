"""Calibration (Core::calibrate_anchor_scores_and_identify_bonds without cyclisation, src/core.cpp:98-191): the per-leaf
intrinsic score scale — self matches, main-diagonal subset, Anchorer::estimate_score_scale — against the compiled reference's
values (tests/golden/calibration.npz; the mean over the four leaves of the 4 x 30 kbp MSA is the score_scale the reference's
own run of that MSA used).  Bit-exact doubles."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from tests import helpers as H
from tests.test_extraction import load_stitch_case

Z = np.load(os.path.join(H.GOLDEN, "calibration.npz"))


def test_fixture_mean_is_the_reference_runs_score_scale():
    run = np.load(os.path.join(H.GOLDEN, "align4_30k_merge2.npz"))
    assert float(Z["msa4_30k.mean"][0]) == float(run["score_scale"][0])
    s = Z["msa4_30k.scales"]
    assert sum(s.tolist()) / len(s) == float(Z["msa4_30k.mean"][0])


@pytest.mark.gpu
def test_gpu_leaf_scales_of_the_msa_match_reference(gpu_ctx):
    got = []
    for m in (0, 1):
        _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        got += [gpu_ctx.leaf_intrinsic_scale(g, max_num_match_pairs=40000) for g in graphs]
    assert got == Z["msa4_30k.scales"].tolist()
    assert sum(got) / len(got) == float(Z["msa4_30k.mean"][0])   # ScoreFunction::score_scale (src/core.cpp:169-184)


@pytest.mark.gpu
@pytest.mark.parametrize("case", H.calibration_leaves(), ids=lambda c: c[0])
def test_gpu_leaf_scale_matches_reference(gpu_ctx, case):
    name, g, budget = case
    assert gpu_ctx.leaf_intrinsic_scale(g, max_num_match_pairs=budget) == float(Z[name][0])


@pytest.mark.gpu
def test_gpu_estimate_score_scale_is_what_anchor_chain_reports(gpu_ctx):
    """cl_estimate_score_scale alone == the scale cl_anchor_chain estimates on the way (same matches, same parameters)"""
    import ctypes as C
    z = np.load(os.path.join(H.GOLDEN, "anchor4_30k_merge2.npz"))
    _, graphs, _ = load_stitch_case("stitch4_30k_merge2.npz")
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    full = gpu_ctx.anchor_chain(graphs[0], graphs[1], ms, score_scale=float(z["score_scale"][0]), max_num_match_pairs=int(z["max_num_match_pairs"][0]))
    ap = capi.AnchorParams()
    ap.chain = capi.default_chain_params()
    ap.max_num_match_pairs = int(z["max_num_match_pairs"][0])
    ap.score_scale = float(z["score_scale"][0])
    ap.autocalibrate_gap_penalties = 1
    ap.do_fill_in_anchoring = 1
    g1, g2, mc, scale = graphs[0].as_c(), graphs[1].as_c(), ms.as_c(), C.c_double(0)
    assert gpu_ctx.lib.cl_estimate_score_scale(gpu_ctx.handle, C.byref(g1), C.byref(g2), C.byref(mc), C.byref(ap), C.byref(scale)) == 0
    assert scale.value == full["scale"] == float(z["f.scale"])   # "f": global anchoring with fill-in, the default configuration
