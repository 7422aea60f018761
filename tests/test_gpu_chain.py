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


GLOBAL = pytest.mark.parametrize("glob", [False, True], ids=["local", "global"])  # Anchorer::global_anchoring


@GLOBAL
@pytest.mark.parametrize("name", FILES)
def test_chain_matches_reference_golden(gpu_ctx, name, glob):
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    for tag in ("a", "b"):
        ms = capi.MatchSets(**{k: z["%s.ms.%s" % (tag, k)] for k in capi.MatchSets._DT})
        scale = float(z[tag + ".scale"][0])
        got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=scale, want_dp=True,
                                          params=capi.default_chain_params(global_anchoring=glob))
        want_chain, want_dp = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=scale, want_dp=True, global_anchoring=glob)
        assert np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32)), "DP values differ"
        assert np.array_equal(got["chain"], z[tag + (".chain_affine_global" if glob else ".chain_affine")])
        assert np.array_equal(got["chain"], want_chain)


@GLOBAL
@pytest.mark.parametrize("seed,budget,scale", [(3, 1500, 1.0), (4, 12000, 0.05), (5, 30000, 0.6)])
def test_chain_vs_oracle_other_subsets(gpu_ctx, seed, budget, scale, glob):
    name = FILES[seed % len(FILES)]
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    full = capi.MatchSets(**{k: z["a.ms." + k] for k in capi.MatchSets._DT})
    ms = po.budget_subset(full, budget, seed=seed)
    got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=scale, want_dp=True,
                                      params=capi.default_chain_params(global_anchoring=glob))
    want_chain, want_dp = po.oracle_chain("affine", graphs[0], graphs[1], ms, scale=scale, want_dp=True, global_anchoring=glob)
    assert np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32))
    assert np.array_equal(got["chain"], want_chain)


@GLOBAL
@pytest.mark.parametrize("name", FILES)
def test_sparse_chain_matches_reference_golden(gpu_ctx, name, glob):
    """sparse_chain_dp (the gap-free chaining of estimate_score_scale / leaf calibration) on the GPU"""
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    for tag in ("a", "b"):
        ms = capi.MatchSets(**{k: z["%s.ms.%s" % (tag, k)] for k in capi.MatchSets._DT})
        got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, want_dp=True, sparse=True,
                                          params=capi.default_chain_params(global_anchoring=glob))
        want_chain, want_dp = po.oracle_chain("sparse", graphs[0], graphs[1], ms, want_dp=True, global_anchoring=glob)
        assert np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32)), "DP values differ"
        assert np.array_equal(got["chain"], z[tag + (".chain_sparse_global" if glob else ".chain_sparse")])
        assert np.array_equal(got["chain"], want_chain)


ANCHOR_FILES = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("anchor4_"))


@pytest.mark.parametrize("tag,glob,auto,fill", [("g", True, True, False), ("l", False, True, False), ("n", True, False, False),
                                                ("f", True, True, True), ("fl", False, True, True)])
@pytest.mark.parametrize("name", ANCHOR_FILES)
def test_anchor_chain_matches_reference_golden(gpu_ctx, name, tag, glob, auto, fill):
    """cl_anchor_chain == Anchorer::anchor_chain of the compiled reference (branch splitting off; with and without fill-in
    re-anchoring): the same reordering of the caller's match sets, the same estimated scale, the same chain with the same
    walks, identities, counts and gap / score annotation"""
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("anchor4_", "stitch4_"))
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    got = gpu_ctx.anchor_chain(graphs[0], graphs[1], ms, max_num_match_pairs=int(z["max_num_match_pairs"][0]),
                               score_scale=float(z["score_scale"][0]), autocalibrate=auto, fill_in=fill,
                               params=capi.default_chain_params(global_anchoring=glob))
    assert got["scale"] == float(z[tag + ".scale"])
    # anchor_t::score included: anchor_weight follows the operation order of the reference as built (-ffast-math)
    for k in ("set_order", "chain", "walk_off", "walk1", "walk2", "count1", "count2", "full_length", "gap_before", "gap_after",
              "gap_score_before", "gap_score_after", "score"):
        assert np.array_equal(got[k], z["%s.%s" % (tag, k)]), k
    if fill:
        assert got["fill_in_pairs"] > 0 and len(got["chain"]) > 2 * len(z["g.chain"])


@pytest.mark.parametrize("tag,glob,fill", [("s", True, True), ("sl", False, False)])
@pytest.mark.parametrize("m", [0, 1, 2])
def test_sparse_chaining_over_chain_merge_matches_the_reference(gpu_ctx, m, tag, glob, fill):
    """the CLI's hidden -g 1: Anchorer::chaining_algorithm = Sparse — no scale estimate, sparse_chain_dp for the chain and its fill-in — over
    ChainMerge structures (every node on ONE chain; include/centrolign/chain_merge.hpp:100-225), which is what Core::execute builds for that
    setting (core.hpp:350-357).  Merge 2 pairs graphs of 2 + 2 paths.  Expected: the compiled reference (tests/golden/make_chainmerge_golden.py)"""
    z = np.load(os.path.join(H.GOLDEN, "anchor4_30k_merge%d.npz" % m))
    zg = np.load(os.path.join(H.GOLDEN, "chainmerge4_30k_g1.npz"))
    _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    got = gpu_ctx.anchor_chain(graphs[0], graphs[1], ms, max_num_match_pairs=int(z["max_num_match_pairs"][0]), score_scale=float(z["score_scale"][0]),
                               fill_in=fill, params=capi.default_chain_params(global_anchoring=glob), chaining_algorithm=1)
    pre = "m%d.%s." % (m, tag)
    assert got["scale"] == float(zg[pre + "scale"]) == 1.0
    for k in ("set_order", "chain", "walk_off", "walk1", "walk2", "count1", "count2", "full_length", "gap_before", "gap_after",
              "gap_score_before", "gap_score_after", "score"):
        assert np.array_equal(got[k], zg[pre + k]), k
    assert len(got["chain"]) != len(z[("f" if fill else "l") + ".chain"])      # (not the SparseAffine chain)


@pytest.mark.parametrize("name", ANCHOR_FILES)
def test_default_anchor_chain_with_splitting(gpu_ctx, name):
    """the default-configured Anchorer::anchor_chain: split_branching_matches, then chaining with fill-in, global anchoring"""
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("anchor4_", "stitch4_"))
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    split = capi.split_branching_matches(graphs[0], graphs[1], ms, 5, 30, 1, 16)
    got = gpu_ctx.anchor_chain(graphs[0], graphs[1], split, max_num_match_pairs=int(z["max_num_match_pairs"][0]),
                               score_scale=float(z["score_scale"][0]))
    assert got["scale"] == float(z["sf.scale"])
    for k in ("set_order", "chain", "walk_off", "walk1", "walk2", "count1", "count2", "full_length", "gap_before", "gap_after",
              "gap_score_before", "gap_score_after", "score"):
        assert np.array_equal(got[k], z["sf." + k]), k


def test_core_align_end_to_end_matches_reference(gpu_ctx):
    """Core::align (core.hpp:181-252) from the reference's match sets of the root merge of a 4-sequence MSA to the merge's
    alignment: the same partitioned anchor segments and the same stitched alignment as the reference's own run"""
    z = np.load(os.path.join(H.GOLDEN, "align4_30k_merge2.npz"))
    _, graphs, _ = load_stitch_case("stitch4_30k_merge2.npz")
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    got = gpu_ctx.core_align(graphs[0], graphs[1], ms, score_scale=float(z["score_scale"][0]), max_num_match_pairs=40000)
    for k in ("seg_off", "walk_off", "walk1", "walk2"):
        assert np.array_equal(got[k], z[k]), k
    assert np.array_equal(got["alignment"].reshape(-1), z["stitched"])
    assert len(got["walk_off"]) > 1000


def test_core_align_small_and_degenerate_inputs(gpu_ctx):
    """cl_core_align on tiny leaf graphs: no matches at all, a handful of matches, a chain below the segment threshold"""
    from centrolign_amd import synth
    rng = np.random.default_rng(3)
    a = rng.integers(0, 4, 400).astype(np.uint8)
    b = a.copy()
    b[rng.choice(400, 12, replace=False)] ^= 1
    g1, g2 = synth.base_graph_from_sequence(a), synth.base_graph_from_sequence(b[:380])
    empty = capi.MatchSets(set_off1=[0], walk_off1=[0], nodes1=[], set_off2=[0], walk_off2=[0], nodes2=[], count1=[], count2=[], full_length=[])
    # permissive partition thresholds so that the short chain survives
    def tweak(ap):
        ap.partition.minimum_segment_score = 1.0
        ap.partition.window_length = 50.0
    for ms in (empty, synth.exact_matches(a, b[:380], k=10)):
        for tw in (None, tweak):
            got = gpu_ctx.core_align(g1, g2, ms, tweak=tw)
            aln = got["alignment"]
            gap = np.uint64(2 ** 64 - 1)
            # a global alignment: every base of both sequences exactly once, in order
            assert np.array_equal(aln[aln[:, 0] != gap, 0], np.arange(400, dtype=np.uint64))
            assert np.array_equal(aln[aln[:, 1] != gap, 1], np.arange(380, dtype=np.uint64))
    assert len(got["walk_off"]) > 5   # the last run (matches + permissive thresholds) did anchor


@pytest.mark.parametrize("name", FILES)
def test_gpu_sparse_chain_is_optimal(gpu_ctx, name):
    """the device's gap-free chain reaches the total weight of the reference's exhaustive O(M^2) chaining (row a23)"""
    from tests.test_chain_oracle import chain_weight
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    full = capi.MatchSets(**{k: z["a.ms." + k] for k in capi.MatchSets._DT})
    for seed, budget in ((1, 1500), (2, 3000)):
        ms = po.budget_subset(full, budget, seed=seed)
        got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, sparse=True, want_dp=True)
        want = float(z["exhaustive.%d.%d" % (seed, budget)][0])
        assert abs(chain_weight(ms, got["chain"]) - want) < 1e-6
        assert abs(float(got["dp"].max()) - want) < 1e-3 * max(1.0, want)   # the float DP value of the chain's last anchor


@pytest.mark.parametrize("mode, message", [("1", "on the per-block kernels"), ("first", "alone on the device")])
def test_walk_stall_falls_back_to_the_per_block_kernels(mode, message):
    """the walk kernel needs all its workgroups resident; if one of its bounded waits ever expires, the DP is repeated — once more on the walk
    kernels with the device's walk budget booked for it alone, then on the per-block kernels.  CL_CHAIN_DEBUG_STALL=1 makes every walk attempt
    report a stall, =first only the shared one (the variable is read once, hence the child process): the multi-path golden chains must still
    come out either way"""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from centrolign_amd import capi\n"
        "from tests import helpers as H\n"
        "from tests.test_extraction import load_stitch_case\n"
        "name = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith('chain4_'))[-1]\n"
        "z = np.load(os.path.join(H.GOLDEN, name))\n"
        "_, graphs, _ = load_stitch_case(name.replace('chain4_', 'stitch4_'))\n"
        "ctx = capi.Context(0)\n"
        "for tag in ('a', 'b'):\n"
        "    ms = capi.MatchSets(**{k: z['%%s.ms.%%s' %% (tag, k)] for k in capi.MatchSets._DT})\n"
        "    got = ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=float(z[tag + '.scale'][0]), params=capi.default_chain_params(global_anchoring=True))\n"
        "    assert np.array_equal(got['chain'], z[tag + '.chain_affine_global']), tag\n"
        "print('fallback ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, CL_CHAIN_DEBUG_STALL=mode, CL_CHAIN_TIMING="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "walk kernel stalled: repeating the DP " + message in r.stderr
    assert ("on the per-block kernels" in r.stderr) == (mode == "1")


def test_match_sets_with_nodes_outside_their_graph_are_refused(gpu_ctx):
    """the chaining DP indexes per-node tables with the first and last node of every walk: a set whose walk is empty or names a node the graph does not have is an
    error of the caller's (CL_ERR_INVALID_ARGUMENT with the set's number), not a crash — found by scripts/fuzz_msa.py on an input the reference dies on"""
    name = FILES[0]
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    ms = capi.MatchSets(**{k: z["a.ms." + k] for k in capi.MatchSets._DT})
    for field, graph in (("nodes1", graphs[0]), ("nodes2", graphs[1])):
        bad = capi.MatchSets(**{k: getattr(ms, k).copy() for k in capi.MatchSets._DT})
        getattr(bad, field)[-1] = len(graph.label) + 7
        with pytest.raises(capi.ClError) as e:
            gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], bad)
        assert "node id outside its graph" in str(e.value)
    got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=float(z["a.scale"][0]), params=capi.default_chain_params(global_anchoring=False))
    assert np.array_equal(got["chain"], z["a.chain_affine"])   # (and the context is as good as before)
