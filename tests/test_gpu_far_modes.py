"""GPU suite: the far pass of the chaining DP (chain_far.hip) skips work on the strength of a floating-point bound and may hand the rest
of a DP to the all-pairs sweep at a checkpoint.  Whatever path is taken, every DP value and the chain must be bit-identical: the same
multi-combination DP (2 + 2 paths, several hundred macro-blocks, repeats dense enough that the automatic choice switches to the sweep
in mid-DP) is run with the automatic choice, with the branch-and-bound pinned, with the sweep pinned, without the far pass, and on the
round-1 per-block kernels — each in a process of its own, because the switches are read once."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from centrolign_amd import capi, synth
from oracle import pyoracle as po
from tests import far_ab_child

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    ("auto", {}),
    ("bb", {"CL_CHAIN_FAR_MODE": "bb"}),
    ("sweep", {"CL_CHAIN_FAR_MODE": "sweep"}),
    ("no_far", {"CL_CHAIN_NO_FAR_PRUNE": "1"}),
    ("per_block", {"CL_CHAIN_NO_FAR_PRUNE": "1", "CL_CHAIN_OLD_WALK": "1"}),
    # the walk's workgroups exchange their candidates through one atomic maximum + arrival count per pair (the way of merges with more than
    # 16 chain combinations) instead of reading one another's granules
    ("reduce", {"CL_CHAIN_WALK_REDUCE": "1"}),
    ("granules", {"CL_CHAIN_WALK_REDUCE": "0"}),
    # lanes per far query: 32 by default at this width (four groups of eight per query); one and two groups
    ("lanes8", {"CL_CHAIN_FAR_LANES": "8"}),
    ("lanes16", {"CL_CHAIN_FAR_LANES": "16"}),
    # round 4: the walk of a macro-block over several compute units per combination (chain_walk2.hip) is the default; one compute unit per
    # combination as in rounds 2-3; a window of 256 instead of 128 queries; no helper workgroups at all (the main workgroup then evaluates
    # everything itself out of LDS: the path it takes whenever a helper is late); too few helpers (some sub-blocks helped, some not)
    ("walk1", {"CL_CHAIN_WALK2": "0"}),
    ("walk2_window256", {"CL_CHAIN_WALK2_QPT": "2"}),
    ("walk2_no_helpers", {"CL_CHAIN_WALK2_HELPERS": "0"}),
    ("walk2_two_helpers", {"CL_CHAIN_WALK2_HELPERS": "2"}),
    ("walk2_reduce", {"CL_CHAIN_WALK_REDUCE": "1", "CL_CHAIN_WALK2_QPT": "2"}),
    # large nodes of the far pass sealed by one wave each (rounds 2-3) instead of a workgroup each (far_seal_big_kernel); the near sweep as one launch
    ("seal_by_waves", {"CL_CHAIN_SEAL_WAVE": "1", "CL_CHAIN_FAR_MODE": "bb"}),
    ("near_in_two_launches", {"CL_CHAIN_NEAR_SPLIT": "1"}),
    # the events between the streams as runtime calls of their own (rounds 2-3) instead of riding on the launches they follow
    ("event_records", {"CL_CHAIN_EXT_EVENTS": "0"}),
    ("far_sent_at_its_own_block", {"CL_CHAIN_FAR_WAIT_LATE": "1"}),   # round 4's order: far(k) enqueued at iteration k, its event taken between near(k) and walk(k)
    ("far_sent_at_its_own_block_bb", {"CL_CHAIN_FAR_WAIT_LATE": "1", "CL_CHAIN_FAR_MODE": "bb"}),
    # the traceback fetches the query results of the pair it stands on step by step (the way of merges with hundreds of combinations, whose
    # results would be tens of gigabytes) instead of downloading all of them
    ("traceback_rows_on_demand", {"CL_CHAIN_LAZY_ACC": "1"}),
    # two and three combinations per workgroup of the walk (chain_walk_fold_kernel: the way of merges with more than 256 combinations)
    # the dense query tables built on the host as in rounds 1-3 (by default the host keeps their factors and the device multiplies them out)
    ("dense_queries", {"CL_CHAIN_DENSE_QUERIES": "1"}),
    ("walk_fold2", {"CL_CHAIN_WALK_FOLD": "2"}),
    ("walk_fold3", {"CL_CHAIN_WALK_FOLD": "3", "CL_CHAIN_FAR_MODE": "bb"}),
    # round 5: the far pass's search structures beyond 2^32 words of their arena (table entries count blocks of 8 words: the root of a 50-sequence MSA needs
    # 1.2 x 10^10 words): 4.4 x 10^9 words are left unused in front of this DP's structures (17.6 GB that are allocated and never touched)
    ("arena_beyond_32_bits", {"CL_CHAIN_FAR_ARENA_SKIP": "4400000000", "CL_CHAIN_FAR_MODE": "bb"}),
    ("arena_beyond_32_bits_fold3", {"CL_CHAIN_FAR_ARENA_SKIP": "9000000008", "CL_CHAIN_FAR_MODE": "bb", "CL_CHAIN_WALK_FOLD": "3"}),
]


@pytest.fixture(scope="module")
def dense_input(gpu_ctx, tmp_path_factory):
    """root merge of a 4 x 100 kbp MSA: 2 + 2 paths; ~350 k match pairs out of the millions the 100 kbp arrays have"""
    seqs = synth.hor_sequences(11, 100000, 4)
    leaves = [capi.leaf_graph(s) for s in seqs]
    scale = sum(gpu_ctx.leaf_intrinsic_scale(g) for g in leaves) / 4
    left = gpu_ctx.merge(leaves[0], leaves[1], score_scale=scale)["fused"]
    right = gpu_ctx.merge(leaves[2], leaves[3], score_scale=scale)["fused"]
    from bench import relabelled
    g1, g2 = relabelled(left, 5, 6), relabelled(right, 7, 8)
    ms = po.budget_subset(gpu_ctx.find_matches(g1, g2), 350000, seed=5)
    assert ms.n_pairs() > 300000
    path = str(tmp_path_factory.mktemp("far_ab") / "dense.npz")
    far_ab_child.save_input(path, g1, g2, ms, 0.7)
    return path


def run_variant(path, kind, env_extra):
    env = dict(os.environ, CL_CHAIN_TIMING="1", **env_extra)
    for k in ("CL_CHAIN_FAR_MODE", "CL_CHAIN_NO_FAR_PRUNE", "CL_CHAIN_OLD_WALK", "CL_CHAIN_WALK_REDUCE", "CL_CHAIN_FAR_LANES", "CL_CHAIN_WALK2", "CL_CHAIN_WALK2_QPT",
              "CL_CHAIN_WALK2_HELPERS", "CL_CHAIN_SEAL_WAVE", "CL_CHAIN_NEAR_SPLIT", "CL_CHAIN_EXT_EVENTS", "CL_CHAIN_LAZY_ACC", "CL_CHAIN_WALK_FOLD", "CL_CHAIN_DENSE_QUERIES", "CL_CHAIN_FAR_WAIT_LATE",
              "CL_CHAIN_GROUP_PATH"):
        if k not in env_extra:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "far_ab_child.py"), path, kind], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r"RESULT pairs=(\d+) anchors=(\d+) dp=(\w+) chain=(\w+)", r.stdout)
    assert m, r.stdout[-2000:]
    return m.groups(), r.stderr


@pytest.mark.parametrize("kind", ["affine", "sparse"])
def test_every_far_mode_gives_the_same_dp_values_and_chain(dense_input, kind):
    results = {}
    for name, env in VARIANTS:
        results[name], err = run_variant(dense_input, kind, env)
        if name == "auto" and kind == "affine":
            # the automatic choice took its decision at a checkpoint in mid-DP, and on this input it is the sweep
            assert "far pass after" in err and "all-pairs sweep" in err, err[-3000:]
        if name == "bb":
            assert "all-pairs sweep" not in err
    want = results["per_block"]
    assert int(want[0]) > 300000 and int(want[1]) > 10
    for name, got in results.items():
        assert got == want, "%s differs from the per-block all-pairs path: %s vs %s" % (name, got, want)


@pytest.mark.parametrize("kind", ["affine", "sparse"])
def test_the_group_by_group_path_of_wide_merges_gives_the_same_dp_values(gpu_ctx, tmp_path, kind):
    """round 6: chaining DPs over more than 768 combinations run group by group, parallel over the combinations (chain_group_reduce / _store / _push kernels; the -c
    goldens of 50 sequences hold fourteen such DPs and are compared as whole GFAs in tests/test_cyclize_flow.py).  Here the path is FORCED (CL_CHAIN_GROUP_PATH=force) on a
    multi-combination DP small enough for an all-pairs sweep — the root merge of a 4 x 30 kbp MSA, 2 + 2 paths, ~40 k match pairs — and every DP value and the chain are
    those of the default path (walk kernels + far pass) and of the round-1 per-block kernels"""
    seqs = synth.hor_sequences(13, 30000, 4)
    leaves = [capi.leaf_graph(s) for s in seqs]
    scale = sum(gpu_ctx.leaf_intrinsic_scale(g) for g in leaves) / 4
    left = gpu_ctx.merge(leaves[0], leaves[1], score_scale=scale)["fused"]
    right = gpu_ctx.merge(leaves[2], leaves[3], score_scale=scale)["fused"]
    from bench import relabelled
    g1, g2 = relabelled(left, 5, 6), relabelled(right, 7, 8)
    ms = po.budget_subset(gpu_ctx.find_matches(g1, g2), 40000, seed=6)
    assert ms.n_pairs() > 20000
    path = str(tmp_path / "group.npz")
    far_ab_child.save_input(path, g1, g2, ms, 0.7)
    want, _ = run_variant(path, kind, {})
    forced, _ = run_variant(path, kind, {"CL_CHAIN_GROUP_PATH": "force"})
    per_block, _ = run_variant(path, kind, {"CL_CHAIN_NO_FAR_PRUNE": "1", "CL_CHAIN_OLD_WALK": "1", "CL_CHAIN_GROUP_PATH": "0"})
    assert int(want[0]) > 20000 and int(want[1]) > 5
    assert forced == want and per_block == want

