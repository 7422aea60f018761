"""CPU-side checks of round-4 host logic: the device-memory model of the chaining DP (scripts/memory_model.py) and the generator of branching
pairs with far forks (synth.far_fork_batch) that the strip kernel's saved columns are tested with on the GPU."""
import importlib.util
import os

import numpy as np

from centrolign_amd import synth
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _memory_model():
    spec = importlib.util.spec_from_file_location("memory_model", os.path.join(ROOT, "scripts", "memory_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_memory_model_of_the_chaining_dp():
    m = _memory_model()
    # the measured DP of profiles/r04_memory_model.json: 625 combinations, 1 249 987 pairs and records, 25 + 25 chains, gap-free, far pass on, folded walk:
    # the context held 35.7 GB; the model stays within 10 % below it (the block cache hands out blocks of up to twice the request)
    held = 35739.9 * 2 ** 20
    model = m.chain_dp_bytes(1249987, 1249987, 625, 25, 25, True, far_pad=1249987 + 625 * 64, far_levels=2, walk="fold")
    assert 0.88 * held < model < 1.02 * held, (model, held)
    # pair x combination is what dominates a wide merge: 44 bytes each
    assert abs(m.chain_dp_bytes(10 ** 6, 10 ** 6, 400, 20, 20, False) - m.chain_dp_bytes(10 ** 6, 10 ** 6, 200, 20, 20, False) - 200 * (44 * 10 ** 6 + 4 * (10 ** 6 // 256 + 2) + 104 + 8192)) < 10 ** 6
    p = m.predict(50, 5000000, 1250000)
    # round 5 (measured on 50 x 100 kbp and 50 x 1 Mbp, profiles/r05_configs4.json): the gap-free DP keeps one record per pair (35 GB), the affine DP one per pair and
    # chain combination its nodes lie on — 0.64 of pairs x combinations at 625 combinations — 61-71 GB of the 288
    # ... and since the far pass reaches beyond 2^32 arena words (round 5, second half) the root's affine DP runs WITH it: 182 GB held at 50 x 1 Mbp (696 M records,
    # 707 M padded; the model gives 202 GB for those numbers), which took the root merge from 209 s to 56 s
    assert p["combinations"] == 625 and 30e9 < p["dp_bytes"]["gap-free"] < 40e9 and 170e9 < p["dp_bytes"]["affine"] < 230e9, p
    held = 182001755750.4
    model = m.chain_dp_bytes(1250000, 696385926, 625, 26, 26, False, far_pad=707133440, far_levels=4, walk="fold")
    assert 0.95 * held < model < 1.15 * held, (model, held)


def test_far_fork_batch_makes_valid_pairs_with_long_range_edges():
    b = synth.far_fork_batch([(120, 900), (300, 700)], seed=4, n_far=3, far_min=100, far_max=500)
    assert b.n_problems == 2
    s = b.side[1]
    # the second graph of every pair has predecessors hundreds of ranks back in SOME topological order: its edge count exceeds a chain's by the bubbles + forks
    for k in range(2):
        lo, hi = int(s.node_off[k]), int(s.node_off[k + 1])
        n_edges = int(s.prev_off[hi] - s.prev_off[lo])
        assert n_edges >= (hi - lo - 1) + 3
    r = po.oracle_stitch_batch(b)          # the oracle aligns them end to end: every pair has a source-to-sink connection
    assert len(r.score) == 2 and all(int(x) > -10 ** 9 for x in r.score)
    assert len(r.alignment(0)) >= 100 and len(r.alignment(1)) >= 100   # (a path may take a fork and skip hundreds of nodes: no tighter bound)
