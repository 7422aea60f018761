"""chaining DP at BASELINE configs[1] scale: the reference's match sets for the 2 x 1 Mbp pair (dump made by
oracle/ref_driver.cpp), budget-selected as Anchorer::anchor_chain does (anchorer.hpp:1108-1173), then
sparse_affine_chain_dp on the GPU vs the compiled reference.  Developer tool (needs oracle/_ref + the dump)."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from centrolign_amd import capi
from oracle import pyoracle as po


def anchor_weight(p, c1, c2, length, full):
    count = float(c1 * c2)
    frac = float(length) / float(full)
    return frac * (length / math.pow(count, p.pair_count_power) - math.pow(length / p.length_intercept, p.length_decay_power) * p.length_intercept)


def select(ms, params, budget):
    """budgeted greedy selection in (stable) descending order of the full-length anchor weight"""
    n = ms.n_sets
    n1 = np.diff(ms.set_off1.astype(np.int64)); n2 = np.diff(ms.set_off2.astype(np.int64))
    wfull = np.array([anchor_weight(params, int(ms.count1[i]), int(ms.count2[i]), int(ms.full_length[i]), int(ms.full_length[i])) for i in range(n)])
    order = np.argsort(-wfull, kind="stable")
    keep, left = [], budget
    wo1 = ms.walk_off1.astype(np.int64); so1 = ms.set_off1.astype(np.int64)
    for i in order:
        ln = int(wo1[so1[i] + 1] - wo1[so1[i]])
        if anchor_weight(params, int(ms.count1[i]), int(ms.count2[i]), ln, ln) < 0.0:
            break
        pc = int(n1[i] * n2[i])
        if left >= pc:
            keep.append(int(i)); left -= pc
    return po.subset_match_sets(ms, keep)


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(here, "bench_data", "c2_chain_input.npz"))
    run_ref = len(sys.argv) > 1 and sys.argv[1] == "ref"
    graphs = []
    for side in ("parent1.", "parent2."):
        t = z[side + "tableau"]
        graphs.append(capi.BaseGraph(*[z[side + k] for k in ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes")], t[0], t[1]))
    g1, g2 = graphs
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    scale = float(z["score_scale"][0])
    print("%d sets / %d pairs, scale %.4f" % (ms.n_sets, ms.n_pairs(), scale), flush=True)
    ctx = capi.Context(0)
    for rep in range(2):
        t0 = time.time(); got = ctx.chain_sparse_affine(g1, g2, ms, scale=scale); t = time.time() - t0
        print("GPU: chain %d anchors, %d ties, device %.1f ms, prep %.0f index %.0f traceback %.0f ms, wall %.2f s" % (len(got["chain"]), got["n_ties"], got["device_ms"], got["prep_ms"], got["index_ms"], got["traceback_ms"], t), flush=True)
    if run_ref:
        ref, secs = po.ref_chain("affine", g1, g2, ms, scale=scale, global_anchoring=True)
        print("reference: chain %d anchors in %.1f s; identical: %s" % (len(ref), secs, np.array_equal(ref, got["chain"])), flush=True)


if __name__ == "__main__":
    main()
