"""developer measurement: cl_core_align on the root merge (2 paths x 2 paths) of the 4 x 30 kbp MSA fixture with the CLI's match
budget (every one of the 2.49 M match pairs takes part): exercises the multi-combination chaining path"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centrolign_amd import capi
from tests import helpers as H
from tests.test_extraction import load_stitch_case

z = np.load(os.path.join(H.GOLDEN, "align4_30k_merge2.npz"))
_, graphs, _ = load_stitch_case("stitch4_30k_merge2.npz")
ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
ctx = capi.Context(0)
print("sets", ms.n_sets, "pairs", ms.n_pairs(), flush=True)
for budget in (40000, 400000, 1250000):
    for rep in range(2):
        t0 = time.perf_counter()
        got = ctx.core_align(graphs[0], graphs[1], ms, score_scale=float(z["score_scale"][0]), max_num_match_pairs=budget)
        print("budget %7d: %.2f s (chain %.0f ms, partition %.0f ms, stitch %.0f ms), %d anchors in %d segments, %d aligned pairs" %
              (budget, time.perf_counter() - t0, got["chain_ms"], got["partition_ms"], got["stitch_ms"], len(got["walk_off"]) - 1,
               len(got["seg_off"]) - 1, len(got["alignment"])), flush=True)
