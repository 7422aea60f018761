"""PathMatchFinder::find_matches on the BASELINE pair (2 x 1 Mbp, seed 7): cl_find_matches phase times (device suffix array /
LCP, host tree / query / walk-out) and the digest check against the compiled reference's output (tests/golden/match_finder.npz
"c2.*"; the reference itself needs 4.6 s for this call in the build container)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    seqs = synth.hor_sequences(7, 1000000, 2)
    g1 = synth.base_graph_from_sequence(seqs[0])
    g2 = synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8))
    ctx = capi.Context(0)
    for rep in range(3):
        t0 = time.perf_counter()
        ms, st = ctx.find_matches(g1, g2, want_stats=True)
        t1 = time.perf_counter()
        print("rep %d: %.3f s wall (incl. the copy into numpy); %d sets; text %d, %d rounds, SA %.2f ms, LCP %.2f ms (device); "
              "tree %.1f ms, query %.1f ms, walk-out %.1f ms (host); %d LCP intervals" %
              (rep, t1 - t0, ms.n_sets, st["text_length"], st["doubling_rounds"], st["sa_ms"], st["lcp_ms"], st["tree_ms"], st["query_ms"],
               st["walk_ms"], st["n_internal_nodes"]), flush=True)
    z = np.load(os.path.join(H.GOLDEN, "match_finder.npz"))
    print("identical to the reference's match sets:", ms.n_sets == int(z["c2.n_sets"][0]) and H.match_sets_digest(ms) == str(z["c2.digest"][0]))


if __name__ == "__main__":
    main()
