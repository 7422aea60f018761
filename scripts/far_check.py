"""A/B of the chaining DP's far pass on the 2 x 1 Mbp input: every DP value and the chain with the branch-and-bound far pass
(default) and with the all-pairs sweep (CL_CHAIN_NO_FAR_PRUNE=1), against the CPU oracle's values where bench_data holds them.
usage: python scripts/far_check.py out.npz [sparse]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from centrolign_amd import capi


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(here, "bench_data", "c2_chain_input.npz"))
    graphs = []
    for side in ("parent1.", "parent2."):
        t = z[side + "tableau"]
        graphs.append(capi.BaseGraph(*[z[side + k] for k in ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes")], t[0], t[1]))
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    ctx = capi.Context(0)
    sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
    for rep in range(2):
        t0 = time.time()
        if sparse:
            got = ctx.chain_sparse_affine(graphs[0], graphs[1], ms, want_dp=True, sparse=True)
        else:
            got = ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=0.47071162112976217, want_dp=True)
        print("%s: %d anchors, device %.1f ms, prep %.0f index %.0f traceback %.0f ms, wall %.2f s" %
              ("sparse" if sparse else "affine", len(got["chain"]), got["device_ms"], got["prep_ms"], got["index_ms"], got["traceback_ms"], time.time() - t0), flush=True)
    np.savez(sys.argv[1], dp=got["dp"], chain=got["chain"])
    op = os.path.join(here, "bench_data", "c2_oracle_dp.npy")
    if os.path.exists(op) and not sparse:
        odp = np.load(op)[:len(got["dp"])]
        same = np.array_equal(odp.view(np.uint32), got["dp"].view(np.uint32))
        print("dp == oracle dp (bitwise):", same, "chain == oracle chain:", np.array_equal(np.load(os.path.join(here, "bench_data", "c2_oracle_chain.npy")), got["chain"]))
        if not same:
            bad = np.nonzero(odp.view(np.uint32) != got["dp"].view(np.uint32))[0]
            print("  %d differ; first %s: oracle %s got %s" % (len(bad), bad[:5], odp[bad[:5]], got["dp"][bad[:5]]))


if __name__ == "__main__":
    main()
