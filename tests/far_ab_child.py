"""Child process of tests/test_gpu_far_modes.py: the chaining DP's far-pass switches (CL_CHAIN_FAR_MODE, CL_CHAIN_NO_FAR_PRUNE,
CL_CHAIN_OLD_WALK) are read once per process, so every variant runs in a process of its own on the same saved input and prints a
digest of every DP value and of the chain.

usage: python tests/far_ab_child.py INPUT.npz affine|sparse"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from centrolign_amd import capi  # noqa: E402

GRAPH_KEYS = ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes")


def save_input(path, g1, g2, ms, scale):
    d = {"scale": np.array([scale])}
    for tag, g in (("g1.", g1), ("g2.", g2)):
        for k in GRAPH_KEYS:
            d[tag + k] = getattr(g, k)
        d[tag + "tableau"] = np.array([g.src_id, g.snk_id], np.uint64)
    for k in capi.MatchSets._DT:
        d["ms." + k] = getattr(ms, k)
    np.savez(path, **d)


def load_input(path):
    z = np.load(path)
    graphs = []
    for tag in ("g1.", "g2."):
        t = z[tag + "tableau"]
        graphs.append(capi.BaseGraph(*[z[tag + k] for k in GRAPH_KEYS], int(t[0]), int(t[1])))
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    return graphs[0], graphs[1], ms, float(z["scale"][0])


def main():
    g1, g2, ms, scale = load_input(sys.argv[1])
    sparse = sys.argv[2] == "sparse"
    ctx = capi.Context(0)
    got = ctx.chain_sparse_affine(g1, g2, ms, scale=scale, want_dp=True, sparse=sparse, params=capi.default_chain_params(global_anchoring=True))
    print("RESULT pairs=%d anchors=%d dp=%s chain=%s device_ms=%.1f" % (got["n_pairs"], len(got["chain"]),
          hashlib.sha256(got["dp"].view(np.uint32).tobytes()).hexdigest(), hashlib.sha256(np.ascontiguousarray(got["chain"]).tobytes()).hexdigest(), got["device_ms"]))
    ctx.close()


if __name__ == "__main__":
    main()
