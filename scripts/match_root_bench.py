"""phase times of cl_find_matches on the merges of the 10 x 1 Mbp MSA (the root: two 5-path graphs, 10 Mbp of path text)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import capi, msa, synth  # noqa: E402


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    names, seqs, tree = synth.c3_workload(length)
    ctx = capi.Context(0)
    r = msa.progressive_msa(ctx, seqs, tree, workers=4, keep_merges=True)
    for k in r["stats"]["kept"][-3:]:
        g1, g2 = k["graphs"]
        for rep in range(2):
            t0 = time.perf_counter()
            ms, st = ctx.find_matches(g1, g2, want_stats=True)
            dt = time.perf_counter() - t0
        print(k["merge"][:50], "sets", ms.n_sets, "wall %.0f ms" % (dt * 1e3), {a: (round(b, 1) if isinstance(b, float) else b) for a, b in st.items()})


if __name__ == "__main__":
    main()
