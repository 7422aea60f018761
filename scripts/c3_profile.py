"""BASELINE configs[2] (10 x 1 Mbp, seed 7, the guide tree of SURVEY.md §8(d)) merge by merge on one context, with the chaining
DP's own phase timings (CL_CHAIN_TIMING=1 prints them to stderr): where the MSA's wall-clock goes.
usage: python scripts/c3_profile.py [n_seq] [length] [workers]"""
import hashlib
import os
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, msa, synth  # noqa: E402

C3_TREE = ((((("s0", "s1"), ("s2", "s3")), "s4")), (((("s5", "s6"), ("s7", "s8")), "s9")))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    t0 = time.perf_counter()
    seqs = synth.hor_sequences(7, length, n)
    names = ["s%d" % i for i in range(n)]
    tree = C3_TREE if n == 10 else msa.balanced_tree(names)
    print("generated %d sequences in %.1f s" % (n, time.perf_counter() - t0), flush=True)
    ctx = capi.Context(0)
    ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
    t0 = time.perf_counter()
    r = msa.progressive_msa(ctx, dict(zip(names, seqs)), tree, workers=workers, verbose=True)
    t1 = time.perf_counter()
    gfa = capi.write_gfa(r["root"], r["paths"])
    print("MSA %.2f s (+ GFA %.2f s); stats %s; root %d nodes; GFA %d bytes sha256 %s" %
          (t1 - t0, time.perf_counter() - t1, r["stats"], len(r["root"].label), len(gfa), hashlib.sha256(gfa).hexdigest()), flush=True)


if __name__ == "__main__":
    main()
