"""fuzzing the reference-free pipeline (centrolign_amd/msa.py over the C ABI) against the compiled reference's whole pipeline
(ref_msa_dump: Core::execute + write_gfa / explicit_cigar) on fresh inputs: random sequence counts, lengths, divergences, match
budgets, tree shapes (balanced, caterpillar, random binary) and worker counts; prints every disagreement"""
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, msa, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def random_tree(rng, names):
    if len(names) == 1:
        return names[0]
    k = int(rng.integers(1, len(names)))
    return (random_tree(rng, names[:k]), random_tree(rng, names[k:]))


def caterpillar(names):
    t = names[0]
    for nm in names[1:]:
        t = (t, nm)
    return t


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ctx = capi.Context(0)
    bad = 0
    t0 = time.time()
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(2, 7))
        length = int(rng.choice([3000, 8000, 20000, 45000]))
        budget = int(rng.choice([5000, 40000, 1250000]))
        kw = dict(seq_div=float(rng.choice([0.003, 0.01, 0.04])), hor_div=float(rng.choice([0.02, 0.05])), indel_hor=int(rng.choice([1, 2, 5])))
        seqs = synth.hor_sequences(seed, length, n, **kw)
        if min(len(q) for q in seqs) < 50:   # whole-HOR deletions can empty a short sequence; the reference asserts on that
            print("seed %d: skipped (a sequence of %d bases)" % (seed, min(len(q) for q in seqs)), flush=True)
            continue
        names = ["s%d" % i for i in range(n)]
        shape = int(rng.integers(0, 3))
        tree = (msa.balanced_tree, caterpillar, lambda nm: random_tree(rng, nm))[shape](names)
        workers = int(rng.choice([1, 2, 4]))
        r = msa.progressive_msa(ctx, dict(zip(names, seqs)), tree, max_num_match_pairs=budget, workers=workers)
        got = msa.output_text(r)
        with tempfile.TemporaryDirectory() as d:
            fa, nwk, out = os.path.join(d, "in.fa"), os.path.join(d, "t.nwk"), os.path.join(d, "out.txt")
            synth.write_fasta(fa, seqs, names)
            open(nwk, "w").write(msa.newick(tree) + ";")
            po.ref_msa_dump(fa, newick_path=nwk, out_path=out, max_num_match_pairs=budget)
            want = open(out, "rb").read()
        if n == 2:
            want = want.rstrip(b"\n")
        ok = got == want
        bad += not ok
        print("seed %d: n %d length %d budget %d %s tree %s workers %d: %d bytes %s" %
              (seed, n, length, budget, kw, msa.newick(tree), workers, len(got), "identical" if ok else "DIFFERENT"), flush=True)
    print("runs", hi - lo, "failures", bad, "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
