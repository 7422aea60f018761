"""fuzzing the -c flow (cl_msa with cyclize) against the compiled reference's CLI flow (oracle/_ref/ref_cli -c) on fresh inputs: random sequence
counts, lengths, tandem duplications (length, divergence, carriers), match budgets and cyclising lengths; prints every disagreement.
usage: python scripts/cyclize_sweep.py first_seed last_seed"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth  # noqa: E402

CLI = os.path.join(ROOT, "oracle", "_ref", "ref_cli")


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    ctx = capi.Context(0)
    bad = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(3, 8))
        length = int(rng.choice([8000, 12000, 16000, 24000]))
        dup = int(rng.choice([2000, 3000, 5000, 8000]))
        dup = min(dup, length // 3)
        carriers = sorted(set(int(x) for x in rng.integers(0, n, int(rng.integers(1, n + 1)))))
        hor_div = float(rng.choice([0.03, 0.05, 0.08, 0.10]))
        dup_div = float(rng.choice([0.001, 0.003, 0.01]))
        min_len = int(rng.choice([1500, 2500, 4000]))
        budget = int(rng.choice([20000, 40000, 80000]))
        workers = int(rng.choice([1, 3]))
        seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=hor_div, dup_div=dup_div)
        if min(len(s) for s in seqs) < 1000:
            print("seed %d: skipped (a short sequence)" % seed, flush=True)
            continue
        names = ["s%d" % i for i in range(n)]
        tree = msa.newick(msa.balanced_tree(names)) + ";"
        fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
        with tempfile.TemporaryDirectory() as tmp:
            synth.write_fasta(os.path.join(tmp, "in.fa"), seqs, names)
            open(os.path.join(tmp, "t.nwk"), "w").write(tree + "\n")
            t0 = time.time()
            try:   # (the reference's polishing step takes a quarter of an hour and more on some inputs: those are left out)
                r = subprocess.run([CLI, os.path.join(tmp, "in.fa"), os.path.join(tmp, "t.nwk"), "-", os.path.join(tmp, "out.gfa"), "0", "0", "0",
                                    "b:cyclize_tandem_duplications=1;i:min_cyclizing_length=%d;i:max_num_match_pairs=%d" % (min_len, budget)], capture_output=True, text=True,
                                   timeout=float(os.environ.get("SWEEP_REF_TIMEOUT", "90")))
            except subprocess.TimeoutExpired:
                print("seed %d: skipped (the reference takes longer than the limit; n %d length %d)" % (seed, n, length), flush=True)
                continue
            t_ref = time.time() - t0
            if r.returncode != 0:
                print("seed %d: reference failed (%s)" % (seed, r.stderr.strip()[-120:]), flush=True)
                continue
            want = open(os.path.join(tmp, "out.gfa"), "rb").read()
        t0 = time.time()
        try:
            got, st = ctx.msa(fasta, newick=tree, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len, workers=workers)
        except capi.ClError as e:
            got, st = b"", dict(n_bonds=-1, n_polished_regions=-1)
            print("seed %d: product failed: %s" % (seed, e), flush=True)
        same = got == want
        note = ""
        if not same and got:
            # The reference's own output depends on the heap layout of its process for some inputs: seed 7016 prints one of two GFAs (same size class,
            # different digests) according to the LENGTH of the path it is given its FASTA file under — the same files, the same options
            # (/tmp/s7016/in.fa against /tmp/tmpabcdefgh/in.fa; identical with address-space randomisation off).  So a mismatch is checked
            # against the reference run from directories of other path lengths before it counts.
            for pad in ("", "x", "xxxxxxxxxx"):
                d2 = tempfile.mkdtemp(prefix="cs" + pad)
                synth.write_fasta(os.path.join(d2, "in.fa"), seqs, names)
                open(os.path.join(d2, "t.nwk"), "w").write(tree + "\n")
                subprocess.run([CLI, os.path.join(d2, "in.fa"), os.path.join(d2, "t.nwk"), "-", os.path.join(d2, "out.gfa"), "0", "0", "0",
                                "b:cyclize_tandem_duplications=1;i:min_cyclizing_length=%d;i:max_num_match_pairs=%d" % (min_len, budget)], capture_output=True, text=True)
                if os.path.exists(os.path.join(d2, "out.gfa")) and open(os.path.join(d2, "out.gfa"), "rb").read() == got:
                    same, note = True, " (the reference prints two different GFAs for this input, by the length of its file paths; this is one of them)"
                    break
        bad += not same
        print("seed %d: n %d length %d dup %d carriers %s hor_div %g dup_div %g min_len %d budget %d workers %d: %d bonds %d regions, %d bytes %s (reference %.0f s, here %.1f s)" %
              (seed, n, length, dup, carriers, hor_div, dup_div, min_len, budget, workers, st["n_bonds"], st["n_polished_regions"], len(want), ("identical" + note) if same else "DIFFERENT", t_ref, time.time() - t0), flush=True)
    print("disagreements:", bad)


if __name__ == "__main__":
    main()
