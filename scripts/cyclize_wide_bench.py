"""the -c flow on many sequences (tests/golden/cyclize_16x12k.json / cyclize_50x8k.json): cl_msa wall-clock and where it goes
usage: python scripts/cyclize_wide_bench.py [cyclize_16x12k|cyclize_50x8k]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth  # noqa: E402

CASES = {"cyclize_16x12k": (16, 41, 12000, 4000, [0, 3, 5, 8, 9, 13], "c"), "cyclize_50x8k": (50, 43, 8000, 3000, [1, 4, 7, 12, 18, 23, 29, 31, 36, 40, 44, 48], "d")}
case = sys.argv[1] if len(sys.argv) > 1 else "cyclize_16x12k"
n, seed, length, dup, carriers, prefix = CASES[case]
gold = json.load(open(os.path.join(ROOT, "tests", "golden", case + ".json")))
seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=0.08)
names = ["%s%02d" % (prefix, i) for i in range(n)]
fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
tree = msa.newick(msa.balanced_tree(names)) + ";"
ctx = capi.Context(0)
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
for w in (1, 4, 8):
    t0 = time.time()
    text, st = ctx.msa(fasta, newick=tree, max_num_match_pairs=gold["max_num_match_pairs"], cyclize=True, min_cyclizing_length=gold["min_cyclizing_length"], workers=w)
    print("workers %d: %.2f s; %s" % (w, time.time() - t0, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
