"""the -c flow on 16 sequences (tests/golden/cyclize_16x12k.json): cl_msa wall-clock and where it goes"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import capi, msa, synth  # noqa: E402

seqs = synth.tandem_dup_sequences(41, 12000, 16, 4000, carriers=[0, 3, 5, 8, 9, 13], hor_div=0.08)
names = ["c%02d" % i for i in range(16)]
fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
tree = msa.newick(msa.balanced_tree(names)) + ";"
ctx = capi.Context(0)
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
for w in (1, 4):
    t0 = time.time()
    text, st = ctx.msa(fasta, newick=tree, max_num_match_pairs=60000, cyclize=True, min_cyclizing_length=2500, workers=w)
    print("workers %d: %.2f s; %s" % (w, time.time() - t0, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
