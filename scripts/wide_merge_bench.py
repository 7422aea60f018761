"""24 sequences x 7 kbp over a balanced tree (root merge 12 + 12 paths = 144 chain combinations): cl_msa wall-clock; CL_CHAIN_OLD_WALK=1 in
the environment shows the per-block path such merges took before the walk kernel's reduction exchange.
usage: python scripts/wide_merge_bench.py [n_seq] [length]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import capi, msa, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
length = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
seqs = synth.hor_sequences(91, length, n, indel_hor=1)
names = ["q%02d" % i for i in range(n)]
fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
tree = msa.newick(msa.balanced_tree(names)) + ";"
ctx = capi.Context(0)
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
for workers in (1, 4):
    t0 = time.time()
    text, st = ctx.msa(fasta, newick=tree, workers=workers)
    print("cl_msa %d x %d, workers %d: %.2f s (align %.2f s summed), sha %s" % (n, length, workers, time.time() - t0, st["align_s"], hashlib.sha256(text).hexdigest()[:16]), flush=True)
