"""the 50 x 5 kbp golden case of tests/test_msa.py (root merge 25 + 25 paths = 625 chain combinations) under CL_CHAIN_TIMING=1: per-DP phase times on stderr,
wall clock and digest on stdout.  usage: CL_CHAIN_TIMING=1 python scripts/dev/wide_root.py [workers] [n] [length]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth  # noqa: E402

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
length = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
seqs = synth.hor_sequences(92, length, n, indel_hor=1)
names = ["r%02d" % i for i in range(n)]
fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "wide_merge_50x5k.json")))
tree = gold["newick"] if (n, length) == (50, 5000) else msa.newick(msa.balanced_tree(names)) + ";"
ctx = capi.Context(0)
m0 = ctx.memory_stats()
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
t0 = time.time()
text, st = ctx.msa(fasta, newick=tree, workers=workers)
sha = hashlib.sha256(text).hexdigest()
print("cl_msa %d x %d, workers %d: %.2f s (align %.2f s summed), sha %s%s" % (n, length, workers, time.time() - t0, st["align_s"], sha[:16],
      (" == reference" if sha == gold["gfa"]["sha256"] else " DIFFERS") if (n, length) == (50, 5000) else ""), flush=True)
m1 = ctx.memory_stats()
print("memory: context peak %.2f GB, cached now %.2f GB; device in use before %.2f GB, now %.2f GB (hipMemGetInfo: total - free), of %.1f GB" %
      (m1["peak_bytes"] / 1e9, m1["cached_bytes"] / 1e9, (m0["device_total_bytes"] - m0["device_free_bytes"]) / 1e9,
       (m1["device_total_bytes"] - m1["device_free_bytes"]) / 1e9, m1["device_total_bytes"] / 1e9), flush=True)
