#!/bin/bash
# round 6, visit L: where a polishing step's region re-alignments spend their time (the -c flow, 50 x 8 kbp golden, one worker)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6l
mkdir -p $OUT
cd $R
cat > /tmp/polish_one.py <<'P'
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth
CASES = {"cyclize_16x12k": (16, 41, 12000, 4000, [0, 3, 5, 8, 9, 13], "c"), "cyclize_50x8k": (50, 43, 8000, 3000, [1, 4, 7, 12, 18, 23, 29, 31, 36, 40, 44, 48], "d")}
case = sys.argv[1]
n, seed, length, dup, carriers, prefix = CASES[case]
gold = json.load(open(os.path.join(ROOT, "tests", "golden", case + ".json")))
seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=0.08)
names = ["%s%02d" % (prefix, i) for i in range(n)]
fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
tree = msa.newick(msa.balanced_tree(names)) + ";"
ctx = capi.Context(0)
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
t0 = time.time()
text, st = ctx.msa(fasta, newick=tree, max_num_match_pairs=gold["max_num_match_pairs"], cyclize=True, min_cyclizing_length=gold["min_cyclizing_length"], workers=1)
print("workers 1: %.2f s; %s" % (time.time() - t0, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
P
CL_POLISH_TIMING=1 timeout 300 python /tmp/polish_one.py cyclize_50x8k > $OUT/polish.out 2> $OUT/polish.err; tail -2 $OUT/polish.out | cut -c1-600; grep -i "polish\|region" $OUT/polish.err | tail -12 | cut -c1-300
CL_POLISH_TIMING=1 CL_CHAIN_TIMING=1 timeout 300 python /tmp/polish_one.py cyclize_16x12k > $OUT/polish16.out 2> $OUT/polish16.err; tail -1 $OUT/polish16.out | cut -c1-400; wc -l $OUT/polish16.err; gzip -f $OUT/polish16.err
