#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6m
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_cyclize_flow.py tests/test_cyclize.py tests/test_msa.py tests/test_restart.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
CL_POLISH_TIMING=1 timeout 300 python /dev/stdin cyclize_50x8k <<'P' 2> $OUT/polish.err | tail -1 | cut -c1-500
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
for w in (1, 1, 4):
    t0 = time.time()
    text, st = ctx.msa(fasta, newick=tree, max_num_match_pairs=gold["max_num_match_pairs"], cyclize=True, min_cyclizing_length=gold["min_cyclizing_length"], workers=w)
    print("workers %d: %.2f s; cyclize %.2f align %.2f bonds %.2f" % (w, time.time() - t0, st["cyclize_s"], st["align_s"], st["bonds_s"]), flush=True)
P
grep "cl_polish" $OUT/polish.err | cut -c1-200
