"""a CASE of tests/golden/make_cyclize_wide.py through cl_msa -c on the GPU: wall clock, digest, device memory; compares with the committed golden when there is one.
usage: python scripts/dev/cyc_wide_run.py CASE [workers]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_cyclize_wide as W  # noqa: E402
from centrolign_amd import capi  # noqa: E402

case = sys.argv[1]
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, seed, length, dup, carriers, hor_div, min_len, budget, prefix, minutes = W.CASES[case]
names, seqs, newick = W.workload(case)
fasta = "".join(">%s\n%s\n" % (nm, sq) for nm, sq in zip(names, seqs))
ctx = capi.Context(0)
t0 = time.time()
text, st = ctx.msa(fasta, newick=newick, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len, workers=workers)
wall = time.time() - t0
m = ctx.memory_stats()
out = dict(case=case, workers=workers, wall_s=wall, gfa_bytes=len(text), gfa_sha256=hashlib.sha256(text).hexdigest(), input_sha256=hashlib.sha256("".join(seqs).encode()).hexdigest(),
           stats={k: v for k, v in st.items() if isinstance(v, (int, float))}, device_peak_GB=m["peak_bytes"] / 1e9)
gold = os.path.join(ROOT, "tests", "golden", case + ".json")
if os.path.exists(gold):
    g = json.load(open(gold))
    out["reference_sha256"] = g["gfa"]["sha256"]
    out["identical_to_the_reference"] = g["gfa"]["sha256"] == out["gfa_sha256"]
print(json.dumps(out))
