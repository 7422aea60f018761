"""step-by-step comparison of the -c flow against a dump of the reference's (made by oracle/pyoracle.ref_cyclize_dump): which step differs first
usage: python scripts/dev/cyc_steps.py dump.npz in.fa t.nwk min_len budget"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from centrolign_amd import capi

d = dict(np.load(sys.argv[1]))
fa = open(sys.argv[2]).read()
newick = open(sys.argv[3]).read().strip()
min_len, budget = int(sys.argv[4]), int(sys.argv[5])
recs = fa.split(">")[1:]
names = [r.split("\n", 1)[0] for r in recs]
seqs = ["".join(r.split("\n")[1:]) for r in recs]
ctx = capi.Context(0)
leaves = [capi.leaf_graph(s) for s in seqs]
memos, scales = [], []
for leaf in leaves:
    sc, h = ctx.leaf_calibrate(leaf, max_num_match_pairs=budget)
    scales.append(sc); memos.append(h)
print("intrinsic scales identical:", np.array_equal(np.array(scales).view(np.uint64), d["intrinsic_scales"].view(np.uint64)))
mean = sum(scales) / len(scales)
print("mean scale identical:", mean == float(d["score_scale"][0]))
bp = capi.bond_params(min_length=min_len)
for i, leaf in enumerate(leaves):
    got = ctx.leaf_bond_alignments(leaf, memos[i], mean, max_num_match_pairs=budget, bonds=bp)
    want_n = int(d["leaf%d.counts" % i][1])
    same = len(got) == want_n and all(np.array_equal(a, d["leaf%d.bond_aln%d" % (i, b)].reshape(-1, 2)) for b, a in enumerate(got))
    print("leaf %d: %d bond alignments (reference %d) identical: %s" % (i, len(got), want_n, same))

def graph_of(pre):
    t = d[pre + "tableau"]
    return capi.BaseGraph(*[d[pre + k] for k in capi.GRAPH_KEYS], int(t[0]), int(t[1]))

text = d["output"].tobytes()
path_names = re.findall(r"^P\t(\S+)", text.decode(), re.M)
# the MSA itself
got_text, st = ctx.msa(fa, newick=newick, max_num_match_pairs=budget)
print("plain MSA text: %d bytes" % len(got_text))
alns, owner = [], []
for i in range(len(seqs)):
    for b in range(int(d["leaf%d.counts" % i][1])):
        alns.append(d["leaf%d.bond_aln%d" % (i, b)].reshape(-1, 2)); owner.append(i)
path_of = [path_names.index(names[i]) for i in owner]
got = capi.apply_bonds(graph_of("msa."), path_of, alns)
print("apply_bonds on the reference's MSA graph identical:", capi.graphs_equal(got, graph_of("simplified.")))
inc = capi.identify_inconsistencies(graph_of("simplified."))
print("inconsistencies identical:", np.array_equal(inc, d["inconsistencies"].reshape(-1, 2)), len(inc))
pol, n_regions = ctx.polish_cyclized_graph(graph_of("simplified."), path_names, names, float(d["score_scale"][0]), newick=newick, max_num_match_pairs=budget)
print("polished graph identical:", capi.graphs_equal(pol, graph_of("polished.")), n_regions)
print("GFA identical:", capi.write_gfa(pol, path_names) == text)
full, st = ctx.msa(fa, newick=newick, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len)
print("cl_msa -c identical:", full == text)
