"""a -c case of scripts/fuzz_msa.py call by call, every call announced before it is made (to find the call a crash happens in).  usage: python scripts/dev/crash_case.py '<json>'"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, synth  # noqa: E402

p = json.loads(sys.argv[1])
seqs = synth.tandem_dup_sequences(p["seed"], p["length"], p["n"], p["dup"], carriers=p["carriers"], seq_div=p["seq_div"], hor_div=p["hor_div"])
names = ["q%02d" % i for i in range(p["n"])]


def say(s):
    print(s, flush=True)
    sys.stderr.write("[crash_case] " + s + "\n"); sys.stderr.flush()


if os.environ.get("SEGV_BT"):
    import ctypes
    import subprocess
    subprocess.check_call(["cc", "-shared", "-fPIC", "-o", "/tmp/segv_bt.so", os.path.join(ROOT, "scripts", "dev", "segv_bt.c")])
    ctypes.CDLL("/tmp/segv_bt.so").install()
ctx = capi.Context(0)
leaves = [capi.leaf_graph(s) for s in seqs]
say("lengths %s" % [len(s) for s in seqs])
memos, scales = [], []
for i, leaf in enumerate(leaves):
    say("leaf_calibrate %d" % i)
    sc, h = ctx.leaf_calibrate(leaf, max_num_match_pairs=p["budget"])
    scales.append(sc); memos.append(h)
mean = sum(scales) / len(scales)
bp = capi.bond_params(min_length=p["min_cyclizing_length"])
for i, leaf in enumerate(leaves):
    say("leaf_bond_alignments %d" % i)
    got = ctx.leaf_bond_alignments(leaf, memos[i], mean, max_num_match_pairs=p["budget"], bonds=bp)
    say("   -> %d bond alignments, %s pairs" % (len(got), [len(a) for a in got]))
all_alns, owner = [], []
for i, leaf in enumerate(leaves):
    for a in ctx.leaf_bond_alignments(leaf, memos[i], mean, max_num_match_pairs=p["budget"], bonds=bp):
        all_alns.append(a); owner.append(i)
from centrolign_amd import msa as M  # noqa: E402
say("progressive msa (python driver)")
tree = M.tree_of_plan(p["newick"], names)
r = M.progressive_msa(ctx, dict(zip(names, seqs)), tree, max_num_match_pairs=p["budget"])
root, paths = r["root"], r["paths"]
say("   -> root graph %d nodes, paths %s" % (len(root.label), paths))
import numpy as np  # noqa: E402
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "crash_inputs.npz"), **{k: getattr(root, k) for k in capi.GRAPH_KEYS}, tableau=np.array([root.src_id, root.snk_id]),
                    owner=np.array(owner), **{"aln%d" % k: a for k, a in enumerate(all_alns)}, paths=np.array(paths))
say("apply_bonds")
g2 = capi.apply_bonds(root, [paths.index(names[i]) for i in owner], all_alns)
say("   -> %d nodes" % len(g2.label))
say("identify_inconsistencies")
inc = capi.identify_inconsistencies(g2)
say("   -> %d regions" % len(inc))
say("polish_cyclized_graph")
pol, nreg = ctx.polish_cyclized_graph(g2, paths, names, r["scale"], newick=p["newick"], max_num_match_pairs=p["budget"])
say("   -> %d nodes, %d regions" % (len(pol.label), nreg))
fasta = "".join(">%s\n%s\n" % (nm, s) for nm, s in zip(names, seqs))
say("msa without -c")
text, st = ctx.msa(fasta, newick=p["newick"], max_num_match_pairs=p["budget"])
say("   -> %d bytes" % len(text))
say("msa with -c, one worker")
text, st = ctx.msa(fasta, newick=p["newick"], max_num_match_pairs=p["budget"], cyclize=True, min_cyclizing_length=p["min_cyclizing_length"])
say("   -> %d bytes, %s" % (len(text), {k: st[k] for k in ("n_bonds", "n_polished_regions")}))
say("msa with -c, %d workers" % p["workers"])
text, st = ctx.msa(fasta, newick=p["newick"], max_num_match_pairs=p["budget"], cyclize=True, min_cyclizing_length=p["min_cyclizing_length"], workers=p["workers"])
say("   -> %d bytes" % len(text))
