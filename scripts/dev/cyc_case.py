"""one -c case of scripts/fuzz_msa.py step by step against the compiled reference's recorded flow (oracle/ref_driver.cpp: ref_cyclize_dump), to find the FIRST step that differs.
  make  (build container, CPU): python scripts/dev/cyc_case.py make '<json of the case as fuzz_msa.py prints it>' NAME  -> bench_data/cyc_cases/NAME.npz
  check (GPU box):              python scripts/dev/cyc_case.py check NAME [NAME ...]"""
import json
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from centrolign_amd import capi, synth  # noqa: E402

DIR = os.path.join(ROOT, "bench_data", "cyc_cases")
CHAIN_KEYS = ("walk_off", "walk1", "walk2", "score", "gap_after", "gap_score_after")


def sequences(p):
    kw = dict(seq_div=p["seq_div"], hor_div=p["hor_div"])
    return synth.tandem_dup_sequences(p["seed"], p["length"], p["n"], p["dup"], carriers=p["carriers"], **kw), ["q%02d" % i for i in range(p["n"])]


def make(p, name):
    from oracle import pyoracle as po
    seqs, names = sequences(p)
    overrides = "i:min_cyclizing_length=%d;i:max_num_match_pairs=%d" % (p["min_cyclizing_length"], p["budget"])
    with tempfile.TemporaryDirectory() as tmp:
        fa, nw = os.path.join(tmp, "in.fa"), os.path.join(tmp, "t.nwk")
        synth.write_fasta(fa, seqs, names)
        open(nw, "w").write(p["newick"] + "\n")
        po.ref_cyclize_dump(fa, nw, os.path.join(tmp, "d.bin"), os.path.join(tmp, "dump.gfa"), overrides)
        d = po.read_dump(os.path.join(tmp, "d.bin"))
        import hashlib
        text = open(os.path.join(tmp, "dump.gfa"), "rb").read()
        print(name, "reference GFA sha", hashlib.sha256(text).hexdigest(), "(the fuzz run wanted", p.get("want_sha256"), ")")
    d["case_json"] = np.frombuffer(json.dumps(p).encode(), np.uint8)
    np.savez_compressed(os.path.join(DIR, name + ".npz"), **d)
    print(name, "bonds per leaf:", [int(d["leaf%d.counts" % i][1]) for i in range(p["n"])], "inconsistencies:", len(d["inconsistencies"]) // 2,
          "nodes:", [len(d[k + "label"]) for k in ("msa.", "fused.", "simplified.", "polished.")])


def graph_of(d, pre):
    t = d[pre + "tableau"]
    return capi.BaseGraph(*[d[pre + k] for k in capi.GRAPH_KEYS], int(t[0]), int(t[1]))


def check(name):
    z = np.load(os.path.join(DIR, name + ".npz"))
    d = {k: z[k] for k in z.files}
    p = json.loads(d["case_json"].tobytes().decode())
    seqs, names = sequences(p)
    budget, min_len = p["budget"], p["min_cyclizing_length"]
    ctx = capi.Context(0)
    ok = True

    def say(step, good, extra=""):
        nonlocal ok
        ok = ok and good
        print("%s  %-46s %s %s" % (name, step, "same" if good else "DIFFERS", extra), flush=True)
    leaves = [capi.leaf_graph(s) for s in seqs]
    memos, scales = [], []
    for leaf in leaves:
        sc, h = ctx.leaf_calibrate(leaf, max_num_match_pairs=budget)
        scales.append(sc); memos.append(h)
    say("intrinsic scales", np.array_equal(np.array(scales).view(np.uint64), d["intrinsic_scales"].view(np.uint64)))
    mean = sum(scales) / len(scales)
    say("mean scale", mean == float(d["score_scale"][0]))
    bp = capi.bond_params(min_length=min_len)
    for i, leaf in enumerate(leaves):
        got = ctx.leaf_bond_alignments(leaf, memos[i], mean, max_num_match_pairs=budget, bonds=bp)
        n_want = int(d["leaf%d.counts" % i][1])
        good = len(got) == n_want and all(np.array_equal(a, d["leaf%d.bond_aln%d" % (i, b)].reshape(-1, 2)) for b, a in enumerate(got))
        say("leaf %d bond alignments (%d)" % (i, n_want), good, "" if good else "got %d" % len(got))
    for h in memos:
        ctx.free_leaf_calibration(h)
    fasta = "".join(">%s\n%s\n" % (nm, s) for nm, s in zip(names, seqs))
    text = d["output"].tobytes()
    path_names = re.findall(r"^P\t(\S+)", text.decode(), re.M)
    # the acyclic MSA
    got_msa, _ = ctx.msa(fasta, newick=p["newick"], max_num_match_pairs=budget)
    want_msa = capi.write_gfa(graph_of(d, "msa."), path_names)
    say("MSA graph (GFA text of the acyclic run)", got_msa == want_msa)
    # apply_bonds on the reference's MSA graph
    alns, owner = [], []
    for i in range(p["n"]):
        for b in range(int(d["leaf%d.counts" % i][1])):
            alns.append(d["leaf%d.bond_aln%d" % (i, b)].reshape(-1, 2)); owner.append(i)
    if alns:
        path_of = [path_names.index(names[i]) for i in owner]
        say("apply_bonds on the reference's MSA graph", capi.graphs_equal(capi.apply_bonds(graph_of(d, "msa."), path_of, alns), graph_of(d, "simplified.")))
        say("simplify_bubbles on the reference's fused graph", capi.graphs_equal(capi.simplify_bubbles(graph_of(d, "fused.")), graph_of(d, "simplified.")))
    got_inc = capi.identify_inconsistencies(graph_of(d, "simplified."))
    say("inconsistencies (%d)" % (len(d["inconsistencies"]) // 2), np.array_equal(got_inc, d["inconsistencies"].reshape(-1, 2)), "" if np.array_equal(got_inc, d["inconsistencies"].reshape(-1, 2)) else "got %d" % len(got_inc))
    pol, n_regions = ctx.polish_cyclized_graph(graph_of(d, "simplified."), path_names, names, float(d["score_scale"][0]), newick=p["newick"], max_num_match_pairs=budget)
    say("polishing from the reference's simplified graph", capi.graphs_equal(pol, graph_of(d, "polished.")) and capi.write_gfa(pol, path_names) == text, "%d regions" % n_regions)
    got, st = ctx.msa(fasta, newick=p["newick"], max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len, workers=p["workers"])
    say("the whole -c flow (cl_msa)", got == text, json.dumps({k: st[k] for k in ("n_bonds", "n_polished_regions")}))
    ctx.close()
    return ok


if __name__ == "__main__":
    if sys.argv[1] == "make":
        os.makedirs(DIR, exist_ok=True)
        make(json.loads(sys.argv[2]), sys.argv[3])
    else:
        res = [check(n) for n in sys.argv[2:]]
        sys.exit(0 if all(res) else 1)
