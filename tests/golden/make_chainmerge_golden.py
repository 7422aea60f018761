"""Golden outputs of Anchorer::anchor_chain with chaining_algorithm = Sparse (the CLI's hidden -g 1) over ChainMerge structures, as Core::execute
runs it for that setting (include/centrolign/core.hpp:350-357, include/centrolign/chain_merge.hpp:100-225): the compiled reference
(oracle/_ref, oracle/ref_driver.cpp: ref_anchor_chain_algo) on the inputs of tests/golden/anchor4_30k_merge*.npz (match sets) and
stitch4_30k_merge*.npz (the two graphs of the merge).  Tags: "s" global anchoring with fill-in (the CLI's configuration apart from -g),
"sl" local anchoring without fill-in.  usage (build container): python tests/golden/make_chainmerge_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests.test_extraction import load_stitch_case  # noqa: E402


G1_EXTRA = ("msa4_30k", 4, 30000, 11, 60000)      # (a case where -g 1 prints another graph than the default: checked below)


def main():
    out = {}
    for m in range(3):
        z = np.load(os.path.join(HERE, "anchor4_30k_merge%d.npz" % m))
        _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        ms = po.MatchSets(**{k: z["ms." + k] for k in po.MatchSets._DT})
        for tag, glob, fill in (("s", True, True), ("sl", False, False)):
            r = po.ref_anchor_chain(graphs[0], graphs[1], ms, max_num_match_pairs=int(z["max_num_match_pairs"][0]), score_scale=float(z["score_scale"][0]),
                                    global_anchoring=glob, fill_in=fill, chaining_algorithm=1)
            for k, v in r.items():
                out["m%d.%s.%s" % (m, tag, k)] = np.asarray(v)
            print("merge %d %s: %d anchors (SparseAffine/PathMerge had %d), paths %d + %d" % (m, tag, len(r["chain"]), len(z[("f" if fill else "l") + ".chain"]),
                                                                                             len(graphs[0].path_off) - 1, len(graphs[1].path_off) - 1))
    # the whole CLI flow with -g 1 (src/main.cpp:129), by the compiled reference's own Core::execute (oracle/_ref/ref_cli): the end-to-end cases
    # of tests/helpers.py:msa_cases() with more than two sequences
    import subprocess
    import tempfile
    from centrolign_amd import msa, synth
    from tests import helpers as H
    cli = os.path.join(ROOT, "oracle", "_ref", "ref_cli")
    with tempfile.TemporaryDirectory() as d:
        for name, n, length, seed, budget in H.msa_cases() + [G1_EXTRA]:
            seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
            names = ["seq%d" % i for i in range(n)]
            synth.write_fasta(os.path.join(d, name + ".fa"), seqs, names)
            open(os.path.join(d, name + ".nwk"), "w").write(msa.newick(msa.balanced_tree(names)) + ";\n")
            subprocess.check_call([cli, os.path.join(d, name + ".fa"), os.path.join(d, name + ".nwk"), "-", os.path.join(d, name + ".out"), str(budget), "0", "0",
                                   "i:chaining_algorithm=1"])
            out["cli." + name] = np.frombuffer(open(os.path.join(d, name + ".out"), "rb").read(), np.uint8)
            print("cli -g 1 %s: %d bytes" % (name, len(out["cli." + name])))
            if name == G1_EXTRA[0]:
                subprocess.check_call([cli, os.path.join(d, name + ".fa"), os.path.join(d, name + ".nwk"), "-", os.path.join(d, name + ".g2"), str(budget), "0"])
                assert open(os.path.join(d, name + ".g2"), "rb").read() != bytes(out["cli." + name])
    np.savez_compressed(os.path.join(HERE, "chainmerge4_30k_g1.npz"), **out)


if __name__ == "__main__":
    main()
