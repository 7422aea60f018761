"""-S / -R of the CLI (src/core.cpp:370-422, 1071-1081; src/execution.cpp:190-203, 222-277): subproblem files under their hashed names,
and a restart that continues on what read_gfa + add_sentinels make of them.  Expected values: the compiled reference's CLI flow
(oracle/ref_cli.cpp; tests/golden/make_golden.py restart -> restart_case.npz)."""
import os

import numpy as np
import pytest

from centrolign_amd import capi, msa

HERE = os.path.dirname(os.path.abspath(__file__))


def case():
    z = np.load(os.path.join(HERE, "golden", "restart_case.npz"))
    d = {k: z[k] for k in z.files}
    d["files"] = [str(f) for f in d["files"]]
    d["names"] = [str(n) for n in d["names"]]
    d["removed"] = [str(f) for f in d["removed"]]
    return d


def sequences(d):
    fa = capi.parse_fasta(bytes(d["fasta"]))
    return dict(fa)


def test_subproblem_hash_names_the_references_files():
    d = case()
    info = bytes(d["info"]).decode().splitlines()
    assert info[0] == "filename\tsequences"
    for line in info[1:]:
        fname, names = line.split("\t")
        assert fname == "sub_%s.gfa" % capi.subproblem_hash_hex(names.split(",")), line
        assert fname == "sub_%s.gfa" % capi.subproblem_hash_hex(list(reversed(names.split(",")))), "the names are sorted before hashing"
        assert fname in d["files"]


def test_read_gfa_round_trip_and_sentinels():
    d = case()
    for f in d["files"]:
        text = bytes(d["file." + f])
        g, names = capi.read_gfa(text)
        assert capi.write_gfa(g, names) == text, f
        n = len(g.label)
        assert (g.src_id, g.snk_id) == (n - 2, n - 1) and g.label[n - 2] == 5 and g.label[n - 1] == 6
        bare, names2 = capi.read_gfa(text, add_sentinels=False)
        assert names2 == names and len(bare.label) == n - 2
        # the paths walk edges of the graph, and every path's first / last node hangs on the sentinels
        for p in range(len(names)):
            nodes = g.path_nodes[int(g.path_off[p]):int(g.path_off[p + 1])]
            assert nodes[0] in g.next_idx[int(g.next_off[g.src_id]):int(g.next_off[g.src_id + 1])]
            assert g.snk_id in g.next_idx[int(g.next_off[nodes[-1]]):int(g.next_off[nodes[-1] + 1])]
            for a, b in zip(nodes[:-1], nodes[1:]):
                assert b in g.next_idx[int(g.next_off[a]):int(g.next_off[a + 1])]
    with pytest.raises(capi.ClError):
        capi.read_gfa(b"S\t1\tACGT\nL\t1\t+\t1\t-\t*\n")   # a reversing edge (src/gfa.cpp:52-54)


@pytest.mark.gpu
def test_gpu_subproblem_files_and_restart_match_the_reference(gpu_ctx, tmp_path):
    d = case()
    seqs = sequences(d)
    tree = msa.tree_of_plan(str(d["newick"][0]), d["names"])
    budget = int(d["budget"][0])
    prefix = str(tmp_path / "sub")
    r = msa.progressive_msa(gpu_ctx, seqs, tree, max_num_match_pairs=budget, subproblems_prefix=prefix)
    assert msa.output_text(r) == bytes(d["full"])
    assert sorted(os.path.basename(f) for f in os.listdir(tmp_path) if f.endswith(".gfa")) == sorted(d["files"])
    for f in d["files"]:
        assert open(os.path.join(tmp_path, f), "rb").read() == bytes(d["file." + f]), f
    want_info = bytes(d["info"]).decode().splitlines()
    got_info = open(prefix + "_info.txt").read().replace(str(tmp_path) + "/", "").splitlines()
    assert got_info[0] == want_info[0] and sorted(got_info[1:]) == sorted(want_info[1:])
    # the interrupted run: the root's and one inner subproblem's files are gone; the others are loaded, not recomputed
    for f in d["removed"]:
        os.remove(os.path.join(tmp_path, f))
    r2 = msa.progressive_msa(gpu_ctx, seqs, tree, max_num_match_pairs=budget, subproblems_prefix=prefix, restart=True)
    assert r2["stats"]["restarted"] == 2 and r2["stats"]["merges"] == 2
    assert msa.output_text(r2) == bytes(d["restart"])
    for f in d["removed"]:   # ... and written again
        assert open(os.path.join(tmp_path, f), "rb").read() == bytes(d["file." + f]), f
    # everything there: nothing to do but load the root
    r3 = msa.progressive_msa(gpu_ctx, seqs, tree, max_num_match_pairs=budget, subproblems_prefix=prefix, restart=True)
    assert r3["stats"]["merges"] == 0 and msa.output_text(r3) == bytes(d["restart"])


@pytest.mark.gpu
def test_gpu_cl_msa_writes_restarts_and_prints_induced_alignments(gpu_ctx, tmp_path):
    """the same through the library's own driver (cl_msa with -S, -R and -A): files, restart output, and one induced pairwise CIGAR per
    pair of sequences, each equal to the compiled reference's where it is built"""
    d = case()
    fasta, newick, budget = bytes(d["fasta"]), str(d["newick"][0]), int(d["budget"][0])
    prefix, aprefix = str(tmp_path / "sub"), str(tmp_path / "ind")
    text, st = gpu_ctx.msa(fasta, newick, max_num_match_pairs=budget, subproblems_prefix=prefix, induced_pairwise_prefix=aprefix)
    assert text == bytes(d["full"]) and st["n_merges"] == 4 and st["n_restarted"] == 0
    for f in d["files"]:
        assert open(os.path.join(tmp_path, f), "rb").read() == bytes(d["file." + f]), f
    want_info = bytes(d["info"]).decode()
    assert open(prefix + "_info.txt").read().replace(str(tmp_path) + "/", "") == want_info   # one context, the reference's order
    root, names = capi.read_gfa(text, add_sentinels=True)
    written = sorted(f for f in os.listdir(tmp_path) if f.startswith("ind_"))
    assert len(written) == len(names) * (len(names) - 1) // 2
    for a in range(len(names)):
        for b in range(a + 1, len(names)):
            got = open(os.path.join(tmp_path, "ind_%s_%s.txt" % (names[a], names[b])), "rb").read()
            assert got == capi.induced_pairwise_cigar(root, a, b) + b"\n"
    for f in d["removed"]:
        os.remove(os.path.join(tmp_path, f))
    text2, st2 = gpu_ctx.msa(fasta, newick, max_num_match_pairs=budget, subproblems_prefix=prefix, restart=True)
    assert text2 == bytes(d["restart"]) and st2["n_restarted"] == 2 and st2["n_merges"] == 2
    text3, st3 = gpu_ctx.msa(fasta, newick, max_num_match_pairs=budget, subproblems_prefix=prefix, restart=True)
    assert text3 == bytes(d["restart"]) and st3["n_restarted"] == 1 and st3["n_merges"] == 0
