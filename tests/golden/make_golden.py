#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref, built from
/root/reference by `make -C oracle ref`).  Runs only in the build container.  Every fixture stores inputs
and the reference's outputs; nothing of the reference's source is stored.

  popoa_random_dags.npz      400 seeded random DAG pairs -> Stitcher::subalign and po_poa<1|2|3> results
  popoa_tie_params.npz       the same kind of batch under tie-heavy unit scoring (match=mismatch=1, open 1/2/3, extend 3/2/1;
                             cf. the reference's own tests, src/test/test_alignment.cpp:711-715) on a 2-letter alphabet
  c2_pair_seed7_intervals.npz  the stitch subproblems the reference extracts for the 2 x 1 Mbp HOR pair (seed 7):
                             (start1,len1,start2,len2) per subproblem, sha256 of the reference's full result,
                             and the full expected alignments of the first 600 subproblems
  msa4_stitch_*.npz          stitch batches of a 4-sequence progressive MSA (graph x graph subproblems with bubbles)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from centrolign_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

SIDE_KEYS = ("node_off", "label", "prev_off", "prev_idx", "next_off", "next_idx", "src_off", "src_idx", "snk_off",
             "snk_idx", "back_translation")


def pack_batch(batch, prefix=""):
    out = {}
    for si, s in enumerate(batch.side):
        for k in SIDE_KEYS:
            out["%sg%d.%s" % (prefix, si + 1, k)] = getattr(s, k)
    out[prefix + "only_deletion_alns"] = batch.only_deletion_alns
    return out


def pack_result(res, prefix):
    return {prefix + "aln_off": res.aln_off, prefix + "pairs": res.pairs, prefix + "score": res.score,
            prefix + "route": res.route, prefix + "num_pw": res.num_pw}


def result_digest(aln_off, pairs):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(aln_off, dtype=np.uint64).tobytes())
    h.update(np.ascontiguousarray(pairs, dtype=np.uint64).tobytes())
    return h.hexdigest()


def host_route_cases():
    """inputs of fixture 9 (also used by the tests to rebuild the batches): name -> (batch, StitchParams)"""
    from tests.helpers import host_route_batches
    seqs = host_route_batches.make_sequences()
    return {"seq1": seqs[0], "seq2": seqs[1], "_batches": host_route_batches.build(seqs[0], seqs[1])}


def chain_weight(ms, chain):
    """sum of ScoreFunction::anchor_weight (score_function.hpp:51-75, CLI parameters) over a chain's anchors"""
    tot = 0.0
    for s in chain[:, 0]:
        c = float(ms.count1[s] * ms.count2[s])
        w0 = int(ms.set_off1[s])
        ln, fl = float(ms.walk_off1[w0 + 1] - ms.walk_off1[w0]), float(ms.full_length[s])
        tot += (ln / fl) * (ln / c ** 0.5 - (ln / 2250.0) ** 2 * 2250.0)
    return tot


def ref_results(batch, params, with_forced=True):
    out = {}
    res, _ = po.ref_stitch_batch(batch, params)
    # the reference does not report route / NumPW / score from subalign; take the oracle's, which the
    # parity tests separately pin to the reference's alignments
    orc = po.oracle_stitch_batch(batch, params)
    assert res.same_as(orc, check_score=False, check_route=False) is None
    res.route, res.num_pw, res.score = orc.route, orc.num_pw, orc.score
    out.update(pack_result(res, "subalign."))
    if with_forced:
        for npw in (1, 2, 3):
            f = np.full(batch.n_problems, npw, np.uint8)
            r, _ = po.ref_stitch_batch(batch, params, force_num_pw=f)
            o = po.oracle_stitch_batch(batch, params, force_num_pw=f)
            r.route = o.route
            out.update(pack_result(r, "po_poa%d." % npw))
    return out


def match_finding_goldens():
    # 10. PathMatchFinder::find_matches (match_finder.hpp:120-212): the seeded inputs of tests/helpers.match_cases() -> the
    #     reference's match sets in full; the leaf-pair merges of the 4 x 30 kbp MSA (graphs in stitch4_30k_merge*.npz) -> per-set
    #     counts/lengths and a digest (the root merge's sets are align4_30k_merge2.npz "ms.*" in full)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests import helpers as H
    from tests.test_extraction import load_stitch_case
    out = {}
    names = []
    for name, g1, g2, mc in H.match_cases():
        r = po.ref_find_matches(g1, g2, max_count=mc)
        names.append(name)
        for k in po.MatchSets._DT:
            out["%s.%s" % (name, k)] = getattr(r, k)
        print(name, "max_count", mc, "->", r.n_sets, "sets")
    for m in (0, 1, 2):
        _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        r = po.ref_find_matches(graphs[0], graphs[1], max_count=3000)
        pre = "merge%d." % m
        for k in ("count1", "count2", "full_length"):
            out[pre + k] = getattr(r, k)
        out[pre + "digest"] = np.array([H.match_sets_digest(r)])
        print("merge", m, "->", r.n_sets, "sets")
    # the BASELINE pair (2 x 1 Mbp, seed 7): digest only
    from centrolign_amd import synth
    seqs = synth.hor_sequences(7, 1000000, 2)
    r = po.ref_find_matches(synth.base_graph_from_sequence(seqs[0]), synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8)), max_count=3000)
    out["c2.n_sets"] = np.array([r.n_sets])
    out["c2.digest"] = np.array([H.match_sets_digest(r)])
    print("c2 ->", r.n_sets, "sets")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "match_finder.npz"), **out)


def fuse_goldens():
    # 11. fuse (fuse.hpp:46-152): the seeded inputs of tests/helpers.fuse_cases() -> the reference's fused graph in full; the three
    #     merges of the 4 x 30 kbp MSA (parents + stitched alignment in stitch4_30k_merge*.npz) -> digest of the fused graph
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests import helpers as H
    from tests.test_extraction import load_stitch_case
    from centrolign_amd import capi
    out = {}
    names = []
    for name, g1, g2, pairs in H.fuse_cases():
        r = po.ref_fuse(g1, g2, pairs)
        names.append(name)
        for k in capi.GRAPH_KEYS:
            out["%s.%s" % (name, k)] = getattr(r, k)
    for m in (0, 1, 2):
        z, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        r = po.ref_fuse(graphs[0], graphs[1], z["stitched"].reshape(-1, 2))
        out["merge%d.digest" % m] = np.array([H.graph_digest(r)])
        out["merge%d.sizes" % m] = np.array([len(r.label), len(r.next_idx), len(r.path_off) - 1])
        print("merge", m, "fused:", out["merge%d.sizes" % m])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "fuse.npz"), **out)


def calibration_goldens():
    # 12. the per-leaf step of Core::calibrate_anchor_scores_and_identify_bonds (src/core.cpp:122-166) with the reference's own
    #     classes: the four leaves of the 4 x 30 kbp MSA (their mean IS the score_scale that run used, align4_30k_merge2.npz)
    #     and seeded leaves of other shapes (tests/helpers.calibration_leaves())
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests import helpers as H
    from tests.test_extraction import load_stitch_case
    out = {}
    msa = []
    for m in (0, 1):
        _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        msa += [po.ref_leaf_intrinsic_scale(g, max_num_match_pairs=40000) for g in graphs]
    out["msa4_30k.scales"] = np.array(msa)
    out["msa4_30k.mean"] = np.array([sum(msa) / len(msa)])   # src/core.cpp:169-173: summed in leaf order
    print("4 x 30 kbp leaves:", msa, "mean", out["msa4_30k.mean"][0])
    names = []
    for name, g, budget in H.calibration_leaves():
        out[name] = np.array([po.ref_leaf_intrinsic_scale(g, max_num_match_pairs=budget)])
        names.append(name)
        print(name, out[name][0])
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "calibration.npz"), **out)


def io_goldens():
    # 13. leaf graphs, explicit CIGAR, GFA: sha256 of the REFERENCE's text for the cases of tests/test_io.io_texts() (one GFA in full),
    #     the reference's leaf graphs of tests/test_io.SEQS in full
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import hashlib
    from tests import helpers as H
    from tests import test_io as T
    from tests.test_extraction import load_stitch_case
    from centrolign_amd import capi
    out = {}
    sha = lambda b: np.array([hashlib.sha256(b).hexdigest()])
    for m in range(3):
        z, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        pairs = z["stitched"].reshape(-1, 2)
        fused = po.ref_fuse(graphs[0], graphs[1], pairs)
        names = ["seq%d" % i for i in range(len(fused.path_off) - 1)]
        out["gfa.merge%d" % m] = sha(po.ref_write_gfa(fused, names))
        out["gfa_raw.merge%d" % m] = sha(po.ref_write_gfa(fused, names, False))
        out["cigar.merge%d" % m] = sha(po.ref_explicit_cigar(graphs[0], graphs[1], pairs))
    for name, g1, g2, pairs in H.fuse_cases()[:10]:
        fused = po.ref_fuse(g1, g2, pairs)
        names = ["p%d" % i for i in range(len(fused.path_off) - 1)]
        out["gfa." + name] = sha(po.ref_write_gfa(fused, names))
        out["cigar." + name] = sha(po.ref_explicit_cigar(g1, g2, pairs))
        if name == "fuse02":
            out["gfa_text.fuse02"] = np.frombuffer(po.ref_write_gfa(fused, ["a", "b", "c", "d", "e", "f"][:len(names)]), np.uint8)
    for i, seq in enumerate(T.SEQS):
        g = po.ref_leaf_graph(seq)
        for k in capi.GRAPH_KEYS:
            out["leaf%d.%s" % (i, k)] = getattr(g, k)
    np.savez_compressed(os.path.join(HERE, "io.npz"), **out)


def msa_goldens():
    # 14. the reference's whole pipeline (Core::execute + write_gfa / explicit_cigar, through ref_msa_dump) on small inputs: the
    #     text a native run must reproduce byte for byte (tests/test_msa.py)
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests import helpers as H
    from centrolign_amd import msa, synth
    out = {}
    for name, n, length, seed, budget in H.msa_cases():
        seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
        names = ["seq%d" % i for i in range(n)]
        with tempfile.TemporaryDirectory() as d:
            fa, nwk, o = os.path.join(d, "in.fa"), os.path.join(d, "t.nwk"), os.path.join(d, "out.txt")
            synth.write_fasta(fa, seqs, names)
            open(nwk, "w").write(msa.newick(msa.balanced_tree(names)) + ";")
            po.ref_msa_dump(fa, newick_path=nwk, out_path=o, max_num_match_pairs=budget)
            text = open(o, "rb").read()
        out[name] = np.frombuffer(text, np.uint8)
        print(name, len(text), "bytes")
    np.savez_compressed(os.path.join(HERE, "msa_text.npz"), **out)


def msa_big_goldens():
    # 15. two larger end-to-end fixtures from the reference's own CLI flow (oracle/_ref/ref_cli, -S for the subproblems):
    #     BASELINE configs[0] — the pairwise 2 x 20 kbp pair, seed 1, default budget (the reference's CPU-runnable case, ~2 min) —
    #     and a 10-sequence MSA over the guide tree of configs[2] at 30 kbp per sequence with a 200 000 pair budget (~7 min), so
    #     that merges of 2+2, 4+1 and 5+5 paths are pinned by the driver-run suite (tests/test_msa.py)
    import hashlib
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from centrolign_amd import synth
    cli = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref_cli")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        seqs = synth.hor_sequences(1, 20000, 2)
        synth.write_fasta(os.path.join(d, "c1.fa"), seqs, ["seq0", "seq1"])
        subprocess.check_call([cli, os.path.join(d, "c1.fa"), "-", "-", os.path.join(d, "c1.out"), "0", "0"])
        out["c1_pair_20k.cigar"] = np.frombuffer(open(os.path.join(d, "c1.out"), "rb").read(), np.uint8)
        names, sq, _ = synth.c3_workload(30000)
        synth.write_fasta(os.path.join(d, "m10.fa"), [sq[n] for n in names], names)
        open(os.path.join(d, "t.nwk"), "w").write(synth.C3_NEWICK + "\n")
        subprocess.check_call([cli, os.path.join(d, "m10.fa"), os.path.join(d, "t.nwk"), os.path.join(d, "sub"), os.path.join(d, "m10.gfa"), "200000", "0"])
        out["msa10_30k.gfa"] = np.frombuffer(open(os.path.join(d, "m10.gfa"), "rb").read(), np.uint8)
        keys, shas = [], []
        with open(os.path.join(d, "sub_info.txt")) as f:
            next(f)
            for ln in f:
                path, leaves = ln.rstrip("\n").split("\t")
                keys.append(leaves)
                shas.append(hashlib.sha256(open(path, "rb").read()).hexdigest())
        out["msa10_30k.sub_leaves"] = np.array(keys)
        out["msa10_30k.sub_sha256"] = np.array(shas)
    np.savez_compressed(os.path.join(HERE, "msa_text_big.npz"), **out)
    print({k: len(v) for k, v in out.items()})


def exhaustive_goldens():
    # 16. exhaustive_chain_dp (anchorer.hpp:1342-1509, the "-g 0" algorithm) by the compiled reference: the CHAINS (not only their
    #     weights) on budgeted subsets of the match sets of the three merges of the 4 x 30 kbp MSA, global and local anchoring
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.test_extraction import load_stitch_case
    out = {}
    for m in range(3):
        z = np.load(os.path.join(HERE, "chain4_30k_merge%d.npz" % m))
        _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        full = po.MatchSets(**{k: z["a.ms." + k] for k in po.MatchSets._DT})
        for seed, budget in ((1, 1500), (2, 3000), (3, 600)):
            sub = po.budget_subset(full, budget, seed=seed)
            for glob in (True, False):
                ex, _ = po.ref_chain("exhaustive", graphs[0], graphs[1], sub, global_anchoring=glob)
                out["m%d.%d.%d.%s" % (m, seed, budget, "g" if glob else "l")] = ex
                print(m, seed, budget, glob, len(ex))
    np.savez_compressed(os.path.join(HERE, "exhaustive_chains.npz"), **out)


PLAN_CASES = [
    # (newick or "" for the in-order tree, sequence names in FASTA order)
    ("((((s0,s1),(s2,s3)),s4),(((s5,s6),(s7,s8)),s9));", ["s%d" % i for i in range(10)]),
    ("", ["a", "b", "c", "d"]),
    ("((a,b),(c,(d,e)),f);", list("abcdef")),                       # a polytomy at the root
    ("((a,b,c,d,e)x,(f,g)y)z;", list("abcdefg")),                   # a five-way polytomy, internal labels
    ("((a,b)x,(c)y);", ["a", "c"]),                                 # leaves of the tree without a sequence, unary nodes
    ("(((((a)))),((b,(c))));", ["c", "a", "b"]),                    # chains of unary nodes; FASTA order differs from the tree's
    ('( "seq one" : 0.5 , ( "s,2" :1e-3, s3:2 ) "in(ner" : 0.1 ) ;', ["seq one", "s,2", "s3"]),   # quotes, distances, blanks
    ("(a:1,(b:2,(c:3,(d:4,(e:5,f:6)))));", list("fedcba")),         # a caterpillar
    ("((a,b),(c,d),(e,f),(g,h));", list("abcdefgh")),               # four cherries under one node
    ("((a,b),((c,d),(e,(f,(g,h)))),i,j);", list("acegij")),         # pruning changes subtree sizes
    ("(a,b,c);", ["a", "b"]),
    ("a;", ["a"]),
    ("((a,b),(c,d))", list("abcd")),                                # no terminating ';'
    ("((a,b),(a,d));", list("abd")),                                # duplicate label
    ("((a,b),(c,d));", list("abce")),                               # a sequence the tree does not have
    ("((a,b)c,d);", list("acd")),                                   # a sequence that is not a leaf
    ("((a,b),(c,d)); x", list("abcd")),                             # text after the ';'
]


def plan_goldens():
    # 17. the order of work: Tree(newick) + Execution (prune / compact / binarize / small_first_postorder) on odd guide trees, as text
    #     (oracle/ref_driver.cpp ref_msa_plan); and parse_fasta on a few inputs
    import ctypes as C
    import json
    lib = po.ref_lib()
    lib.ref_msa_plan.restype = C.c_int
    lib.ref_msa_plan.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint64, C.POINTER(C.c_char_p)]
    out = []
    for nwk, names in PLAN_CASES:
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        txt = C.c_char_p()
        lib.ref_msa_plan(nwk.encode(), arr, len(names), C.byref(txt))
        out.append({"newick": nwk, "names": names, "plan": txt.value.decode()})
        print(repr(nwk), "->", txt.value.decode().replace("\n", " | "))
    json.dump(out, open(os.path.join(HERE, "msa_plans.json"), "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "induced":
        return induced_cigar_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "ifuse":
        return internal_fuse_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "restart":
        return restart_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "cyclize":
        return cyclize_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "cyclize_flow":
        return cyclize_flow_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "plans":
        return plan_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "exhaustive":
        return exhaustive_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "msa_big":
        return msa_big_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "msa":
        return msa_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "io":
        return io_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "calibration":
        return calibration_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "match":
        return match_finding_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "fuse":
        return fuse_goldens()
    sp = capi.default_stitch_params()
    # 1. random DAG pairs, CLI scoring
    b = synth.random_dag_batch(400, seed=20261002, max_n=36)
    d = pack_batch(b)
    d.update(ref_results(b, sp))
    np.savez_compressed(os.path.join(HERE, "popoa_random_dags.npz"), **d)
    # 2. tie-heavy scoring
    tp = capi.default_stitch_params()
    tp.alignment_params.match = 1
    tp.alignment_params.mismatch = 1
    # (all-equal penalties would make the reference's subalign divide by zero, src/stitcher.cpp:43)
    tp.alignment_params.gap_open[:] = [1, 2, 3]
    tp.alignment_params.gap_extend[:] = [3, 2, 1]
    b = synth.random_dag_batch(300, seed=77, max_n=30, alphabet=2)
    d = pack_batch(b)
    d.update(ref_results(b, tp))
    np.savez_compressed(os.path.join(HERE, "popoa_tie_params.npz"), **d)
    # 3. the 2 x 1 Mbp pair: needs the dump made by ref_msa_dump (94 s); reuse it if present
    dump = "/tmp/c2/s7_1000000_2.fa.dump"
    if not os.path.exists(dump):
        os.makedirs("/tmp/c2", exist_ok=True)
        seqs = synth.hor_sequences(7, 1000000, 2)
        synth.write_fasta("/tmp/c2/s7_1000000_2.fa", seqs)
        po.ref_msa_dump("/tmp/c2/s7_1000000_2.fa", None, dump, "/tmp/c2/s7_1000000_2.fa.out")
    dd = po.read_dump(dump)
    batch, aln_off, pairs = po.batch_from_dump(dd, "m0.")
    iv = []
    for s in batch.side:
        no = s.node_off.astype(np.int64)
        bt = s.back_translation.astype(np.int64)
        ln = np.diff(no)
        st = np.where(ln > 0, bt[np.minimum(no[:-1], len(bt) - 1)], 0)
        assert np.array_equal(bt, np.repeat(st, ln) + np.arange(len(bt)) - np.repeat(no[:-1], ln))
        iv += [st, ln]
    head = 600
    np.savez_compressed(os.path.join(HERE, "c2_pair_seed7_intervals.npz"),
                        intervals=np.stack(iv, 1).astype(np.int64), only_del=batch.only_deletion_alns,
                        ref_sha256=np.frombuffer(result_digest(aln_off, pairs).encode(), dtype=np.uint8),
                        n_pairs=np.array([len(pairs)], np.int64), dp_cells=np.array([batch.dp_cells()], np.int64),
                        head_aln_off=aln_off[:head + 1], head_pairs=pairs[:int(aln_off[head])],
                        cigar_sha256=np.frombuffer(hashlib.sha256(dd["output"].tobytes()).hexdigest().encode(), dtype=np.uint8))
    # 4. MSA-derived graph x graph batches (dump made by /tmp/c2/run_msa.py-style call)
    for tag, path in (("msa4_40k", "/tmp/c2/msa_s13_40000_4.fa.dump"),):
        if not os.path.exists(path):
            print("skip", tag, "(no dump at %s)" % path)
            continue
        try:
            dd = po.read_dump(path)
            n_merges = int(dd["n_merges"][0])
        except Exception as ex:  # dump still being written
            print("skip", tag, "(%s)" % ex)
            continue
        for m in range(n_merges):
            batch, aln_off, pairs = po.batch_from_dump(dd, "m%d." % m)
            d = pack_batch(batch)
            d["subalign.aln_off"], d["subalign.pairs"] = aln_off, pairs
            np.savez_compressed(os.path.join(HERE, "%s_merge%d.npz" % (tag, m)), **d)
    # 5. Stitcher::stitch level: parent graphs + partitioned anchor chain -> extracted batch -> stitched alignment
    #    (dump of a 4 x 30 kbp MSA, seed 22, made by ref_msa_dump with max_num_match_pairs=40000)
    path = "/tmp/c2/t_22_30000_4.fa.dump"
    if os.path.exists(path):
        dd = po.read_dump(path)
        for m in range(int(dd["n_merges"][0])):
            pre = "m%d." % m
            batch, aln_off, pairs = po.batch_from_dump(dd, pre)
            d = pack_batch(batch)
            d["subalign.aln_off"], d["subalign.pairs"] = aln_off, pairs
            for side in ("parent1.", "parent2."):
                for k in ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes", "tableau"):
                    d[side + k] = dd[pre + side + k]
            for k in ("seg_off", "walk_off", "walk1", "walk2", "stitched"):
                d[k] = dd[pre + k]
            np.savez_compressed(os.path.join(HERE, "stitch4_30k_merge%d.npz" % m), **d)
        # 6. chaining DP seam: budgeted subsets of the match sets PathMatchFinder returned for those merges -> the chains
        #    the reference's sparse_chain_dp / sparse_affine_chain_dp select (graphs are in stitch4_30k_merge*.npz)
        for m in range(int(dd["n_merges"][0])):
            pre = "m%d." % m
            g1, g2 = po.graphs_from_dump(dd, pre)
            full = po.MatchSets.from_dump(dd, pre)
            out = {}
            for tag, budget, seed, scale in (("a", 20000, m, 0.7), ("b", 6000, 10 + m, 0.25)):
                ms = po.budget_subset(full, budget, seed=seed)
                for k in po.MatchSets._DT:
                    out["%s.ms.%s" % (tag, k)] = getattr(ms, k)
                out[tag + ".scale"] = np.array([scale])
                out[tag + ".chain_sparse"], _ = po.ref_chain("sparse", g1, g2, ms)
                out[tag + ".chain_affine"], _ = po.ref_chain("affine", g1, g2, ms, scale=scale)
                # the same with Anchorer::global_anchoring (the CLI default): sources/sinks = the graph's end nodes
                out[tag + ".chain_sparse_global"], _ = po.ref_chain("sparse", g1, g2, ms, global_anchoring=True)
                out[tag + ".chain_affine_global"], _ = po.ref_chain("affine", g1, g2, ms, scale=scale, global_anchoring=True)
            # optimality: the total anchor weight of the chain exhaustive_chain_dp (anchorer.hpp:1342-1509, the O(M^2) "-g 0"
            # algorithm) finds on two small subsets of "a" — what any gap-free chaining of those matches must reach
            full_a = po.MatchSets(**{k: out["a.ms." + k] for k in po.MatchSets._DT})
            for seed, budget in ((1, 1500), (2, 3000)):
                sub = po.budget_subset(full_a, budget, seed=seed)
                ex, _ = po.ref_chain("exhaustive", g1, g2, sub, global_anchoring=True)
                out["exhaustive.%d.%d" % (seed, budget)] = np.array([chain_weight(sub, ex)])
            np.savez_compressed(os.path.join(HERE, "chain4_30k_merge%d.npz" % m), **out)
        # 6b. Anchorer::anchor_chain seam (budgeted selection + reorder, scale estimate, affine chain, annotation) with
        #     split_matches_at_branchpoints = false; "g" = global anchoring (CLI default), "l" = local, "n" = no
        #     autocalibration, all three without fill-in; "f" / "fl" = with fill-in re-anchoring (global / local)
        for m in range(int(dd["n_merges"][0])):
            pre = "m%d." % m
            g1, g2 = po.graphs_from_dump(dd, pre)
            ms = po.budget_subset(po.MatchSets.from_dump(dd, pre), 60000, seed=20 + m)
            out = {"score_scale": dd[pre + "score_scale"], "max_num_match_pairs": np.array([20000], np.uint64)}
            for k in po.MatchSets._DT:
                out["ms." + k] = getattr(ms, k)
            for tag, glob, auto, fill in (("g", True, True, False), ("l", False, True, False), ("n", True, False, False),
                                          ("f", True, True, True), ("fl", False, True, True)):
                r = po.ref_anchor_chain(g1, g2, ms, max_num_match_pairs=20000, score_scale=float(dd[pre + "score_scale"][0]),
                                        autocalibrate=auto, global_anchoring=glob, fill_in=fill)
                for k, v in r.items():
                    out["%s.%s" % (tag, k)] = np.asarray(v)
            # "sf": the whole default-configured call — split_branching_matches first (anchorer.hpp:971-973; tunables
            # (5, 30, 1, 16) so that the small graphs split), then chaining with fill-in on the split sets
            split = po.ref_split_branching_matches(g1, g2, ms, 5, 30, 1, 16)
            r = po.ref_anchor_chain(g1, g2, split, max_num_match_pairs=20000, score_scale=float(dd[pre + "score_scale"][0]), fill_in=True)
            for k, v in r.items():
                out["sf.%s" % k] = np.asarray(v)
            np.savez_compressed(os.path.join(HERE, "anchor4_30k_merge%d.npz" % m), **out)
        # 6c. Anchorer::split_branching_matches on the match sets of 6b (inputs: anchor4_30k_merge*.npz "ms.*"); "a" = the CLI
        #     tunables with a spread threshold the small graphs reach, "b" = aggressive splitting
        for m in range(int(dd["n_merges"][0])):
            pre = "m%d." % m
            g1, g2 = po.graphs_from_dump(dd, pre)
            ms = po.budget_subset(po.MatchSets.from_dump(dd, pre), 60000, seed=20 + m)
            out = {}
            for tag, prm in (("a", (5, 30, 1, 16)), ("b", (3, 10, 0, 64))):
                r = po.ref_split_branching_matches(g1, g2, ms, *prm)
                out[tag + ".params"] = np.array(prm, np.uint64)
                for k in po.MatchSets._DT:
                    out["%s.%s" % (tag, k)] = getattr(r, k)
            np.savez_compressed(os.path.join(HERE, "split4_30k_merge%d.npz" % m), **out)
        # 6d. Core::align end to end (core.hpp:181-252) on the root merge: ALL the match sets PathMatchFinder returned, the
        #     calibrated score scale, and what the reference made of them — the partitioned + despecified anchor segments
        #     and the stitched alignment (graphs are in stitch4_30k_merge2.npz; max_num_match_pairs was 40000)
        pre = "m2."
        out = {k[len(pre):]: v for k, v in dd.items() if k.startswith(pre) and (k[len(pre):].startswith("ms.") or
               k[len(pre):] in ("score_scale", "seg_off", "walk_off", "walk1", "walk2", "stitched"))}
        np.savez_compressed(os.path.join(HERE, "align4_30k_merge2.npz"), **out)
    else:
        print("skip stitch-level fixtures (no dump at %s)" % path)
    # 8. greedy_partial_alignment (the route of unalignable gaps above max_trivial_size): stretches of two related sequences
    #    (identical, mutated, shifted, with an indel, unrelated) and random DAG pairs, all flagged only_deletion_alns
    rng = np.random.default_rng(5)
    L = 6000
    a = rng.integers(0, 4, L).astype(np.uint8)
    b = a.copy()
    idx = rng.choice(L, 60, replace=False)
    b[idx] = (b[idx] + rng.integers(1, 4, 60)) % 4
    b = np.concatenate([b[:2000], b[2100:4000], rng.integers(0, 4, 150).astype(np.uint8), b[4000:]])
    rows = np.array([(0, 200, 0, 200), (300, 300, 300, 310), (1000, 400, 1000, 380), (1900, 400, 1900, 300), (3000, 500, 2900, 500),
                     (3900, 400, 3800, 550), (5000, 400, 5050, 400), (100, 200, 100, 200), (0, 180, 0, 180), (700, 250, 700, 250),
                     (4500, 1000, 4550, 1000), (50, 300, 3000, 300), (2000, 1, 2000, 400)], np.int64)
    sb = synth.batch_from_intervals(a, b, rows, np.ones(len(rows), np.uint8))
    res, _ = po.ref_stitch_batch(sb)
    out = {"seq1": a, "seq2": b, "rows": rows, "linear.aln_off": res.aln_off, "linear.pairs": res.pairs,
           "dag_cases": np.array([(11, 300, 40), (13, 400, 30)], np.int64)}
    for seed, max_n, cnt in out["dag_cases"]:
        db = synth.random_dag_batch(int(cnt), seed=int(seed), max_n=int(max_n))
        db.only_deletion_alns[:] = 1
        res, _ = po.ref_stitch_batch(db)
        out["dag%d.aln_off" % seed], out["dag%d.pairs" % seed] = res.aln_off, res.pairs
    np.savez_compressed(os.path.join(HERE, "popoa_greedy.npz"), **out)
    # 9. the host routes of Stitcher::do_alignment (deletion-WFA "ad1/ad2", pruned WFA "w", greedy "u", pure deletion) on
    #    stretches of two related sequences and on random DAG batches with shrunken thresholds: whole-batch reference results
    out = host_route_cases()
    for tag, (batch, params) in list(out["_batches"].items()):
        res, _ = po.ref_stitch_batch(batch, params)
        out[tag + ".aln_off"], out[tag + ".pairs"] = res.aln_off, res.pairs
    del out["_batches"]
    np.savez_compressed(os.path.join(HERE, "host_routes.npz"), **out)
    # 7. despecify_indel_breakpoints: random anchor chains -> the reference's kept set and updated gap fields
    sys.path.insert(0, os.path.dirname(HERE))
    from tests.test_despecify import random_case, ref_despecify
    rng = np.random.default_rng(2026)
    out = {}
    n_cases = 60
    for k in range(n_cases):
        n = int(rng.integers(1, 80)) if k < 55 else int(rng.integers(300, 1500))
        case = random_case(rng, n)
        min_len = int(rng.choice([3, 15, 50, 88]))
        prop = float(rng.choice([0.001, 0.05, 0.45, 0.77]))
        res = ref_despecify(*case, min_len, prop)
        pre = "c%d." % k
        for name, arr in zip(("score", "gb", "gsb", "ga", "gsa"), case):
            out[pre + name] = arr
        out[pre + "min_len"], out[pre + "prop"] = np.array([min_len]), np.array([prop])
        for name, arr in zip(("keep", "o_gb", "o_gsb", "o_ga", "o_gsa"), res):
            out[pre + name] = arr
    out["n_cases"] = np.array([n_cases])
    np.savez_compressed(os.path.join(HERE, "despecify.npz"), **out)
    match_finding_goldens()
    fuse_goldens()
    calibration_goldens()
    io_goldens()
    msa_goldens()
    print("golden vectors written to", HERE)



def cyclize_goldens():
    # 16. the tandem-duplication rounds of cyclisation (src/core.cpp:196-296), the parts on the hot path: a leaf's
    #     self-matches, Core::generate_diagonal_mask, Anchorer::anchor_chain with masked matches and the leaf's intrinsic scale as the
    #     overriding scale, Core::update_mask, a second round on the reordered sets, and Stitcher::internal_stitch of the secondary chain
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    out = {}
    names = []
    for k, (length, budget, kw) in enumerate([(6000, 1250000, {}), (9000, 4000, dict(hor_div=0.05)), (3000, 1250000, dict(mono_len=31, hor_n=4))]):
        name = "leaf%d" % k
        names.append(name)
        seq = synth.hor_sequences(60 + k, length, 1, **kw)[0]
        g = g5 = synth.base_graph_from_sequence(seq, (5, 6))
        g7 = synth.base_graph_from_sequence(seq, (7, 8))                  # the second copy under its own sentinels, src/core.cpp:128-133
        ms = po.ref_find_matches(g5, g7, max_count=3000)
        scale = po.ref_leaf_intrinsic_scale(g, max_num_match_pairs=budget)
        mask0 = po.ref_masks(ms, 0)
        d = {"seq": np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), np.uint8), "budget": np.array([budget]), "scale": np.array([scale]), "mask0": mask0}
        d.update({"ms." + f: getattr(ms, f) for f in capi.MatchSets._DT})
        cur, mask = ms, mask0
        for rnd in (1, 2):
            r = po.ref_anchor_chain_masked(g5, g5, cur, mask, override_scale=scale, max_num_match_pairs=budget, score_scale=scale)
            pre = "r%d." % rnd
            for f in ("chain", "score", "walk_off", "walk1", "walk2", "set_order"):
                d[pre + f] = r[f]
            # what Core does next: the sets are now reordered and the mask re-indexed (r["mask"]); update_mask on them (src/core.cpp:288-291)
            cur = cur.reordered(r["set_order"])
            mask = po.ref_masks(cur, 1, chain=r, mask=r["mask"], mask_reciprocal=True)
            d[pre + "mask_after_chain"] = r["mask"]
            d[pre + "mask_after_update"] = mask
            print(name, "round", rnd, "anchors", len(r["chain"]), "mask", len(r["mask"]), "->", len(mask))
            if rnd == 1:
                n = min(len(r["chain"]), 60)
                wo = r["walk_off"][:n + 1]
                d["stitch.pairs"] = po.ref_internal_stitch(g5, wo, r["walk1"][:int(wo[-1])], r["walk2"][:int(wo[-1])])
                d["stitch.n_anchors"] = np.array([n])
                print(name, "internal_stitch of", n, "anchors:", len(d["stitch.pairs"]), "pairs")
        for f, v in d.items():
            out[name + "." + f] = v
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "cyclize_rounds.npz"), **out)


CYCLIZE_FLOW_CASES = [
    # name, seed, length, sequences, duplicated bases, hor_div, carriers of the duplication, min_cyclizing_length, max_num_match_pairs
    ("tri16k", 33, 16000, 3, 5000, 0.10, [0], 3000, 40000),   # (two sequences would print a CIGAR, src/main.cpp:290-296)
    ("three24k", 32, 24000, 3, 8000, 0.05, [0, 1, 2], 6000, 60000),
]


def cyclize_flow_goldens():
    # 18. the CLI's -c flow end to end (src/core.cpp:63-94, 196-297, 594-767) on inputs with real tandem duplications, from the compiled
    #     reference with the cyclisation steps opened up (oracle/ref_driver.cpp: ref_cyclize_dump): per leaf the calibration chain, per
    #     tandem-duplication round the secondary chain, the bonds before and after deduplication and the bond alignments; the MSA graph,
    #     what internal_fuse and simplify_bubbles make of it, the inconsistencies, the polished graph and the GFA text.  The text is
    #     checked against the UNMODIFIED flow (oracle/_ref/ref_cli with the same parameters) before anything is written.
    import subprocess
    import tempfile
    cli = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref_cli")
    out = {"names": np.array([c[0] for c in CYCLIZE_FLOW_CASES])}
    for name, seed, length, n, dup, hor_div, carriers, min_len, budget in CYCLIZE_FLOW_CASES:
        seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=hor_div)
        names = ["s%d" % i for i in range(n)]
        overrides = "i:min_cyclizing_length=%d;i:max_num_match_pairs=%d" % (min_len, budget)
        with tempfile.TemporaryDirectory() as tmp:
            fa = os.path.join(tmp, "in.fa")
            synth.write_fasta(fa, seqs, names)
            po.ref_cyclize_dump(fa, None, os.path.join(tmp, "d.bin"), os.path.join(tmp, "dump.gfa"), overrides)
            subprocess.run([cli, fa, "-", "-", os.path.join(tmp, "cli.gfa"), "0", "0", "0", "b:cyclize_tandem_duplications=1;" + overrides], check=True)
            assert open(os.path.join(tmp, "cli.gfa"), "rb").read() == open(os.path.join(tmp, "dump.gfa"), "rb").read(), name
            d = po.read_dump(os.path.join(tmp, "d.bin"))
        for k, v in d.items():
            out[name + "." + k] = v
        out[name + ".params"] = np.array([seed, length, n, dup, min_len, budget], np.int64)
        out[name + ".hor_div"] = np.array([hor_div])
        out[name + ".carriers"] = np.array(carriers, np.int64)
        print(name, "bonds per leaf:", [int(d["leaf%d.counts" % i][1]) for i in range(n)], "inconsistencies:", len(d["inconsistencies"]) // 2,
              "nodes:", [len(d[k + "label"]) for k in ("msa.", "fused.", "simplified.", "polished.")])
    np.savez_compressed(os.path.join(HERE, "cyclize_flow.npz"), **out)


def restart_goldens():
    # 17. -S / -R of the CLI (src/core.cpp:370-422,1071-1081; src/execution.cpp:222-277) with the compiled reference (oracle/_ref/ref_cli):
    #     a 5-sequence MSA with every finished subproblem written out; then the root's and one inner subproblem's files are removed and
    #     the run restarted — the restarted run continues on read_gfa + add_sentinels graphs, so its output is pinned separately
    import glob
    import shutil
    import subprocess
    import tempfile
    cli = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref_cli")
    seqs = synth.hor_sequences(8, 5000, 5)
    names = ["seq%d" % i for i in range(5)]
    newick = "(((seq0,seq1),seq2),(seq3,seq4));"
    budget = 8000
    tmp = tempfile.mkdtemp()
    try:
        fa, nwk = os.path.join(tmp, "in.fa"), os.path.join(tmp, "tree.nwk")
        synth.write_fasta(fa, seqs)
        with open(nwk, "w") as f:
            f.write(newick)
        prefix = os.path.join(tmp, "sub")
        subprocess.run([cli, fa, nwk, prefix, os.path.join(tmp, "full.gfa"), str(budget), "0"], check=True)
        files = sorted(glob.glob(prefix + "_*.gfa"))
        out = {"newick": np.array([newick]), "budget": np.array([budget]), "names": np.array(names),
               "fasta": np.frombuffer(open(fa, "rb").read(), np.uint8), "full": np.frombuffer(open(os.path.join(tmp, "full.gfa"), "rb").read(), np.uint8),
               "info": np.frombuffer(open(prefix + "_info.txt", "rb").read().replace(tmp.encode() + b"/", b""), np.uint8),
               "files": np.array([os.path.basename(f) for f in files])}
        for f in files:
            out["file." + os.path.basename(f)] = np.frombuffer(open(f, "rb").read(), np.uint8)
        # which file is which: the root's lists all five names in the info file; drop it and the ((seq0,seq1),seq2) one
        info = open(prefix + "_info.txt").read().splitlines()[1:]
        by_names = {line.split("\t")[1]: line.split("\t")[0] for line in info}
        removed = [by_names["seq0,seq1,seq2,seq3,seq4"], by_names["seq0,seq1,seq2"]]
        for f in removed:
            os.remove(f)
        out["removed"] = np.array([os.path.basename(f) for f in removed])
        subprocess.run([cli, fa, nwk, prefix, os.path.join(tmp, "restart.gfa"), str(budget), "0", "1"], check=True)
        out["restart"] = np.frombuffer(open(os.path.join(tmp, "restart.gfa"), "rb").read(), np.uint8)
        print("full == restart:", out["full"].tobytes() == out["restart"].tobytes(), "files", [os.path.basename(f) for f in files])
        np.savez_compressed(os.path.join(HERE, "restart_case.npz"), **out)
    finally:
        shutil.rmtree(tmp)


def internal_fuse_goldens():
    # 18. internal_fuse (fuse.hpp:144-247; Core::apply_bonds, src/core.cpp:631-636): a graph merged with itself along an alignment — the
    #     tandem-duplication alignments of the cyclisation goldens (leaf graphs, Stitcher::internal_stitch output) and random node pairs on
    #     multi-path bubble graphs (transitive merges, groups with several labels, cycles)
    z = np.load(os.path.join(HERE, "cyclize_rounds.npz"))
    out = {}
    names = []

    def put(name, g, pairs):
        fused, trans = po.ref_internal_fuse(g, pairs)
        names.append(name)
        out[name + ".pairs"] = np.asarray(pairs, np.uint64).reshape(-1, 2)
        out[name + ".trans"] = trans
        for k in capi.GRAPH_KEYS:
            out[name + ".g." + k] = getattr(g, k)
            out[name + ".f." + k] = getattr(fused, k)
        out[name + ".g.ids"] = np.array([g.src_id, g.snk_id], np.uint64)
        out[name + ".f.ids"] = np.array([fused.src_id, fused.snk_id], np.uint64)
        print(name, len(g.label), "->", len(fused.label), "nodes")
    for leaf in z["names"]:
        seq = bytes(z[str(leaf) + ".seq"]).decode()
        put(str(leaf), synth.base_graph_from_sequence(seq, (5, 6)), z[str(leaf) + ".stitch.pairs"])
    rng = np.random.default_rng(12)
    for k in range(6):
        anc = "".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(30, 200))))
        g = synth.bubble_graph(anc, int(rng.integers(1, 5)), seed=k)
        n = len(g.label)
        m = int(rng.integers(1, 3 * n))
        pairs = np.stack([rng.integers(0, n, m), rng.integers(0, n, m)], 1).astype(np.uint64)
        pairs[::5, int(k % 2)] = np.uint64(2 ** 64 - 1)   # gaps are skipped
        put("random%d" % k, g, pairs)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "internal_fuse.npz"), **out)


def induced_cigar_goldens():
    # 19. the -A output (src/core.cpp:546-550): explicit_cigar(induced_pairwise_alignment(graph, p1, p2), ...) of the compiled reference for
    #     every ordered path pair of the restart fixture's subproblem graphs (as read_gfa + add_sentinels loads them)
    z = np.load(os.path.join(HERE, "restart_case.npz"))
    out, keys = {}, []
    for f in z["files"]:
        f = str(f)
        text = bytes(z["file." + f])
        g, names = capi.read_gfa(text)
        out["gfa." + f] = np.frombuffer(text, np.uint8)
        for a in range(len(names)):
            for b in range(len(names)):
                if a != b:
                    k = "%s|%d|%d" % (f, a, b)
                    keys.append(k)
                    out["cigar." + k] = np.frombuffer(po.ref_induced_pairwise_cigar(g, a, b), np.uint8)
    out["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(HERE, "induced_cigars.npz"), **out)


if __name__ == "__main__":
    main()
