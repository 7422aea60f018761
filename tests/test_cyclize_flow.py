"""Cyclisation (the CLI's -c) beyond the hot path's own calls: Bonder::identify_bonds / deduplicate_self_bonds, the
per-leaf tandem-duplication rounds (src/core.cpp:196-297), Core::apply_bonds up to the polishing step (internal_fuse + simplify_bubbles,
:594-645).  Expected values: the compiled reference's -c flow with its steps recorded (tests/golden/make_golden.py cyclize_flow ->
cyclize_flow.npz; the recorded flow's GFA was checked against the unmodified CLI flow when the file was made) on inputs with real tandem
duplications."""
import os

import numpy as np
import pytest

from centrolign_amd import capi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
CHAIN_KEYS = ("walk_off", "walk1", "walk2", "score", "gap_after", "gap_score_after")
BOND_KEYS = ("interval_off", "offset1", "offset2", "length")


def cases():
    z = np.load(os.path.join(HERE, "golden", "cyclize_flow.npz"))
    for name in z["names"]:
        name = str(name)
        d = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
        seed, length, n, dup, min_len, budget = [int(x) for x in d["params"]]
        seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=[int(c) for c in d["carriers"]], hor_div=float(d["hor_div"][0]))
        yield name, d, seqs, min_len, budget


def graph_of(d, pre):
    t = d[pre + "tableau"]
    return capi.BaseGraph(*[d[pre + k] for k in capi.GRAPH_KEYS], int(t[0]), int(t[1]))


def bond_alignments(d, n_leaves):
    alns, owner = [], []
    for i in range(n_leaves):
        for b in range(int(d["leaf%d.counts" % i][1])):
            alns.append(d["leaf%d.bond_aln%d" % (i, b)].reshape(-1, 2))
            owner.append(i)
    return alns, owner


def test_identify_bonds_matches_the_reference():
    """host only: every round of every leaf, before and after deduplication; offsets, lengths and — to the last bit — scores"""
    n_rounds = n_bonds = 0
    for name, d, seqs, min_len, budget in cases():
        bp = capi.bond_params(min_length=min_len)
        for i, seq in enumerate(seqs):
            leaf = synth.base_graph_from_sequence(seq, (5, 6))
            opt = {k: d["leaf%d.opt.%s" % (i, k)] for k in CHAIN_KEYS}
            for r in range(int(d["leaf%d.counts" % i][0])):
                sec = {k: d["leaf%d.r%d.sec.%s" % (i, r, k)] for k in CHAIN_KEYS}
                for dedup, key in ((False, "raw_bonds."), (True, "bonds.")):
                    got = capi.identify_bonds(leaf, opt, sec, bp, deduplicate=dedup)
                    pre = "leaf%d.r%d.%s" % (i, r, key)
                    for k in BOND_KEYS:
                        assert np.array_equal(got[k], d[pre + k]), (name, pre, k)
                    assert np.array_equal(got["score"].view(np.uint64), d[pre + "score"].view(np.uint64)), (name, pre)
                    n_bonds += len(got["interval_off"]) - 1
                n_rounds += 1
    assert n_rounds >= 8 and n_bonds >= 10


def test_apply_bonds_and_simplify_bubbles_match_the_reference():
    """host only: the MSA graph and the bond alignments (path positions) -> the graph the polishing step starts from; and simplify_bubbles
    alone on what internal_fuse made"""
    import re
    for name, d, seqs, min_len, budget in cases():
        alns, owner = bond_alignments(d, len(seqs))
        assert alns
        path_names = re.findall(r"^P\t(\S+)", d["output"].tobytes().decode(), re.M)
        path_of = [path_names.index("s%d" % i) for i in owner]
        want = graph_of(d, "simplified.")
        got = capi.apply_bonds(graph_of(d, "msa."), path_of, alns)
        assert capi.graphs_equal(got, want), name
        fused = graph_of(d, "fused.")
        assert capi.graphs_equal(capi.simplify_bubbles(fused), want), name
        assert len(want.label) < len(fused.label) < len(d["msa.label"])          # the bonds merged something, the bubbles went, cycles exist
        # simplifying again changes nothing
        assert capi.graphs_equal(capi.simplify_bubbles(want), want), name


def test_simplify_bubbles_small_cases():
    """hand-made graphs: identical alleles merge onto the first, different ones stay, a bubble at a sentinel is left alone"""
    def g(labels, edges, paths, src, snk):
        n = len(labels)
        nxt, prv = [[] for _ in range(n)], [[] for _ in range(n)]
        for a, b in edges:
            nxt[a].append(b); prv[b].append(a)
        off = lambda ls: np.cumsum([0] + [len(x) for x in ls]).astype(np.uint64)
        flat = lambda ls: np.array([v for x in ls for v in x], np.uint32)
        return capi.BaseGraph(np.array(labels, np.uint8), off(nxt), flat(nxt), off(prv), flat(prv), off(paths), flat(paths), src, snk)
    # 0:src 1:A 2:C 3:C 4:G 5:T 6:snk   bubble 1 -> {2, 3, 4} -> 5, alleles C, C, G
    graph = g([5, 0, 1, 1, 2, 3, 6], [(0, 1), (1, 2), (1, 3), (1, 4), (2, 5), (3, 5), (4, 5), (5, 6)], [[1, 2, 5], [1, 3, 5], [1, 4, 5]], 0, 6)
    out = capi.simplify_bubbles(graph)
    assert len(out.label) == 6 and out.label.tolist() == [5, 0, 1, 2, 3, 6]
    assert out.path_nodes.tolist() == [1, 2, 4, 1, 2, 4, 1, 3, 4]
    # the same bubble hanging off the source sentinel stays
    graph = g([5, 1, 1, 3, 6], [(0, 1), (0, 2), (1, 3), (2, 3), (3, 4)], [[1, 3], [2, 3]], 0, 4)
    assert capi.graphs_equal(capi.simplify_bubbles(graph), graph)
    # an allele with a second way in is not a plain run: nothing happens
    graph = g([5, 0, 1, 1, 2, 3, 6], [(0, 1), (1, 2), (1, 3), (2, 3), (2, 5), (3, 5), (5, 6), (1, 4), (4, 5)], [[1, 2, 5], [1, 3, 5], [1, 2, 3, 5], [1, 4, 5]], 0, 6)
    assert capi.graphs_equal(capi.simplify_bubbles(graph), graph)


@pytest.mark.gpu
def test_leaf_calibration_and_bond_rounds_match_the_reference(gpu_ctx):
    """device: per leaf cl_leaf_calibrate (self matches + main-diagonal chain + intrinsic scale), then the tandem-duplication rounds with the
    mean scale — masked anchor chains, bonds, internal_stitch — give the reference's bond alignments, in its order, in path positions"""
    for name, d, seqs, min_len, budget in cases():
        leaves = [capi.leaf_graph(s) for s in seqs]
        memos, scales = [], []
        try:
            for leaf in leaves:
                sc, h = gpu_ctx.leaf_calibrate(leaf, max_num_match_pairs=budget)
                scales.append(sc); memos.append(h)
            assert np.array_equal(np.array(scales).view(np.uint64), d["intrinsic_scales"].view(np.uint64)), name
            mean = sum(scales) / len(scales)
            assert mean == float(d["score_scale"][0])
            bp = capi.bond_params(min_length=min_len)
            for i, leaf in enumerate(leaves):
                got = gpu_ctx.leaf_bond_alignments(leaf, memos[i], mean, max_num_match_pairs=budget, bonds=bp)
                assert len(got) == int(d["leaf%d.counts" % i][1]), (name, i)
                for b, aln in enumerate(got):
                    assert np.array_equal(aln, d["leaf%d.bond_aln%d" % (i, b)].reshape(-1, 2)), (name, i, b)
        finally:
            for h in memos:
                gpu_ctx.free_leaf_calibration(h)


def test_inconsistencies_match_the_reference():
    """host only: InconsistencyIdentifier::identify_inconsistencies on the cyclised graphs — snarl tree of a cyclic graph, tight cycles,
    indels placed inconsistently across a bond, merging along chains, padding: the same regions in the same order"""
    n = 0
    for name, d, seqs, min_len, budget in cases():
        got = capi.identify_inconsistencies(graph_of(d, "simplified."))
        assert np.array_equal(got, d["inconsistencies"].reshape(-1, 2)), name
        n += len(got)
    assert n >= 20


@pytest.mark.gpu
def test_polishing_matches_the_reference(gpu_ctx):
    """device: Core::polish_cyclized_graph — every region realigned through the hot path with induced matches — gives the reference's polished
    graph, and its GFA is the text the reference's -c run printed"""
    import re
    for name, d, seqs, min_len, budget in cases():
        text = d["output"].tobytes()
        path_names = re.findall(r"^P\t(\S+)", text.decode(), re.M)
        seq_names = ["s%d" % i for i in range(len(seqs))]
        got, n_regions = gpu_ctx.polish_cyclized_graph(graph_of(d, "simplified."), path_names, seq_names, float(d["score_scale"][0]), max_num_match_pairs=budget)
        assert n_regions == len(d["inconsistencies"]) // 2
        assert capi.graphs_equal(got, graph_of(d, "polished.")), name
        assert capi.write_gfa(got, path_names) == text, name


@pytest.mark.gpu
def test_cl_msa_with_cyclisation_prints_the_reference_gfa(gpu_ctx, tmp_path):
    """device: the CLI's -c flow inside the library — cl_msa(cyclize) from FASTA text: calibration with bond search, the MSA, apply_bonds,
    polishing — byte for byte the GFA of the compiled reference (oracle/_ref/ref_cli -c, checked when the golden was made); with worker
    contexts; and a -S run followed by a -R restart that reads the bond alignments back from PREFIX_bonds.txt"""
    for name, d, seqs, min_len, budget in cases():
        want = d["output"].tobytes()
        fasta = "".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs))
        got, st = gpu_ctx.msa(fasta, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len)
        assert got == want, name
        assert st["n_bonds"] == sum(int(d["leaf%d.counts" % i][1]) for i in range(len(seqs))) and st["n_polished_regions"] == len(d["inconsistencies"]) // 2
        got4, _ = gpu_ctx.msa(fasta, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len, workers=3)
        assert got4 == want, name
        prefix = str(tmp_path / (name + "_sub"))
        got_s, _ = gpu_ctx.msa(fasta, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len, subproblems_prefix=prefix)
        assert got_s == want and os.path.exists(prefix + "_bonds.txt")
        lines = open(prefix + "_bonds.txt").read().splitlines()
        assert sum(1 for ln in lines if ln.startswith("#")) == st["n_bonds"]
        got_r, st_r = gpu_ctx.msa(fasta, max_num_match_pairs=budget, cyclize=True, min_cyclizing_length=min_len, subproblems_prefix=prefix, restart=True)
        assert got_r == want and st_r["n_restarted"] >= 1 and st_r["n_bonds"] == st["n_bonds"]
        break   # (the second golden is covered step by step above; one end-to-end flow keeps the suite short)


WIDE = {   # tests/golden/make_cyclize_wide.py CASES: sequences, seed, length, duplicated bases, carriers, name prefix, merges, bonds, regions
    "cyclize_16x12k": (16, 41, 12000, 4000, [0, 3, 5, 8, 9, 13], "c", 15, 74, 455),
    "cyclize_50x8k": (50, 43, 8000, 3000, [1, 4, 7, 12, 18, 23, 29, 31, 36, 40, 44, 48], "d", 49, 133, 550),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(WIDE))
def test_cl_msa_with_cyclisation_on_many_sequences(gpu_ctx, case):
    """BASELINE configs[4] scaled down: the -c flow over a balanced guide tree on 16 sequences (merges up to 8 + 8 paths, 64 chain combinations)
    and on 50 sequences (root merge 25 + 25 paths = 625 combinations: beyond the walk kernel, on the per-block kernels; 133 bonds, 550
    polished regions realigned on the worker contexts) prints the GFA of the unmodified reference (tests/golden/make_cyclize_wide.py;
    1.8 and 5.1 minutes there), byte for byte"""
    import gzip
    import hashlib
    import json
    from centrolign_amd import msa
    n, seed, length, dup, carriers, prefix, n_merges, n_bonds, n_regions = WIDE[case]
    gold = json.load(open(os.path.join(HERE, "golden", case + ".json")))
    seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=0.08)
    names = ["%s%02d" % (prefix, i) for i in range(n)]
    assert hashlib.sha256("".join(seqs).encode()).hexdigest() == gold["input_sha256"]
    want = gzip.open(os.path.join(HERE, "golden", case + ".gfa.gz")).read()
    assert hashlib.sha256(want).hexdigest() == gold["gfa"]["sha256"]
    fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
    got, st = gpu_ctx.msa(fasta, newick=msa.newick(msa.balanced_tree(names)) + ";", max_num_match_pairs=gold["max_num_match_pairs"], cyclize=True,
                          min_cyclizing_length=gold["min_cyclizing_length"], workers=4)
    assert got == want
    assert st["n_merges"] == n_merges and st["n_bonds"] == n_bonds and st["n_polished_regions"] == n_regions


def test_identify_bonds_where_the_reference_reads_past_a_vector():
    """host only.  Bonder::trim_partition_ends takes off intervening_segments[interval.second] when it trims an interval's end (src/bonder.cpp:753-757): one element PAST
    that vector when the interval ends at the last shared segment, i.e. whatever the heap holds there (the unmodified reference's -c text changes with MALLOC_PERTURB_ and,
    on some inputs, from run to run: profiles/r06_reference_undefined_read.json; an ASan build reports the read: profiles/r06_reference_asan_report.txt).  cl_identify_bonds
    counts that element as zeros and says so in cl_fallback_counters.  Two recorded leaf rounds of scripts/fuzz_msa.py that stand at that spot (the reference's chains in,
    its bonds out): in one the reference's run happened to read nothing that matters and the bonds are its bonds; in the other it stopped trimming early — every interval
    is the reference's except that the trimmed one is a PREFIX of the reference's"""
    z = np.load(os.path.join(HERE, "golden", "bonds_past_the_end.npz"))
    for tag in ("agrees", "differs"):
        leaf = synth.base_graph_from_sequence(z[tag + ".sequence"].tobytes().decode(), (5, 6))
        opt = {k: z["%s.opt.%s" % (tag, k)] for k in CHAIN_KEYS}
        sec = {k: z["%s.sec.%s" % (tag, k)] for k in CHAIN_KEYS}
        capi.fallback_counters(reset=True)
        got = capi.identify_bonds(leaf, opt, sec, capi.bond_params(min_length=int(z[tag + ".min_length"][0])), deduplicate=False)
        assert capi.fallback_counters()["bond_trims_past_the_end"] >= 1, tag
        want = {k: z["%s.raw_bonds.%s" % (tag, k)] for k in BOND_KEYS + ("score",)}
        same = all(np.array_equal(got[k], want[k]) for k in BOND_KEYS) and np.array_equal(got["score"].view(np.uint64), want["score"].view(np.uint64))
        assert same == (tag == "agrees"), tag
        if not same:
            assert len(got["interval_off"]) == len(want["interval_off"])
            shorter = 0
            for iv in range(len(want["interval_off"]) - 1):
                a, b = int(want["interval_off"][iv]), int(want["interval_off"][iv + 1])
                ga, gb = int(got["interval_off"][iv]), int(got["interval_off"][iv + 1])
                assert gb - ga <= b - a
                shorter += (gb - ga) < (b - a)
                for k in ("offset1", "offset2"):
                    assert np.array_equal(got[k][ga:gb], want[k][a:a + gb - ga]), (iv, k)
                assert np.array_equal(got["length"][ga:gb - 1], want["length"][a:a + gb - ga - 1]), iv     # (the cut may fall inside the last bond)
            assert shorter == 1


@pytest.mark.gpu
def test_polishing_stops_where_the_reference_reads_past_the_last_path(gpu_ctx):
    """device.  InducedMatchFinderComponentView::find_matches lets a hit that starts ONE step behind a realigned stretch through (upper bound (path_end + 1, 0),
    include/centrolign/induced_match_finder.hpp:190) and reads it from the step behind the subpath; behind the last path of a subproblem graph there is nothing to read.
    The unmodified reference ends in a segmentation fault on this input (two sequences, one with a 5-kbp tandem duplication; scripts/fuzz_msa.py seed 3); the library
    returns an error that says where it stands — and keeps working afterwards"""
    seqs = synth.tandem_dup_sequences(439391627, 14000, 2, 5000, carriers=[1], seq_div=0.01, hor_div=0.02)
    fasta = "".join(">q%02d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    with pytest.raises(capi.ClError) as e:
        gpu_ctx.msa(fasta, newick="(q00,q01);", max_num_match_pairs=8000, cyclize=True, min_cyclizing_length=2500)
    assert "read past the end of the subproblem's paths" in str(e.value)
    text, st = gpu_ctx.msa(fasta, newick="(q00,q01);", max_num_match_pairs=8000)
    assert text.endswith(b"\n") and st["n_merges"] == 1


@pytest.mark.gpu
def test_stray_hits_can_be_left_out_instead(tmp_path):
    """device.  CL_POLISH_SKIP_STRAY_HITS=1 (read once: a child process): the hit that begins one step behind a realigned stretch is none of the stretch's — what the bound
    of induced_match_finder.hpp:190 evidently means — and an input the reference dies on goes through: five sequences, 16 bonds, 52 polished regions, and every path of the
    GFA spells its sequence"""
    import subprocess
    import sys
    code = (
        "import re, sys\n"
        "sys.path.insert(0, %r)\n"
        "from centrolign_amd import capi, synth\n"
        "seqs = synth.tandem_dup_sequences(518117772, 9000, 5, 5000, carriers=[0, 2], seq_div=0.03, hor_div=0.02)\n"
        "names = ['q%%02d' %% i for i in range(5)]\n"
        "fasta = ''.join('>%%s\\n%%s\\n' %% (nm, s) for nm, s in zip(names, seqs))\n"
        "text, st = capi.Context(0).msa(fasta, newick='((q00,q01),(q02,(q03,q04)));', max_num_match_pairs=8000, cyclize=True, min_cyclizing_length=1000)\n"
        "seg = {m.group(1): m.group(2) for m in re.finditer(r'^S\\t(\\S+)\\t(\\S+)', text.decode(), re.M)}\n"
        "paths = {m.group(1): ''.join(seg[x[:-1]] for x in m.group(2).split(',')) for m in re.finditer(r'^P\\t(\\S+)\\t(\\S+)', text.decode(), re.M)}\n"
        "assert paths == dict(zip(names, seqs))\n"
        "print('RESULT', st['n_bonds'], st['n_polished_regions'])\n" % os.path.dirname(HERE))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CL_POLISH_SKIP_STRAY_HITS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RESULT 16 52" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
