"""End to end without the reference in the loop: sequences + guide tree -> leaf graphs, calibration, one cl_merge per tree node, output
text — byte-identical to what the compiled reference's whole pipeline (Core::execute + write_gfa / explicit_cigar) printed for the same
FASTA + Newick (tests/golden/msa_text.npz, made by ref_msa_dump)."""
import os

import numpy as np
import pytest

from centrolign_amd import msa, synth
from tests import helpers as H

Z = np.load(os.path.join(H.GOLDEN, "msa_text.npz"))


def test_tree_helpers():
    t = msa.balanced_tree(["a", "b", "c", "d", "e"])
    assert msa.newick(t) == "((a,b),(c,(d,e)))" and msa.leaves_of(t) == ["a", "b", "c", "d", "e"]


class _StubCtx:
    """host-only stand-in for a device context: the scale is a function of the leaf, merge = cl_fuse along the first bases"""
    def leaf_intrinsic_scale(self, g, **kw):
        return float(int(g.label.astype(np.int64).sum()) % 9973) / 7.0

    def merge(self, g1, g2, score_scale=1.0, **kw):
        from centrolign_amd import capi
        k = min(5, len(g1.label) - 2, len(g2.label) - 2)
        p1 = g1.path_nodes[int(g1.path_off[0]):int(g1.path_off[0]) + k].astype(np.uint64)
        p2 = g2.path_nodes[int(g2.path_off[0]):int(g2.path_off[0]) + k].astype(np.uint64)
        pairs = np.stack([p1, p2], 1)
        return dict(fused=capi.fuse(g1, g2, pairs), alignment=pairs, match_ms=0.0, align_ms=0.0, fuse_ms=0.0, n_match_sets=0)


def test_worker_threads_schedule_the_same_msa():
    """the wave scheduler with several worker contexts (host-only stub contexts): same root graph, paths and scale as one worker"""
    from centrolign_amd import capi
    rng = np.random.default_rng(2)
    names = ["s%d" % i for i in range(9)]
    seqs = {nm: "".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(20, 80)))) for nm in names}
    for tree in (msa.balanced_tree(names), ((("s0", "s1"), "s2"), ("s3", ("s4", ("s5", ("s6", ("s7", "s8"))))))):
        want = msa.progressive_msa(_StubCtx(), seqs, tree)
        for workers in (2, 5):
            got = msa.progressive_msa(_StubCtx(), seqs, tree, workers=workers, make_context=_StubCtx)
            assert capi.graphs_equal(got["root"], want["root"]) and got["paths"] == want["paths"] and got["scales"] == want["scales"]
            assert got["stats"]["merges"] == len(names) - 1
        assert msa.output_text(want).startswith(b"H\tVN:Z:1.0\n")


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 3])
@pytest.mark.parametrize("case", H.msa_cases(), ids=lambda c: c[0])
def test_gpu_native_run_prints_the_references_text(gpu_ctx, case, workers):
    """workers = 3: leaf calibrations and sibling merges side by side on one device, one cl_context per worker thread"""
    name, n, length, seed, budget = case
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    names = ["seq%d" % i for i in range(n)]
    r = msa.progressive_msa(gpu_ctx, dict(zip(names, seqs)), msa.balanced_tree(names), max_num_match_pairs=budget, workers=workers)
    want = bytes(Z[name])
    got = msa.output_text(r)
    assert got == (want.rstrip(b"\n") if n == 2 else want)   # the CIGAR line ends in a newline in the dump
    assert r["stats"]["merges"] == n - 1 and len(r["root"].path_off) - 1 == n


ZB = np.load(os.path.join(H.GOLDEN, "msa_text_big.npz"))


@pytest.mark.gpu
def test_gpu_baseline_configs0_pairwise_20k(gpu_ctx):
    """BASELINE configs[0]: the pairwise 2 x 20 kbp HOR pair (seed 1, default budget: 1.14 M match pairs), explicit CIGAR byte for byte
    as the reference's CLI flow printed it (tests/golden/make_golden.py msa_big)"""
    seqs = synth.hor_sequences(1, 20000, 2)
    names = ["seq0", "seq1"]
    r = msa.progressive_msa(gpu_ctx, dict(zip(names, seqs)), msa.balanced_tree(names), workers=2)
    assert msa.output_text(r) == bytes(ZB["c1_pair_20k.cigar"]).rstrip(b"\n")


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 4])
def test_gpu_ten_sequence_msa_30k(gpu_ctx, workers):
    """ten sequences over the guide tree of BASELINE configs[2] (30 kbp each, 200 000 pair budget): merges of 1+1, 2+2, 4+1 and 5+5
    paths; the root GFA byte for byte and every subproblem GFA (-S of the reference) by digest"""
    import hashlib
    from centrolign_amd import capi
    names, seqs, tree = synth.c3_workload(30000)
    r = msa.progressive_msa(gpu_ctx, seqs, tree, max_num_match_pairs=200000, workers=workers, keep_merges=True)
    assert msa.output_text(r) == bytes(ZB["msa10_30k.gfa"])
    want = dict(zip(ZB["msa10_30k.sub_leaves"].tolist(), ZB["msa10_30k.sub_sha256"].tolist()))
    got = {",".join(sorted(m["paths"])): hashlib.sha256(capi.write_gfa(m["fused"], m["paths"])).hexdigest() for m in r["stats"]["kept"]}
    assert got == want and len(got) == 9


@pytest.mark.gpu
@pytest.mark.parametrize("case", H.msa_cases()[:4], ids=lambda c: c[0])
def test_gpu_cl_msa_is_the_cli_flow(gpu_ctx, case):
    """cl_msa: FASTA text + Newick text in, the reference's output text out — parse_fasta, Tree / Execution order, calibration, merges
    and writer all inside the library, no Python driver"""
    name, n, length, seed, budget = case
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    names = ["seq%d" % i for i in range(n)]
    fasta = "".join(">%s\n%s\n" % (nm, "\n".join(s[i:i + 80] for i in range(0, len(s), 80))) for nm, s in zip(names, seqs))
    text, st = gpu_ctx.msa(fasta, msa.newick(msa.balanced_tree(names)) + ";", max_num_match_pairs=budget)
    want = bytes(Z[name])
    assert text == want and st["n_merges"] == n - 1   # (two sequences: the CIGAR and the line end main() prints, src/main.cpp:295)
    # three worker contexts inside the library: calibrations and independent merges side by side, the same text
    text_w, st_w = gpu_ctx.msa(fasta, msa.newick(msa.balanced_tree(names)) + ";", max_num_match_pairs=budget, workers=3)
    assert text_w == text and st_w["n_merges"] == n - 1 and st_w["score_scale"] == st["score_scale"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", H.msa_cases() + [("msa4_30k", 4, 30000, 11, 60000)], ids=lambda c: c[0])
def test_gpu_cl_msa_with_the_sparse_chaining_algorithm(gpu_ctx, case):
    """the CLI's -g 1 (src/main.cpp:129): every merge chains with Anchorer::Sparse over ChainMerge structures (include/centrolign/core.hpp:350-357);
    text of the compiled reference's Core::execute with that setting (tests/golden/make_chainmerge_golden.py)"""
    name, n, length, seed, budget = case
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    names = ["seq%d" % i for i in range(n)]
    fasta = "".join(">%s\n%s\n" % (nm, s) for nm, s in zip(names, seqs))
    want = bytes(np.load(os.path.join(H.GOLDEN, "chainmerge4_30k_g1.npz"))["cli." + name])
    text, st = gpu_ctx.msa(fasta, msa.newick(msa.balanced_tree(names)) + ";", max_num_match_pairs=budget, chaining_algorithm=1)
    assert text == want and st["n_merges"] == n - 1
    if name == "pair_60k":
        assert want != bytes(Z[name])     # (another text than the default algorithm's; msa4_30k too: checked where the golden is made)


@pytest.mark.gpu
def test_gpu_context_memory_statistics():
    """cl_context_memory: what a context's calls hold, held at most and keep cached, beside hipMemGetInfo's numbers"""
    from centrolign_amd import capi
    ctx = capi.Context(0)
    try:
        m0 = ctx.memory_stats()
        assert m0["device_total_bytes"] > 200e9 and m0["device_free_bytes"] <= m0["device_total_bytes"] and m0["peak_bytes"] >= m0["live_bytes"]
        seqs = synth.hor_sequences(5, 60000, 2)
        g = [capi.leaf_graph(x) for x in seqs]
        ctx.merge(g[0], g[1], score_scale=1.0, max_num_match_pairs=200000)
        m1 = ctx.memory_stats()
        assert m1["peak_bytes"] > 10 * 2 ** 20 and m1["peak_bytes"] >= m1["live_bytes"] and m1["cached_bytes"] > 0   # the merge's blocks went back to the context's cache
        m2 = ctx.memory_stats(reset_peak=True)
        assert m2["peak_bytes"] == m1["peak_bytes"] and ctx.memory_stats()["peak_bytes"] == m2["live_bytes"]
        # plans give back everything they took (round-4 advisor: every plan kept its block of launch clocks): live bytes
        # after further merges and one-shot stitch batches equal live bytes after the first
        base = ctx.memory_stats()["live_bytes"]
        batch = synth.random_dag_batch(32, seed=5, max_n=40)
        for _ in range(3):
            ctx.merge(g[0], g[1], score_scale=1.0, max_num_match_pairs=200000)
            ctx.stitch_batch_align(batch)
        assert ctx.memory_stats()["live_bytes"] == base
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_cl_msa_ten_sequences(gpu_ctx):
    names, seqs, _ = synth.c3_workload(30000)
    fasta = "".join(">%s some description\n%s\n" % (nm, seqs[nm]) for nm in names)
    text, st = gpu_ctx.msa(fasta, synth.C3_NEWICK, max_num_match_pairs=200000)
    assert text == bytes(ZB["msa10_30k.gfa"])
    text4, st4 = gpu_ctx.msa(fasta, synth.C3_NEWICK, max_num_match_pairs=200000, workers=4)
    assert text4 == text and st4["n_merges"] == 9


WIDE = {"wide_merge_24x7k": (24, 91, 7000, "q"), "wide_merge_50x5k": (50, 92, 5000, "r")}   # tests/golden/make_wide_merge.py CASES


@pytest.mark.gpu
def test_gpu_worker_contexts_on_a_list_of_devices(gpu_ctx):
    """one process, several devices: the worker contexts of cl_msa / progressive_msa are spread over a list of device ordinals
    (cl_msa_params.devices; worker w on devices[w % n]).  A box here has one GPU, so the list names it several times — the code path a
    multi-GPU node takes (contexts created per ordinal, hipSetDevice at every entry point), with the reference's ten-sequence GFA"""
    names, seqs, tree = synth.c3_workload(30000)
    want = bytes(ZB["msa10_30k.gfa"])
    r = msa.progressive_msa(gpu_ctx, seqs, tree, max_num_match_pairs=200000, workers=4, devices=[0, 0, 0])
    assert msa.output_text(r) == want
    fasta = "".join(">%s\n%s\n" % (nm, seqs[nm]) for nm in names)
    got, st = gpu_ctx.msa(fasta, newick=synth.C3_NEWICK, max_num_match_pairs=200000, workers=3, devices=[0, 0])
    assert got == want and st["n_merges"] == 9


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(WIDE))
def test_gpu_wide_merges(gpu_ctx, case):
    """sequences over a balanced tree whose root merge pairs 12 + 12 paths = 144 chain combinations (the walk kernel with its reduction exchange,
    cl_chain_api.cpp) and 25 + 25 paths = 625 (beyond the walk kernel's 256: the per-block kernels) — BASELINE configs[4]'s width at small
    length.  The GFA is the one the unmodified reference printed (tests/golden/make_wide_merge.py: 26 and 34 CPU-minutes there), byte for
    byte; the narrower case also with one worker"""
    import gzip
    import hashlib
    import json
    n, seed, length, prefix = WIDE[case]
    gold = json.load(open(os.path.join(H.GOLDEN, case + ".json")))
    names, seqs = ["%s%02d" % (prefix, i) for i in range(n)], synth.hor_sequences(seed, length, n, indel_hor=1)
    assert hashlib.sha256("".join(seqs).encode()).hexdigest() == gold["input_sha256"]
    fasta = "".join(">%s\n%s\n" % (nm, sq) for nm, sq in zip(names, seqs))
    want = gzip.open(os.path.join(H.GOLDEN, case + ".gfa.gz")).read()
    assert hashlib.sha256(want).hexdigest() == gold["gfa"]["sha256"] and len(want) == gold["gfa"]["bytes"]
    for workers in ((1, 4) if n < 30 else (4,)):
        got, st = gpu_ctx.msa(fasta, newick=gold["newick"], workers=workers)
        assert got == want, workers
        assert st["n_merges"] == n - 1


@pytest.mark.gpu
def test_gpu_cl_msa_with_the_calibration_skipped(gpu_ctx):
    """the CLI's developer switch skip_calibration (src/core.cpp:66): the run keeps ScoreFunction::score_scale's start value 0.303092 (include/centrolign/score_function.hpp:39)
    — cl_msa_params_default's — and prints another text than the calibrated run; both byte for byte the compiled reference's (tests/golden/make_skip_calibration.py; the
    input is one scripts/fuzz_msa.py found: before round 6's end the library kept 1.0)"""
    from tests.golden.make_skip_calibration import CASE
    z = np.load(os.path.join(H.GOLDEN, "msa_skip_calibration.npz"))
    seqs = synth.hor_sequences(CASE["seed"], CASE["length"], CASE["n"], seq_div=CASE["seq_div"], hor_div=CASE["hor_div"])
    fasta = "".join(">q%02d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    text, st = gpu_ctx.msa(fasta, newick=CASE["newick"], max_num_match_pairs=CASE["budget"], skip_calibration=True)
    assert text == z["skipped"].tobytes() and st["score_scale"] == 0.303092
    text, st = gpu_ctx.msa(fasta, newick=CASE["newick"], max_num_match_pairs=CASE["budget"], workers=3)
    assert text == z["calibrated"].tobytes()
