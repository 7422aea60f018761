"""GPU suite (-m gpu): the HIP stitch path, called through the C ABI, must be bit-identical to the oracle and to
the golden vectors of the compiled reference: same int32 scores, same AlignedPair sequences."""
import os

import numpy as np
import pytest

from centrolign_amd import capi, synth
from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_golden_random_dags(gpu_ctx):
    z = np.load(os.path.join(H.GOLDEN, "popoa_random_dags.npz"))
    b = H.load_batch(z)
    assert gpu_ctx.stitch_batch_align(b).same_as(H.load_result(z, "subalign.")) is None
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(H.load_result(z, "po_poa%d." % npw), check_route=False) is None


def test_golden_tie_params(gpu_ctx):
    z = np.load(os.path.join(H.GOLDEN, "popoa_tie_params.npz"))
    b = H.load_batch(z)
    tp = H.tie_params()
    assert gpu_ctx.stitch_batch_align(b, tp).same_as(H.load_result(z, "subalign.")) is None
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, tp.alignment_params)
        assert got.same_as(H.load_result(z, "po_poa%d." % npw), check_route=False) is None


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("msa4_")))
def test_golden_msa_batches(gpu_ctx, name):
    z = np.load(os.path.join(H.GOLDEN, name))
    b = H.load_batch(z)
    got = gpu_ctx.stitch_batch_align(b)
    assert got.same_as(H.load_result(z, "subalign."), check_score=False, check_route=False) is None


def test_c2_pair_full_size(gpu_ctx):
    """BASELINE configs[1] at full size: all 13 245 subproblems / 42.4 M cells, digest of the reference's result"""
    b, z = H.c2_batch()
    plan = gpu_ctx.plan(b)
    st = plan.stats()
    assert st["dp_cells"] == 42416142 and st["n_po_poa"] == 12304 and st["n_linear"] == 12304
    plan.execute()
    plan.sync()
    r = plan.collect()
    assert H.result_digest(r.aln_off, r.pairs) == bytes(z["ref_sha256"]).decode()
    # idempotence: a second pass over the same resident batch gives the same bytes
    plan.execute()
    plan.sync()
    r2 = plan.collect()
    assert r2.same_as(r) is None
    plan.destroy()


def test_known_answers(gpu_ctx):
    for g1, g2, npw, params, expected in H.known_answer_cases():
        b = H.batch_from_graphs([(g1, g2)])
        r = gpu_ctx.po_poa_batch(b, np.array([npw], np.uint8), params)
        assert H.as_signed_pairs(r.alignment(0)) == expected
    subs, anchors, expected, sp = H.stitcher_known_answer()
    b = H.batch_from_graphs([(s[0], s[1]) for s in subs], np.array([s[2] for s in subs], np.uint8))
    r = gpu_ctx.stitch_batch_align(b, sp)
    stitched = []
    for k in range(len(subs)):
        stitched += H.as_signed_pairs(r.alignment(k))
        if k < len(anchors):
            stitched += anchors[k]
    assert stitched == expected


@pytest.mark.parametrize("seed,max_n,count", [(1, 8, 500), (2, 40, 400), (3, 100, 120), (4, 330, 40)])
def test_random_dags_vs_oracle(gpu_ctx, seed, max_n, count):
    b = synth.random_dag_batch(count, seed=seed, max_n=max_n)
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None


def test_chain_problems_all_kernel_geometries(gpu_ctx):
    """chain x chain sizes that hit every (rows per lane, waves) variant, strip and chunk boundaries"""
    sizes = [(1, 1), (1, 2), (2, 1), (63, 64), (64, 64), (65, 63), (64, 127), (128, 128), (129, 5), (5, 129),
             (256, 256), (257, 257), (300, 1500), (1024, 70), (1025, 64), (64, 1025), (2100, 190), (190, 2100),
             (3000, 2900), (40, 0), (0, 40), (0, 0)]
    for div in (0.0, 0.08, 0.6):
        lb = synth.linear_batch(sizes, seed=int(div * 100) + 7, divergence=div)
        assert gpu_ctx.stitch_batch_align(lb).same_as(po.oracle_stitch_batch(lb)) is None
        for npw in (1, 2, 3):
            f = np.full(lb.n_problems, npw, np.uint8)
            got = gpu_ctx.po_poa_batch(lb, f, capi.default_stitch_params().alignment_params)
            assert got.same_as(po.oracle_stitch_batch(lb, force_num_pw=f)) is None


def test_chain_kernel_agrees_with_general_kernel(gpu_ctx):
    """the same chain problems presented with a shuffled node numbering take the general (any-DAG) kernel;
    both device paths must give the oracle's answer"""
    rng = np.random.default_rng(5)
    b1, b2 = synth._SideBuilder(), synth._SideBuilder()
    for n1, n2 in [(30, 41), (90, 77), (200, 260)]:
        for bld, n in ((b1, n1), (b2, n2)):
            perm = rng.permutation(n)
            edges = [(int(perm[i]), int(perm[i + 1])) for i in range(n - 1)]
            bld.add_graph(rng.integers(1, 5, n, dtype=np.uint8), edges, [int(perm[0])], [int(perm[n - 1])])
    b = capi.StitchBatch(b1.finish(), b2.finish(), np.zeros(3, np.uint8))
    plan = gpu_ctx.plan(b)
    assert plan.stats()["n_linear"] == 3  # rank order makes them chains again
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None


def test_tie_heavy_chains(gpu_ctx):
    tp = H.tie_params()
    rng = np.random.default_rng(9)
    sizes = [(int(rng.integers(1, 200)), int(rng.integers(1, 200))) for _ in range(200)]
    b1, b2 = synth._SideBuilder(), synth._SideBuilder()
    for n1, n2 in sizes:
        b1.add_chain(rng.integers(1, 3, n1, dtype=np.uint8), 0)
        b2.add_chain(rng.integers(1, 3, n2, dtype=np.uint8), 0)
    b = capi.StitchBatch(b1.finish(), b2.finish(), np.zeros(len(sizes), np.uint8))
    assert gpu_ctx.stitch_batch_align(b, tp).same_as(po.oracle_stitch_batch(b, tp)) is None
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, tp.alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, tp, force_num_pw=f)) is None


def test_greedy_route_vs_oracle(gpu_ctx):
    """gaps that look unalignable (only_deletion_alns and above max_trivial_size) take greedy_partial_alignment on the host
    (alignment.hpp:1212-1611), next to the device problems of the same batch"""
    z = np.load(os.path.join(H.GOLDEN, "popoa_greedy.npz"))
    sb = synth.batch_from_intervals(z["seq1"], z["seq2"], z["rows"], np.ones(len(z["rows"]), np.uint8))
    got = gpu_ctx.stitch_batch_align(sb)
    assert (got.route[:-1] == 6).all()
    assert np.array_equal(got.aln_off, z["linear.aln_off"]) and np.array_equal(got.pairs, z["linear.pairs"])
    for seed, max_n, cnt in z["dag_cases"]:
        b = synth.random_dag_batch(int(cnt), seed=int(seed), max_n=int(max_n))
        b.only_deletion_alns[:] = 1
        assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None


def test_every_route_in_one_batch(gpu_ctx):
    """a batch in which every route of Stitcher::do_alignment occurs (device PO-POA next to the host heuristics), against the
    compiled reference's results for the whole batch"""
    z = np.load(os.path.join(H.GOLDEN, "host_routes.npz"))
    for tag, (batch, params) in H.host_route_batches.build(z["seq1"], z["seq2"]).items():
        got = gpu_ctx.stitch_batch_align(batch, params)
        assert np.array_equal(got.aln_off, z[tag + ".aln_off"]) and np.array_equal(got.pairs, z[tag + ".pairs"]), tag
        if tag == "mixed_dags":
            assert {1, 2, 3, 4, 6} <= set(np.unique(got.route).tolist())


def test_a_batch_larger_than_one_workspace_runs_in_chunks(gpu_ctx, monkeypatch):
    """one plan holds one workspace for its whole batch; a batch that would exceed it (CL_STITCH_WORKSPACE_WORDS, 12 GB by default) is run as
    consecutive chunks of subproblems on the same context and gives the same result — here with the limit set so low that the 700-problem
    batches fall into dozens of chunks, and with three 30 M-cell graph x graph pairs under a limit that puts each into a chunk of its own"""
    z = np.load(os.path.join(H.GOLDEN, "popoa_random_dags.npz"))
    batch = H.load_batch(z)
    want = gpu_ctx.stitch_batch_align(batch)
    monkeypatch.setenv("CL_STITCH_WORKSPACE_WORDS", "60000")
    got = gpu_ctx.stitch_batch_align(batch)
    assert got.same_as(want) is None and np.array_equal(got.route, want.route) and np.array_equal(got.num_pw, want.num_pw)
    assert got.same_as(H.load_result(z, "subalign.")) is None          # the compiled reference's results for this batch
    big = synth.sized_dag_batch([(5400, 5500)] * 3, seed=77)
    assert big.dp_cells() > 85000000
    monkeypatch.setenv("CL_STITCH_WORKSPACE_WORDS", str(5401 * 5501 * 7 + 1000))
    got = gpu_ctx.stitch_batch_align(big)
    monkeypatch.delenv("CL_STITCH_WORKSPACE_WORDS")
    want = gpu_ctx.stitch_batch_align(big)
    assert got.same_as(want) is None
    for k in range(3):
        H.check_alignment_valid(big, got, k)


def test_error_reporting(gpu_ctx):
    bad = capi.default_stitch_params()
    bad.alignment_params.gap_open[:] = [60, 50, 2500]
    with pytest.raises(capi.ClError) as e:
        gpu_ctx.stitch_batch_align(synth.linear_batch([(5, 5)]), bad)
    assert e.value.code == -2
    # a cyclic "subgraph"
    b1, b2 = synth._SideBuilder(), synth._SideBuilder()
    b1.add_graph(np.array([1, 2], np.uint8), [(0, 1), (1, 0)], [0], [1])
    b2.add_chain(np.array([1, 2], np.uint8), 0)
    with pytest.raises(capi.ClError) as e:
        gpu_ctx.stitch_batch_align(capi.StitchBatch(b1.finish(), b2.finish(), np.zeros(1, np.uint8)))
    assert e.value.code == -7


def test_lopsided_and_large_dag_pairs(gpu_ctx):
    """graph x graph matrices far from square (a whole-repeat indel between two MSA graphs: 7 x 2 051, 83 x 4 398) in both orientations,
    mid-size squares, a 2 000 x 2 000 pair and a 10 M-cell pair, against the oracle: the systolic kernel (shorter side on the threads,
    either graph), the LDS-ring and the HBM-plane kernels all take part"""
    sizes = [(7, 2051), (2051, 7), (83, 4398), (4398, 83), (60, 3000), (200, 2500), (2500, 200), (1, 900), (900, 1), (63, 64), (64, 63),
             (255, 256), (256, 255), (500, 500), (1000, 1200), (1023, 1500), (1024, 1500), (2000, 2000), (1000, 10000)]
    b = synth.sized_dag_batch(sizes, seed=5)
    got = gpu_ctx.stitch_batch_align(b)
    want = po.oracle_stitch_batch(b)
    assert got.same_as(want) is None
    for npw in (1, 2, 3):
        small = synth.sized_dag_batch(sizes[:12], seed=40 + npw, extra_edge_p=0.4, skip_max=6, alphabet=2)
        f = np.full(small.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(small, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(small, force_num_pw=f)) is None


def test_strips_of_rows_for_large_branching_pairs(gpu_ctx, monkeypatch):
    """(CL_NO_LANE=1: since round 5 near-chain pairs — most of these — take popoa_lane_kernel first; the strips stay the route of large pairs with denser bubbles, and this
    test keeps them covered on the same inputs as in round 4.)
    popoa_strip_kernel (round 4): branching pairs whose rows do not fit one workgroup's LDS are cut into strips of rows, one workgroup per strip,
    every strip reading the last rows of the one in front through a hand-off area in HBM while both run.  Against the oracle: both orientations,
    one to twenty strips, sparse and dense bubbles, in-degrees beyond the six predecessors a column record carries, every NumPW"""
    monkeypatch.setenv("CL_NO_LANE", "1")
    seen = set()
    for kw, sizes in ((dict(), [(2000, 2000), (1000, 10000), (10000, 1000), (700, 900), (3000, 400), (192, 1700)]),
                      (dict(extra_edge_p=0.02, skip_max=2), [(2000, 2000), (5500, 5500), (900, 700), (400, 6000)]),
                      (dict(extra_edge_p=0.4, skip_max=6, alphabet=2), [(1500, 1300), (640, 2000)])):
        b = synth.sized_dag_batch(sizes, seed=17, **kw)
        plan = gpu_ctx.plan(b)
        for li in plan.launches():
            seen.add(li["kernel"].split("<")[0])
        plan.destroy()
        got = gpu_ctx.stitch_batch_align(b)
        assert got.same_as(po.oracle_stitch_batch(b)) is None, kw
    assert "popoa_strip_kernel" in seen, seen
    for npw in (1, 2, 3):
        b = synth.sized_dag_batch([(900, 1100), (2500, 400), (1300, 1300)], seed=60 + npw, extra_edge_p=0.1, skip_max=3)
        f = np.full(b.n_problems, npw, np.uint8)
        plan = gpu_ctx.plan(b, force_num_pw=f)
        assert any(li["kernel"].startswith("popoa_strip_kernel<%d>" % npw) for li in plan.launches()), plan.launches()
        plan.destroy()
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
    # a few forks that reach hundreds of columns back: SAVED columns (their cells kept in an LDS area of their own, ghost rows included), every NumPW
    for npw in (1, 2, 3):
        b = synth.far_fork_batch([(1200, 2500), (700, 4000), (2000, 2100)], seed=70 + npw, n_far=2 + npw)
        f = np.full(b.n_problems, npw, np.uint8)
        plan = gpu_ctx.plan(b, force_num_pw=f)
        assert any(li["kernel"].startswith("popoa_strip_kernel") for li in plan.launches()), plan.launches()
        plan.destroy()
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
    # the far forks in the SHORTER graph: its predecessors reach too far back for ghost rows, so the longer graph gives the rows and the shorter one the columns
    b = synth.far_fork_batch([(2500, 1200), (3000, 900), (1500, 1400)], seed=91, n_far=3)
    plan = gpu_ctx.plan(b)
    assert all(li["kernel"].startswith("popoa_strip_kernel") for li in plan.launches()), plan.launches()
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None
    # pairs of different NumPW in one plan: their strip launches run side by side on different streams, each with progress words of its own
    # (more than 1 024 rows each: shorter near-chain pairs are popoa_lane_kernel's since round 5)
    b = synth.sized_dag_batch([(1100, 1200), (1500, 1100), (1300, 2000), (1200, 1200), (1800, 1500), (2000, 1700)], seed=21, extra_edge_p=0.05, skip_max=3)
    f = np.array([2, 3, 1, 2, 3, 1], np.uint8)
    plan = gpu_ctx.plan(b, force_num_pw=f)
    assert len({li["kernel"] for li in plan.launches() if li["kernel"].startswith("popoa_strip_kernel")}) == 3, plan.launches()
    want = po.oracle_stitch_batch(b, force_num_pw=f)
    for _ in range(6):
        plan.execute(); plan.sync()
        assert plan.collect().same_as(want) is None
    plan.destroy()
    # a chain pair of 4 096 rows and more took the strips too in rounds 4-5 (the chain kernel would need four passes of its one workgroup); since round 6 chain
    # pairs above 1 024 rows span several workgroups of the chain kernel (popoa_linear_span_kernel), and CL_LINEAR_SPAN=0 is the older routing
    b = synth.sized_dag_batch([(4500, 4200), (4100, 9000)], seed=8, extra_edge_p=0.0, n_alt=0)
    want = po.oracle_stitch_batch(b)
    plan = gpu_ctx.plan(b)
    assert all(li["kernel"] == "popoa_linear_span_kernel" for li in plan.launches() if li["n_problems"]), plan.launches()
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(b).same_as(want) is None
    monkeypatch.setenv("CL_LINEAR_SPAN", "0")
    plan = gpu_ctx.plan(b)
    assert all(li["kernel"].startswith("popoa_strip_kernel") for li in plan.launches()), plan.launches()
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(b).same_as(want) is None
    monkeypatch.delenv("CL_LINEAR_SPAN")
    # a plan executed again and again: the strips' progress words start every pass at zero
    b = synth.sized_dag_batch([(2000, 2000), (800, 1500)], seed=3)
    plan = gpu_ctx.plan(b)
    want = po.oracle_stitch_batch(b)
    for _ in range(3):
        plan.execute(); plan.sync()
        assert plan.collect().same_as(want) is None
    plan.execute_profiled(); plan.sync()
    assert plan.collect().same_as(want) is None
    plan.destroy()


def test_strips_that_give_up_are_run_again_by_the_anti_diagonal_kernel():
    """the strips of a pair wait for one another with a bound; a pair whose strips report failure (CL_STRIP_DEBUG_FAIL=1 makes every second pair do so — the
    switch is read once, hence the child process) is run again by the kernel that waits for nobody, and the caller sees the same alignments"""
    import subprocess
    import sys
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from centrolign_amd import capi, synth\n"
            "from oracle import pyoracle as po\n"
            "ctx = capi.Context(0)\n"
            "b = synth.sized_dag_batch([(900, 700), (1300, 800), (600, 2000), (2000, 600), (700, 700)], seed=33, extra_edge_p=0.05, skip_max=3)\n"
            "plan = ctx.plan(b)\n"
            "assert any(li['kernel'].startswith('popoa_strip_kernel') for li in plan.launches())\n"
            "plan.execute(); plan.sync(); got = plan.collect()\n"
            "print(json.dumps(dict(same=got.same_as(po.oracle_stitch_batch(b)) is None, fallbacks=plan.stats()['n_strip_fallbacks'])))\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CL_STRIP_DEBUG_FAIL="1", CL_NO_LANE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["same"] and res["fallbacks"] >= 2, res


def test_dag_pairs_at_the_lds_ceiling(gpu_ctx, monkeypatch):
    """(CL_NO_LANE=1, as for the strips above: the systolic kernel's LDS classes on the inputs of rounds 2-4.)
    branching pairs whose per-row column rings fill the systolic kernel's LDS budget (launches with 98-160 KB of dynamic LDS), next to
    ones that overflow it and take the strip kernel, the LDS-ring or the HBM-plane kernel: all against the oracle"""
    monkeypatch.setenv("CL_NO_LANE", "1")
    sizes = [(480, 500), (510, 300), (300, 509), (440, 2000), (1000, 620), (255, 3000), (64, 5000), (150, 4000), (191, 2500), (180, 500)]
    seen = set()
    for kw in (dict(extra_edge_p=0.02, skip_max=2), dict(extra_edge_p=0.05, skip_max=3), dict(extra_edge_p=0.15, skip_max=2)):
        b = synth.sized_dag_batch(sizes, seed=11, **kw)
        plan = gpu_ctx.plan(b)
        for li in plan.launches():
            seen.add((li["kernel"].split("<")[0], li["lds_bytes"] > 128 * 1024))
        plan.destroy()
        got = gpu_ctx.stitch_batch_align(b)
        assert got.same_as(po.oracle_stitch_batch(b)) is None, kw
    # (since round 4 the pairs of 192 rows and more that overflow the systolic kernel's LDS go to the strip kernel; the narrower ones still take the
    # LDS-ring or the HBM-plane kernel)
    names = {k for k, _ in seen}
    assert ("popoa_sys_kernel", True) in seen and "popoa_strip_kernel" in names, seen
    # ... and one whose column predecessors reach further back than any ring holds (a 1 500-column fork): not for the strips, not for the systolic kernel
    b = synth.sized_dag_batch([(150, 4000), (400, 2000)], seed=12, extra_edge_p=0.05, skip_max=1500)
    plan = gpu_ctx.plan(b)
    kernels = {li["kernel"].split("<")[0] for li in plan.launches()}
    plan.destroy()
    assert kernels & {"popoa_ring_kernel", "popoa_general_kernel"}, kernels
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None


def _kernels(plan):
    return {li["kernel"].split("<")[0] for li in plan.launches() if li["n_problems"]}


@pytest.mark.parametrize("round_waves", [3, 4])
def test_near_chain_pairs_in_registers(gpu_ctx, monkeypatch, round_waves):
    """(round_waves: strips per round of a pair above 64 rows = waves of its workgroup: three by default — a workgroup's step costs 0.30 us with three active waves, 0.55 with
    four —, four as first built: CL_LANE_WAVES.)
    (CL_LANE_MIN_SWEEP=0: by default only pairs of 512 rows + columns and more take this kernel — where it pays — here every eligible pair does.)
    popoa_lane_kernel (popoa_lane.h): graph pairs that are chains but for SNP / short-deletion bubbles and a long bubble or two — the long sweeps of a progressive MSA's stitch
    passes — swept in registers with DPP moves (row predecessors on a conveyor, column predecessors in the lane's history, saved columns in LDS).  Every workgroup shape
    (1 / 4 waves: strips of 64 rows pipelined over the waves, further rounds for more than 256 rows, several workgroups above 512 rows), both orientations,
    every NumPW, both shapes of the cell (predecessors 2 rows / 3 columns back, 4 / 4), against the oracle"""
    monkeypatch.setenv("CL_LANE_MIN_SWEEP", "0")
    monkeypatch.setenv("CL_LANE_MAX_ROWS", "1024")   # (default 192 rows: up to three active waves; beyond, rounds of four waves — slower per step, parity-tested here)
    monkeypatch.setenv("CL_LANE_WIDE", "0")
    monkeypatch.setenv("CL_LANE_WAVES", str(round_waves))
    # lopsided pairs, the long graph with a long bubble (a saved column): 1 wave, 4 waves, 8 waves, two rounds
    sizes = [(5, 2100), (30, 700), (64, 300), (65, 400), (165, 2225), (256, 500), (300, 330), (420, 418), (512, 520), (600, 640), (1000, 1010)]
    b = synth.near_chain_batch(sizes, seed=5, n_long=(0, 1))
    plan = gpu_ctx.plan(b)
    plan.execute(); plan.sync()
    names = [li["kernel"] for li in plan.launches() if li["n_problems"]]
    assert {"popoa_lane_kernel<1>", "popoa_lane_kernel<%d>" % round_waves} <= set(names), names
    # (the generator's deletion bubbles behind an SNP bubble reach five ranks back now and then: such a pair is not a lane pair)
    assert sum(li["n_problems"] for li in plan.launches() if li["kernel"].startswith("popoa_lane_kernel")) >= len(sizes) // 2, plan.launches()
    want = po.oracle_stitch_batch(b)
    assert plan.collect().same_as(want) is None
    for _ in range(3):                      # a resident plan executed again
        plan.execute(); plan.sync()
        assert plan.collect().same_as(want) is None
    plan.destroy()
    # the other orientation (graph 2 gives the rows), long bubbles in the ROW graph too (then it is not a lane pair: whatever kernel takes it must agree)
    for seed, nl in ((6, (1, 0)), (7, (1, 1)), (8, (0, 2))):
        b = synth.near_chain_batch([(700, 30), (2100, 12), (300, 64), (500, 200), (330, 300), (90, 90)], seed=seed, n_long=nl)
        assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None, (seed, nl)
    # every NumPW forced, tie-heavy parameters (match = mismatch = 1: every equality test of the traceback is exercised on the planes this kernel writes)
    for npw in (1, 2, 3):
        b = synth.near_chain_batch([(40, 900), (130, 600), (280, 300), (700, 90), (20, 20), (3, 400)], seed=20 + npw, p_snp=0.08, p_del=0.06, n_long=(0, 1))
        f = np.full(b.n_problems, npw, np.uint8)
        plan = gpu_ctx.plan(b, force_num_pw=f)
        assert "popoa_lane_kernel" in _kernels(plan), plan.launches()
        plan.destroy()
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
        tp = H.tie_params()
        got = gpu_ctx.po_poa_batch(b, f, tp.alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f, params=tp)) is None, ("ties", npw)
    # dense bubbles: predecessors up to four ranks back in both graphs (the 4 / 4 shape), in-degrees up to 4, several sources
    b = synth.sized_dag_batch([(50, 60), (200, 220), (400, 90), (64, 1000), (520, 530), (7, 7), (1, 30), (30, 1)], seed=44, extra_edge_p=0.3, skip_max=3, n_alt=2)
    plan = gpu_ctx.plan(b)
    assert "popoa_lane_kernel" in _kernels(plan), plan.launches()
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None
    if round_waves != 4:
        return
    monkeypatch.setenv("CL_LANE_WIDE", "1")
    # WIDE pairs (CL_LANE_WIDE=1: off by default, slower than the strips so far): more than 192 rows — groups of eight strips, a workgroup each, on different compute units, progress words between them; chain pairs of 4 096 rows
    # and more take this route too; saved columns whose cells cross a group boundary; every NumPW; a resident plan executed again
    b = synth.near_chain_batch([(1100, 1200), (2000, 2100), (1500, 5200), (3000, 3100)], seed=31, p_snp=0.03, p_del=0.01, n_long=(0, 1), long_min=300, long_max=700)
    plan = gpu_ctx.plan(b)
    assert sum(li["n_problems"] for li in plan.launches() if li["kernel"] == "popoa_lane_kernel<4, wide>") >= 3, plan.launches()
    want = po.oracle_stitch_batch(b)
    for _ in range(3):
        plan.execute(); plan.sync()
        assert plan.collect().same_as(want) is None
    assert plan.stats()["n_strip_fallbacks"] == 0
    plan.destroy()
    b = synth.near_chain_batch([(1100, 1200), (2000, 2100), (1500, 5200), (3000, 3100), (5000, 1500), (1030, 300)], seed=32, p_snp=0.03, p_del=0.01, n_long=(1, 1), long_min=300, long_max=700)
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None      # (long bubbles in the row graph too: whatever kernel takes such a pair)
    for npw in (1, 2, 3):
        b = synth.sized_dag_batch([(1300, 1300), (2500, 1100), (1100, 2600)], seed=80 + npw, extra_edge_p=0.1, skip_max=2)
        f = np.full(b.n_problems, npw, np.uint8)
        plan = gpu_ctx.plan(b, force_num_pw=f)
        assert all(li["kernel"] == "popoa_lane_kernel<4, wide>" for li in plan.launches() if li["n_problems"]), plan.launches()
        plan.destroy()
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
    lin = synth.linear_batch([(4100, 4300), (6300, 6300), (4096, 9000)], seed=14)
    want_lin = po.oracle_stitch_batch(lin)
    plan = gpu_ctx.plan(lin)   # (round 6: chain pairs above 1 024 rows go to the chain kernel over several workgroups whatever CL_LANE_WIDE says ...)
    assert all(li["kernel"] == "popoa_linear_span_kernel" for li in plan.launches() if li["n_problems"]), plan.launches()
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(lin).same_as(want_lin) is None
    monkeypatch.setenv("CL_LINEAR_SPAN", "0")   # (... and with the older routing they are WIDE pairs)
    plan = gpu_ctx.plan(lin)
    assert all(li["kernel"] == "popoa_lane_kernel<4, wide>" for li in plan.launches() if li["n_problems"]), plan.launches()
    plan.destroy()
    assert gpu_ctx.stitch_batch_align(lin).same_as(want_lin) is None
    monkeypatch.delenv("CL_LINEAR_SPAN")
    # many small near-chain pairs in one launch
    rng = np.random.default_rng(9)
    sizes = [(int(rng.integers(1, 120)), int(rng.integers(1, 400))) for _ in range(300)]
    b = synth.near_chain_batch(sizes, seed=10, n_long=(0, 0))
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None


def test_far_forks_in_both_graphs_take_the_level_order(gpu_ctx, monkeypatch):
    """(round 5) Long bubbles whose two branches are BOTH long — two diverged copies of a repeat unit — in both graphs of a pair.  In the reference's topological order
    (branches one after the other) the second branch and the closing node read a whole branch back in the row graph AND the column graph: such a pair was left to the
    anti-diagonal sweep in rounds 1-4 (missing #3).  The packer now ranks a graph by LEVEL (longest path from a source) when that shortens its reads
    (choose_rank_order, cl_api.cpp): the branches interleave, the pair meets the conditions of the register kernel / the strips.  Nothing in the result may depend on
    the order: against the oracle, and the reference's order (CL_RANK_ORDER=lifo) against the level order (=level) and the choice (default), every NumPW"""
    monkeypatch.setenv("CL_LANE_MIN_SWEEP", "0")
    sizes = [(3000, 3200), (1500, 1400), (700, 650), (300, 2500), (2600, 420), (150, 160)]
    b = synth.near_chain_batch(sizes, seed=41, p_snp=0.03, p_del=0.01, n_long=(2, 2), long_min=100, long_max=300, long_other=12)
    want = po.oracle_stitch_batch(b)
    kernels = {}
    for order in ("lifo", "level", ""):
        if order:
            monkeypatch.setenv("CL_RANK_ORDER", order)
        else:
            monkeypatch.delenv("CL_RANK_ORDER")
        plan = gpu_ctx.plan(b)
        plan.execute(); plan.sync()
        kernels[order] = {li["kernel"].split("<")[0]: li["n_problems"] for li in plan.launches() if li["n_problems"]}
        assert plan.collect().same_as(want) is None, order
        plan.destroy()
    # the reference's order leaves (at least) the large pairs to the anti-diagonal kernels; the choice leaves none
    slow = ("popoa_general_kernel", "popoa_ring_kernel")
    assert any(k in kernels["lifo"] for k in slow), kernels
    assert not any(k in kernels[""] for k in slow), kernels
    assert kernels[""] == kernels["level"], kernels
    for npw in (1, 2, 3):
        b = synth.near_chain_batch([(1300, 1250), (400, 380), (90, 2000)], seed=50 + npw, p_snp=0.05, p_del=0.02, n_long=(1, 2), long_min=60, long_max=200, long_other=20)
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
        tp = H.tie_params()
        got = gpu_ctx.po_poa_batch(b, f, tp.alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f, params=tp)) is None, ("ties", npw)
    # random DAGs with dense extra edges (several sources and sinks): forced level order against the oracle
    monkeypatch.setenv("CL_RANK_ORDER", "level")
    b = synth.random_dag_batch(300, seed=77, max_n=60)
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None
    b = synth.sized_dag_batch([(500, 500), (64, 1000), (2000, 2000), (255, 256)], seed=78)
    assert gpu_ctx.stitch_batch_align(b).same_as(po.oracle_stitch_batch(b)) is None


def test_small_chain_pairs_four_per_wave(gpu_ctx):
    """popoa_linear_quad_kernel: chain pairs whose shorter side has at most 16 nodes, four to a wave (16 lanes each, row-wise DPP moves): every length of the short side,
    either orientation, long and short partners in one quad, counts that are not a multiple of four, every NumPW, tie-heavy scoring; against the oracle"""
    rng = np.random.default_rng(3)
    sizes = [(a, int(rng.integers(1, 400))) for a in range(1, 17)] + [(int(rng.integers(1, 400)), a) for a in range(1, 17)] + [(16, 16), (1, 1), (16, 2000), (2000, 16), (3, 3)]
    sizes += [(int(rng.integers(1, 17)), int(rng.integers(1, 60))) for _ in range(401)]
    b = synth.linear_batch(sizes, seed=8)
    plan = gpu_ctx.plan(b)
    assert any(li["kernel"] == "popoa_linear_quad_kernel" and li["n_problems"] >= 400 for li in plan.launches()), plan.launches()
    want = po.oracle_stitch_batch(b)
    for _ in range(2):
        plan.execute(); plan.sync()
        assert plan.collect().same_as(want) is None
    plan.destroy()
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
        tp = H.tie_params()
        got = gpu_ctx.po_poa_batch(b, f, tp.alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f, params=tp)) is None, ("ties", npw)


def test_passes_of_a_resident_plan_overlap(gpu_ctx, monkeypatch):
    """(second half of round 5) A pass of a plan no longer makes the context's stream wait for the auxiliary streams its launches went to: the next pass's launches follow in
    stream order, passes enqueued back to back overlap, and whoever needs a pass complete joins first (cl_stitch_join: sync, collect, the profiled pass, a re-dealing of
    the launches, the chaining DP, destroy).  Every kernel kind in one plan, twenty passes without a wait between them — through the first-stage and the second-stage
    re-dealing of the launches — then the results; a second plan of the same context interleaved pass by pass; a chaining DP of the same context right behind an
    unjoined pass; the launch clocks (pass-tagged, never zeroed between passes) give a duration for every launch of the last pass; and the same with a join per pass"""
    sizes = [(5, 2100), (300, 330), (40, 44), (12, 9), (700, 650), (64, 300), (2000, 2000), (1500, 1300), (130, 4000)] + [(int(a), int(b)) for a, b in np.random.default_rng(5).integers(1, 90, (400, 2))]
    b = synth.near_chain_batch(sizes[:8], seed=61, n_long=(0, 1))
    b = capi.StitchBatch.concat([b, synth.linear_batch(sizes, seed=62), synth.sized_dag_batch([(500, 500), (64, 1000), (255, 256), (1200, 1100)], seed=63)])
    want = po.oracle_stitch_batch(b)
    other = synth.sized_dag_batch([(90, 100), (400, 380), (30, 900)], seed=64)
    want_other = po.oracle_stitch_batch(other)
    for join in ("", "eager"):
        if join:
            monkeypatch.setenv("CL_STITCH_JOIN", join)
        plan = gpu_ctx.plan(b)
        assert len({li["kernel"].split("<")[0] for li in plan.launches() if li["n_problems"]}) >= 4, plan.launches()
        for _ in range(20):
            plan.execute()
        assert plan.collect().same_as(want) is None, join
        plan.sync()
        assert all(li["in_pass_ms"] > 0 for li in plan.launches() if li["n_problems"]), plan.launches()
        plan2 = gpu_ctx.plan(other)
        for _ in range(6):
            plan.execute(); plan2.execute()
        assert plan2.collect().same_as(want_other) is None, join
        assert plan.collect().same_as(want) is None, join
        # a chaining DP on the same context (its far launches use the same auxiliary streams) right behind a pass nobody has joined
        z = np.load(os.path.join(H.GOLDEN, sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("chain4_"))[0]))
        from tests.test_extraction import load_stitch_case
        name = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("chain4_"))[0]
        _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
        ms = capi.MatchSets(**{k: z["a.ms." + k] for k in capi.MatchSets._DT})
        cp = capi.default_chain_params(global_anchoring=False)
        plan.sync()
        quiet = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=float(z["a.scale"][0]), params=cp, want_dp=True)
        assert np.array_equal(quiet["chain"], z["a.chain_affine"])
        plan.execute()
        got = gpu_ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=float(z["a.scale"][0]), params=cp, want_dp=True)
        assert np.array_equal(got["chain"], quiet["chain"]) and np.array_equal(got["dp"].view(np.uint32), quiet["dp"].view(np.uint32))
        plan.execute_profiled(); plan.execute(); plan.execute()
        assert plan.collect().same_as(want) is None, join
        plan2.destroy(); plan.destroy()


def test_chain_pairs_two_per_wave(gpu_ctx, monkeypatch):
    """popoa_linear_duo_kernel (second half of round 5): chain pairs whose shorter side has 17 to 32 nodes, two to a wave (32 lanes each; the wave forms of the DPP moves,
    lane 32 takes its fill by a select): every length of the short side, either orientation, partners of very different length, an odd count, every NumPW, tie-heavy
    scoring; against the oracle, and the same pairs one per wave (CL_LINEAR_DUOS=0); in plans of 2 048 subproblems and more the kernel is not taken by default"""
    rng = np.random.default_rng(4)
    sizes = [(a, int(rng.integers(1, 500))) for a in range(17, 33)] + [(int(rng.integers(17, 500)), a) for a in range(17, 33)] + [(32, 32), (17, 17), (32, 3000), (3000, 17), (20, 21)]
    sizes += [(int(rng.integers(17, 33)), int(rng.integers(17, 90))) for _ in range(300)]
    b = synth.linear_batch(sizes, seed=9)
    plan = gpu_ctx.plan(b)
    assert any(li["kernel"] == "popoa_linear_duo_kernel" and li["n_problems"] == len(sizes) for li in plan.launches()), plan.launches()
    want = po.oracle_stitch_batch(b)
    for _ in range(2):
        plan.execute(); plan.sync()
        assert plan.collect().same_as(want) is None
    plan.destroy()
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        got = gpu_ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f)) is None, npw
        tp = H.tie_params()
        got = gpu_ctx.po_poa_batch(b, f, tp.alignment_params)
        assert got.same_as(po.oracle_stitch_batch(b, force_num_pw=f, params=tp)) is None, ("ties", npw)
    monkeypatch.setenv("CL_LINEAR_DUOS", "0")
    plan = gpu_ctx.plan(b)
    assert not any(li["kernel"] == "popoa_linear_duo_kernel" for li in plan.launches()), plan.launches()
    plan.execute(); plan.sync()
    assert plan.collect().same_as(want) is None
    plan.destroy()


def test_strips_and_wide_launches_on_a_busy_device(gpu_ctx):
    """ The strips of a pair wait for one another and rely on being resident together; the walk of the chaining DP and the wide form of the
    register kernel have waits of the same kind.  Here the device is otherwise FULL: three more contexts on threads of their own run chaining DPs (walk + far launches
    on six streams each) and stitch plans of thousands of small pairs back to back while this context runs pairs on the strips twenty times.  Every pass must give the
    oracle's alignments; a strip that gives up is re-run anti-diagonal-wise by design (counted, reported), a wrong result or a hang is a failure"""
    import threading
    from tests.test_extraction import load_stitch_case
    name = sorted(f for f in os.listdir(H.GOLDEN) if f.startswith("chain4_"))[0]
    z = np.load(os.path.join(H.GOLDEN, name))
    _, graphs, _ = load_stitch_case(name.replace("chain4_", "stitch4_"))
    ms = capi.MatchSets(**{k: z["a.ms." + k] for k in capi.MatchSets._DT})
    cp = capi.default_chain_params(global_anchoring=False)
    filler = synth.linear_batch([(int(a), int(b)) for a, b in np.random.default_rng(12).integers(20, 300, (6000, 2))], seed=13)
    stop = threading.Event()
    errors = []
    rounds = {"chain": 0, "stitch": 0}

    def chain_worker():
        try:
            c = capi.Context(0)
            while not stop.is_set():
                got = c.chain_sparse_affine(graphs[0], graphs[1], ms, scale=float(z["a.scale"][0]), params=cp)
                rounds["chain"] += 1
                if not np.array_equal(got["chain"], z["a.chain_affine"]):
                    errors.append("chain differs under load")
                    break
            c.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    def stitch_worker():
        try:
            c = capi.Context(0)
            p = c.plan(filler)
            while not stop.is_set():
                for _ in range(10):
                    p.execute()
                p.sync()
                rounds["stitch"] += 1
            p.destroy(); c.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=chain_worker), threading.Thread(target=chain_worker), threading.Thread(target=stitch_worker)]
    for t in threads:
        t.start()
    try:
        b = synth.sized_dag_batch([(2000, 2000), (1500, 4000), (5500, 5500), (900, 1300)], seed=23, extra_edge_p=0.05, skip_max=3)
        want = po.oracle_stitch_batch(b)
        plan = gpu_ctx.plan(b)
        assert any(li["kernel"].startswith("popoa_strip_kernel") for li in plan.launches()), plan.launches()
        fallbacks, passes = 0, 0
        while passes < 20 or ((rounds["chain"] < 6 or rounds["stitch"] < 6) and passes < 2000 and not errors):   # (until the other contexts have really been at work beside it)
            plan.execute(); plan.sync()
            assert plan.collect().same_as(want) is None
            fallbacks = plan.stats()["n_strip_fallbacks"]
            passes += 1
        plan.destroy()
    finally:
        stop.set()
        for t in threads:
            t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    assert rounds["chain"] >= 6 and rounds["stitch"] >= 6, rounds
    print("strip passes %d beside %d chaining DPs and %d x 10 stitch passes of 6 000 pairs on other contexts; strip fallbacks: %d" % (passes, rounds["chain"], rounds["stitch"], fallbacks))


@pytest.mark.gpu
def test_large_chain_pairs_span_several_workgroups(gpu_ctx, monkeypatch):
    """popoa_linear_span_kernel (round 6): chain pairs of more than 1 024 rows, their strips dealt over groups of four on different compute units (hand-off rows written
    through to memory, a progress word per group) — sizes on both sides of the routing boundary, either graph the shorter one, one / two / many groups, a last group of fewer
    than four strips, every NumPW, default and tie-heavy scoring, against the oracle; and with CL_SPAN_DEBUG_FAIL=1 every such pair reports a failed wait and is run again
    by the one-workgroup chain kernel on the same workspace: the same alignments, counted in n_strip_fallbacks"""
    sizes = [(1024, 1100), (1025, 1100), (1100, 1025), (1280, 1281), (1300, 4000), (4000, 1290), (2048, 2048), (2049, 2300), (3000, 2500), (4096, 4100), (1030, 9000)]
    b = synth.linear_batch(sizes, seed=71)
    plan = gpu_ctx.plan(b)
    names = {li["kernel"]: li["n_problems"] for li in plan.launches() if li["n_problems"]}
    plan.destroy()
    assert names.get("popoa_linear_span_kernel") == len(sizes) - 1, names          # (1 024 rows: still one workgroup)
    want = po.oracle_stitch_batch(b)
    assert gpu_ctx.stitch_batch_align(b).same_as(want) is None
    tp = H.tie_params()
    small = synth.linear_batch([(1025, 1200), (1500, 1100), (2100, 2050), (1300, 3000)], seed=72, divergence=0.3)
    for npw in (1, 2, 3):
        f = np.full(small.n_problems, npw, np.uint8)
        for params in (capi.default_stitch_params(), tp):
            got = gpu_ctx.po_poa_batch(small, f, params.alignment_params)
            assert got.same_as(po.oracle_stitch_batch(small, params, force_num_pw=f)) is None, npw
    # one pair at the size the stress set holds (6 300 x 6 300: 25 groups)
    big = synth.linear_batch([(6300, 6300)], seed=73)
    assert gpu_ctx.stitch_batch_align(big).same_as(po.oracle_stitch_batch(big)) is None
    # every wait "expires": the pairs come back from the one-workgroup kernel
    monkeypatch.setenv("CL_SPAN_DEBUG_FAIL", "1")
    plan = gpu_ctx.plan(b)
    plan.execute()
    got = plan.collect()
    st = plan.stats()
    plan.destroy()
    assert got.same_as(want) is None and st["n_strip_fallbacks"] == len(sizes) - 1

