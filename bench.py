#!/usr/bin/env python3
"""bench.py — the anchor-and-stitch hot path on MI355X, on BASELINE.json configs[2]: the progressive MSA of 10 synthetic 1 Mbp HOR
arrays (seed 7) over the guide tree of SURVEY.md §8(d) (driver contract: see the task statement / DESIGN.md §Measurement).

What runs:
  1. the whole MSA once, natively through the C ABI (cl_leaf_graph, cl_leaf_intrinsic_scale, nine cl_merge calls = match finding +
     Core::align + fuse), timed as `msa_wall_s`; its nine merges leave behind the nine stitch batches the reference's
     Stitcher::stitch would have aligned (every between-anchor subproblem of every merge);
  2. those batches are packed into nine resident stitch plans (inputs in HBM), and one "step" = one pass of the stitcher's
     PO-POA DP (fill + traceback of every subproblem) over all nine.  W warm-up steps, then exactly K timed steps bracketed by
     barrier + synchronize; `value` = DP cells of all steps / elapsed.

N > 1 (torchrun, one process per GPU): ONE MSA over the N ranks (centrolign_amd.msa.progressive_msa_distributed: leaf calibrations and
sibling subtrees of the guide tree on different ranks, fused graphs travel between owners) => `"scaling": "strong"` (at every N, 1 included:
the N = 1 line is the anchor of that curve).  The stitch
batches stay on the rank that made them; a step is every rank's pass over its own batches, `value` = all ranks' cells / the slowest
rank's time, `msa_wall_s` = the slowest rank's wall-clock of the distributed MSA.  No data-path collective in the timed steps.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

WORKLOAD = "BASELINE configs[2]: 10 x 1 Mbp synthetic HOR arrays (seed 7), guide tree ((((s0,s1),(s2,s3)),s4),(((s5,s6),(s7,s8)),s9)), " \
           "full progressive MSA; step = stitcher PO-POA DP over the stitch batches of all nine merges"


def relabelled(g, src_label, snk_label):
    """reassign_sentinels (core.hpp:287-288) on a copy: what cl_merge does to its inputs before anything else"""
    from centrolign_amd import capi
    lab = g.label.copy()
    lab[g.src_id], lab[g.snk_id] = src_label, snk_label
    return capi.BaseGraph(lab, g.next_off, g.next_idx, g.prev_off, g.prev_idx, g.path_off, g.path_nodes, g.src_id, g.snk_id)


def stitch_batches(kept):
    """the stitch batch of every kept merge: Extractor::extract_graphs_between on the anchor segments the merge stitched"""
    from centrolign_amd import capi
    out = []
    for m in kept:
        g1, g2 = relabelled(m["graphs"][0], 5, 6), relabelled(m["graphs"][1], 7, 8)
        al = m["align"]
        seg = capi.AnchorSegments(al["seg_off"], al["walk_off"], al["walk1"], al["walk2"])
        out.append((m["merge"], capi.extract_stitch_batch(g1, g2, seg)))
    return out


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "cpu_cores": os.cpu_count()}


def cpu_baseline(batches, min_seconds=8.0, max_reps=6):
    """single-thread CPU time of the same DP: the compiled reference's Stitcher::subalign where oracle/_ref travelled with the repo
    ("reference"), else our C restatement ("port"), on a bounded sample: the first leaf merge's batch and the first graph x graph
    (multi-path) merge's batch"""
    from oracle import pyoracle as po
    kind = "reference" if po.have_ref() else "port"
    sample = [batches[0]]
    multi = [b for b in batches if "((" in b[0] or b[0].count(",") > 1]
    if multi:
        sample.append(multi[0])
    cells = total = 0.0
    reps = 0
    while total < min_seconds and reps < max_reps:
        for _, b in sample:
            if kind == "reference":
                _, secs = po.ref_stitch_batch(b)
            else:
                t0 = time.perf_counter()
                po.oracle_stitch_batch(b)
                secs = time.perf_counter() - t0
            total += secs
            cells += b.dp_cells()
        reps += 1
    out = {"value": cells / total, "unit": "DP cells/s", "cores": 1, "kind": kind,
           "sample": "stitch batches of merges %s (%d subproblems, %d cells) x %d passes, %.1f s of CPU inside Stitcher::subalign; "
                     "the other merges' batches are the same kind of work" %
                     (" and ".join(m for m, _ in sample), sum(b.n_problems for _, b in sample), sum(b.dp_cells() for _, b in sample), reps, total)}
    out.update(host_info())
    if kind != "reference":
        out["skipped"] = "oracle/_ref (the compiled reference) is not on this box: the C restatement of the path was timed instead"
    return out


def reference_leaf_merge(seqs, names):
    """the reference's whole pipeline (Core::execute + explicit_cigar) on the first leaf pair of the guide tree, wall-clock: the one
    piece of configs[2] the single-threaded reference finishes within the bench's time budget"""
    import tempfile
    from centrolign_amd import synth
    from oracle import pyoracle as po
    if not po.have_ref():
        return {"skipped": "oracle/_ref is not on this box"}
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "in.fa")
        synth.write_fasta(fa, [seqs[n] for n in names[:2]], names[:2])
        t0 = time.perf_counter()
        tm = po.ref_msa_dump(fa, out_path=os.path.join(d, "o.txt"))
        secs = time.perf_counter() - t0
    return {"merge": "(%s,%s) incl. the two leaf calibrations" % (names[0], names[1]), "seconds": secs, "cores": 1, "kind": "reference", "phases_s": tm}


def big_dag_section(ctx):
    """one branching pair of 30 M cells (5 500 x 5 500 nodes: the reference's ceiling is 40 M, src/parameters.cpp:79) through popoa_strip_kernel — strips of
    rows, one workgroup each, all in flight — and, in a child process with CL_NO_STRIP=1, through the anti-diagonal kernel it took in rounds 1-3"""
    import subprocess
    from centrolign_amd import synth
    res = {}
    code = ("import sys, json, hashlib; sys.path.insert(0, %r)\n"
            "from centrolign_amd import capi, synth\n"
            "ctx = capi.Context(0); b = synth.sized_dag_batch([(5500, 5500)], seed=5, extra_edge_p=0.02, skip_max=2); plan = ctx.plan(b)\n"
            "for _ in range(2): plan.execute(); plan.sync()\n"
            "plan.execute_profiled(); plan.sync(); li = plan.launches()[0]\n"
            "r = plan.collect(); print(json.dumps(dict(kernel=li['kernel'], ms=li['ms'], cells=li['dp_cells'], score=int(r.score[0]), pairs=hashlib.sha256(r.pairs.tobytes()).hexdigest()[:16])))\n" % HERE)
    for key, env in (("strips", {}), ("anti_diagonal_kernel", {"CL_NO_STRIP": "1"})):
        try:
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=180)
            res[key] = json.loads(r.stdout.strip().splitlines()[-1])
            res[key]["g_cells_per_s"] = res[key]["cells"] / res[key]["ms"] / 1e6
        except Exception as e:   # noqa: BLE001
            res[key] = {"error": str(e)[:200]}
    if "ms" in res.get("strips", {}) and "ms" in res.get("anti_diagonal_kernel", {}):
        res["speedup"] = res["anti_diagonal_kernel"]["ms"] / res["strips"]["ms"]
        res["same_alignment"] = (res["strips"]["score"], res["strips"]["pairs"]) == (res["anti_diagonal_kernel"]["score"], res["anti_diagonal_kernel"]["pairs"])
    res["workload"] = "synth.sized_dag_batch([(5500, 5500)], seed 5, extra_edge_p 0.02, skip_max 2): 30.3 M cells, NumPW 3"
    # long bubbles with two long branches in BOTH graphs (missing #3): in the reference's topological order both graphs read a branch back and the pair
    # is left to the anti-diagonal kernel (CL_RANK_ORDER=lifo: rounds 1-4); ranked by level (the default choice, choose_rank_order in cl_api.cpp) it meets the strips' conditions
    code2 = code.replace("synth.sized_dag_batch([(5500, 5500)], seed=5, extra_edge_p=0.02, skip_max=2)",
                         "synth.near_chain_batch([(5000, 5200)], seed=3, p_snp=0.03, p_del=0.01, n_long=(3, 3), long_min=200, long_max=400, long_other=16)")
    ff = {}
    for key, env in (("level_order", {}), ("reference_order", {"CL_RANK_ORDER": "lifo"})):
        try:
            r = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, **env), capture_output=True, text=True, timeout=180)
            ff[key] = json.loads(r.stdout.strip().splitlines()[-1])
            ff[key]["g_cells_per_s"] = ff[key]["cells"] / ff[key]["ms"] / 1e6
        except Exception as e:   # noqa: BLE001
            ff[key] = {"error": str(e)[:200]}
    if "ms" in ff.get("level_order", {}) and "ms" in ff.get("reference_order", {}):
        ff["speedup"] = ff["reference_order"]["ms"] / ff["level_order"]["ms"]
        ff["same_alignment"] = (ff["level_order"]["score"], ff["level_order"]["pairs"]) == (ff["reference_order"]["score"], ff["reference_order"]["pairs"])
    ff["workload"] = "synth.near_chain_batch([(5000, 5200)], seed 3, n_long (3, 3), long 200-400, long_other 16): two-long-branch bubbles in both graphs"
    res["far_forks_in_both_graphs"] = ff
    return res


def chaining_section(ctx, with_reference):
    """seam S3 at BASELINE configs[1] scale: Anchorer::anchor_chain in the CLI's default configuration on the match sets of the
    2 x 1 Mbp pair (bench_data/c2_chain_input.npz where it travelled with the repo, else made in place), the inner affine DP alone,
    the whole Core::align, and — with oracle/_ref — the compiled reference on the same input with a field-by-field comparison"""
    from centrolign_amd import capi, synth
    path = os.path.join(HERE, "bench_data", "c2_chain_input.npz")
    if os.path.exists(path):
        z = np.load(path)
        graphs = []
        for side in ("parent1.", "parent2."):
            t = z[side + "tableau"]
            graphs.append(capi.BaseGraph(*[z[side + k] for k in ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes")], t[0], t[1]))
        ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
        score_scale = float(z["score_scale"][0])
        source = "bench_data/c2_chain_input.npz (graphs, match sets and calibrated scale dumped from the reference's run)"
    else:
        seqs = synth.hor_sequences(7, 1000000, 2)
        graphs = [synth.base_graph_from_sequence(seqs[0]), synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8))]
        ms = ctx.find_matches(graphs[0], graphs[1])
        scales = [ctx.leaf_intrinsic_scale(g) for g in graphs]
        score_scale = sum(scales) / len(scales)
        source = "made in place (bench_data/ did not travel): cl_find_matches + cl_leaf_intrinsic_scale on the synthetic pair"

    def run():
        t0 = time.perf_counter()
        split = capi.split_branching_matches(graphs[0], graphs[1], ms)
        got = ctx.anchor_chain(graphs[0], graphs[1], split, score_scale=score_scale)
        return got, time.perf_counter() - t0
    run()
    got, wall = run()
    n = ms.n_pairs()
    out = {"seam": "Anchorer::anchor_chain (default configuration), 2 x 1 Mbp pair", "input": source, "match_sets": ms.n_sets, "match_pairs": n,
           "chain_anchors": int(len(got["chain"])), "estimated_scale": got["scale"], "tie_resolutions": got["n_ties"], "wall_s": wall}
    if n <= 1300000:
        dp = ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=got["scale"])
        out["affine_dp_only"] = {"match_pairs": dp["n_pairs"], "device_dp_ms": dp["device_ms"], "host_prep_ms": dp["prep_ms"],
                                 "value_index_ms": dp["index_ms"], "traceback_ms": dp["traceback_ms"]}
    t0 = time.perf_counter()
    al = ctx.core_align(graphs[0], graphs[1], ms, score_scale=score_scale)
    out["core_align"] = {"wall_s": time.perf_counter() - t0, "chain_ms": al["chain_ms"], "partition_ms": al["partition_ms"],
                         "stitch_ms": al["stitch_ms"], "aligned_pairs": int(len(al["alignment"]))}
    if with_reference:
        from oracle import pyoracle as po
        if po.have_ref():
            t0 = time.perf_counter()
            split = po.ref_split_branching_matches(graphs[0], graphs[1], ms)
            ref = po.ref_anchor_chain(graphs[0], graphs[1], split, score_scale=score_scale, fill_in=True)
            secs = time.perf_counter() - t0
            same = all(np.array_equal(ref[k], got[k]) for k in ("set_order", "chain", "walk1", "walk2", "gap_before", "gap_after",
                                                                 "gap_score_before", "gap_score_after", "score"))
            out["cpu_reference"] = {"seconds": secs, "cores": 1, "kind": "reference", "identical_result": bool(same and ref["scale"] == got["scale"])}
        else:
            out["cpu_reference"] = {"skipped": "oracle/_ref is not on this box"}
    return out


def pairwise_section(ctx, with_reference):
    """BASELINE configs[1] end to end without the reference in the loop: sequences -> leaf graphs -> calibration -> match finding ->
    Core::align -> fuse -> explicit CIGAR, wall-clock; beside it the compiled reference's whole pipeline and a byte comparison"""
    import tempfile
    from centrolign_amd import msa, synth
    seqs = synth.hor_sequences(7, 1000000, 2)
    names = ["seq0", "seq1"]
    t0 = time.perf_counter()
    r = msa.progressive_msa(ctx, dict(zip(names, seqs)), msa.balanced_tree(names), workers=2)
    text = msa.output_text(r)
    wall = time.perf_counter() - t0
    out = {"pipeline": "BASELINE configs[1] (2 x 1 Mbp): leaf graphs + calibration + find_matches + Core::align + fuse + explicit_cigar", "wall_s": wall,
           "match_ms": r["stats"]["match_ms"], "align_ms": r["stats"]["align_ms"], "cigar_bytes": len(text), "cigar_sha256": hashlib.sha256(text).hexdigest()}
    if with_reference:
        from oracle import pyoracle as po
        if po.have_ref():
            with tempfile.TemporaryDirectory() as d:
                fa, o = os.path.join(d, "in.fa"), os.path.join(d, "out.txt")
                synth.write_fasta(fa, seqs, names)
                t0 = time.perf_counter()
                tm = po.ref_msa_dump(fa, out_path=o)
                secs = time.perf_counter() - t0
                want = open(o, "rb").read().rstrip(b"\n")
            out["cpu_reference"] = {"seconds": secs, "cores": 1, "kind": "reference", "phases_s": tm, "identical_output": bool(want == text),
                                    "speedup": secs / wall}
        else:
            out["cpu_reference"] = {"skipped": "oracle/_ref is not on this box"}
    return out


def spawn_ranks(n):
    """`python -m torch.distributed.run --nproc-per-node n bench.py <the same arguments>` as a child; rank 0's JSON line goes to stdout"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workers", type=int, default=4, help="worker contexts of the single-GPU MSA (sibling merges side by side)")
    ap.add_argument("--length", type=int, default=1000000, help="sequence length (the headline is 1 000 000; smaller only for dry runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-skip", type=int, default=0, help="measurements only: CL_DEBUG_SKIP_TRACEBACK for the timed plans (1 no traceback, 3 no plane stores either); the line is then NOT a result")
    ap.add_argument("--share-merges", type=int, default=0,
                    help="N > 1: ranks per merge group (one merge over several GPUs: the far pass of its chaining DP divided between them); 0 = one rank per merge "
                         "(the default until merge groups have passed their self-test across real xGMI links: they have only run between processes on ONE device)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the MSA and the timed passes: the command the rocprofv3 summaries under profiles/ are taken with")
    ap.add_argument("--plans", choices=("one", "nine"), default="one",
                    help="the resident stitch batches of a rank as ONE plan (every subproblem of the nine merges in one batch: a dozen large launches) or as "
                         "one plan per merge run side by side (rounds 1-3: ~110 small launches that share the hardware queues); the other layout is timed too "
                         "and reported in config.other_plan_layout")
    ap.add_argument("--no-shard-stitch", action="store_true",
                    help="N > 1: keep every merge's stitch batch on the rank that made it instead of dealing the subproblems of ALL batches over the ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): start the N ranks ourselves, as a CHILD process, before anything here has touched the
        # GPU (nothing has: torch is not even imported yet; a process that has initialised HIP must never exec another program), relay the
        # ranks' output and leave with their exit code
        sys.exit(spawn_ranks(args.gpus))

    import torch
    from centrolign_amd import dist as cd
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = world_env > 1 and n_dev < world_env          # dry run of the multi-rank path on fewer devices: ranks share them, gloo collectives
    # before HIP initialises (device_count above does not): independent launches overlap on separate hardware queues.  Measured on the
    # nine concurrent stitch plans (ms per step): 4 queues 8.9, 8: 8.2, 12: 5.6, 16: 4.9, 20: 3.8, 22: 4.9, 24: 11.5, 32: 28.5 — beyond
    # ~23 PER DEVICE the queues are oversubscribed and time-sliced, so ranks that share a device share the 20; the MSA's wall-clock does
    # not depend on it (15.2-16.3 s for 8..23)
    # (the library asks for 20 itself when it is loaded — cl_api.cpp, cl_library_loaded — but in this process torch has loaded the HIP runtime
    # first, and the runtime has read its settings by then: measured 9.3 ms per step without the line below, 4.1 ms with it)
    per_device = max(1, -(-world_env // max(1, n_dev)))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, 20 // per_device)))
    rank, world, dist = cd.init_distributed(None if world_env == 1 else ("gloo" if share else "nccl"))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    device = local_rank % max(1, n_dev)
    torch.cuda.set_device(device)

    from centrolign_amd import capi, msa, synth
    ctx = capi.Context(device)          # raises without a GPU / without the HIP library: no fallback
    names, seqs, tree = synth.c3_workload(args.length)
    ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))   # first-use initialisation outside the timed regions
    # ... and (round 6) one whole merge of two 3-kbp arrays on the main context: the first use of every kernel family's code object — chaining, stitching, fuse — is a cost of the
    # PROCESS (the HIP runtime loads a code object when a kernel of it is first launched), not of the MSA; nothing of the MSA's size is allocated or computed here, and the worker
    # contexts of the timed MSA are still created — with empty pools — inside the timed region
    _w = synth.hor_sequences(3, 3000, 2)
    _wl = [capi.leaf_graph(x) for x in _w]
    ctx.merge(_wl[0], _wl[1], score_scale=sum(ctx.leaf_intrinsic_scale(g) for g in _wl) / 2)
    del _w, _wl

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- 1. the MSA ------------------------------------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    if world == 1:
        res = msa.progressive_msa(ctx, seqs, tree, workers=args.workers, keep_merges=True)
        host_group = None
    else:
        host_group = dist.new_group(backend="gloo") if not share else None
        # merges whose children come from different ranks run as merge groups of up to --share-merges ranks (cl_peer_api.cpp): every member runs
        # the merge, the far pass of its chaining DP is divided between the members' devices
        res = msa.progressive_msa_distributed(ctx, seqs, tree, dist, rank, world, group=host_group, keep_merges=True, all_ranks=True,
                                              workers=args.workers, share_merges=args.share_merges)
    msa_wall = time.perf_counter() - t0
    msa_wall = cd.max_over_ranks(msa_wall, dist, device="cpu" if share else "cuda")
    gfa_sha = gfa_bytes = None
    if rank == 0:
        gfa = capi.write_gfa(res["root"], res["paths"])
        gfa_sha, gfa_bytes = hashlib.sha256(gfa).hexdigest(), len(gfa)
    merge_groups_retry = None
    if world > 1 and args.share_merges > 1 and args.length == 1000000:
        # merge groups (one merge over several GPUs through peer stores) have only ever run between processes on ONE device here.  A GFA that is not the reference's is a
        # HARD failure (round-5 verdict item 8; round-4 advisor: a silent second run with one rank per merge masked exactly the class of fault a ring-slot collision would be):
        # every rank leaves with a non-zero exit code and no line is printed
        want = None
        try:
            with open(os.path.join(HERE, "tests", "golden", "c3_10x1M_subproblems.json")) as f:
                want = json.load(f)["root_default_budget_reference"]["sha256"]
        except Exception:   # noqa: BLE001
            pass
        force = os.environ.get("CL_BENCH_FORCE_RETRY") == "1"   # test hook: behave as if the GFA had been wrong
        flag = torch.tensor([1 if (rank != 0 or want is None or (gfa_sha == want and not force)) else 0], dtype=torch.int32)
        dist.broadcast(flag, 0, group=host_group)
        if int(flag.item()) == 0:
            if rank == 0:
                print("bench.py: the MSA run with merge groups of %d ranks printed GFA %s, the reference's is %s: wrong multi-GPU result, nothing is timed" % (args.share_merges, gfa_sha, want), file=sys.stderr)
            ctx.close()
            dist.destroy_process_group()
            sys.exit(3)
    kept = res["stats"].get("kept", []) if res is not None and "stats" in res else []
    # the same MSA once more in the same process (one GPU, not under --no-extras): msa_wall_s above is the FIRST run of the process, as in every round — it carries the first
    # use of every kernel's code object, of the pools and of the page-locked areas; the second run shows the wall-clock without them (its worker contexts are new ones again)
    msa_second = None
    if world == 1 and not args.no_extras:
        barrier()
        t0 = time.perf_counter()
        res2 = msa.progressive_msa(ctx, seqs, tree, workers=args.workers)
        msa_second = {"seconds": time.perf_counter() - t0}
        gfa2 = capi.write_gfa(res2["root"], res2["paths"])
        msa_second["same_gfa"] = bool(hashlib.sha256(gfa2).hexdigest() == gfa_sha)
        msa_second["timeline_s"] = res2["stats"].get("timeline_s")
        del res2, gfa2

    # ---- 2. the stitch batches of this rank's merges, resident in HBM ----------------------------------------------------------
    batches = stitch_batches(kept)
    stitch_sharding = None
    if world > 1 and not args.no_shard_stitch:
        # north_star: "the thousands of independent between-anchor stitch subproblems ... shard naturally across the GPUs".  A merge's batch is
        # made where the merge ran; for the timed passes every batch goes to every rank once (host arrays over the host group — set-up, outside
        # the timed region) and each rank keeps its share of EVERY batch: longest-processing-time assignment by DP cells (dist.shard_problems),
        # so the few long sweeps that bound a launch are spread over the devices.  No exchange in the timed steps.
        gathered = [None] * world
        dist.all_gather_object(gathered, batches, group=host_group)
        every = [mb for part in gathered for mb in part]
        batches, cells_all = [], 0
        for m, b in every:
            cells_all += b.dp_cells()
            mine = cd.shard_problems(b, world)[rank]
            if len(mine):
                batches.append((m, b.subset(mine)))
        stitch_sharding = {"batches": len(every), "dp_cells_all_ranks": int(cells_all), "rule": "LPT by (n1+1)(n2+1) over the ranks, per batch",
                           "applies_to": "the RESIDENT plans of the timed steps (a plan's share is fixed when it is made)"}
        # north_star's work stealing, as it would run on batches that are NOT resident: every rank holds every batch, the ranks' contexts form a group and pull
        # chunks of the LPT-ordered subproblem list (2 M cells each) from ONE atomic counter in rank 0's exported device memory (cl_context_peer_steal: system-scope
        # atomics over xGMI; no collective).  One untimed-for-the-headline pass, reported beside the static sharding; a failure never costs the line
        try:
            big = capi.StitchBatch.concat([b for _, b in every])
            handles = [None] * world
            dist.all_gather_object(handles, ctx.peer_export(), group=host_group)
            ctx.peer_group(handles, rank, 2000)
            dist.barrier(group=host_group)
            t0 = time.perf_counter()
            idx, res_steal, took = cd.stitch_by_stealing(ctx, big, lambda: ctx.peer_steal(1), chunk_cells=2000000)
            t_steal = cd.max_over_ranks(time.perf_counter() - t0, dist, device="cpu")
            parts = [None] * world
            dist.all_gather_object(parts, (len(took), int(sum(((np.diff(big.side[0].node_off).astype(np.int64)[idx] + 1) * (np.diff(big.side[1].node_off).astype(np.int64)[idx] + 1))))), group=host_group)
            stitch_sharding["work_stealing_pass"] = {"seconds_max_over_ranks": t_steal, "chunk_cells": 2000000, "chunks_and_cells_per_rank": parts,
                                                     "note": "plan + execute + collect per chunk (nothing resident): not comparable with ms_per_step; what it shows is the balance the one counter gives"}
            ctx.peer_group([], 0, 0)
        except Exception as e:   # noqa: BLE001
            stitch_sharding["work_stealing_pass"] = {"error": repr(e)[:300]}
    if args.debug_skip:
        os.environ["CL_DEBUG_SKIP_TRACEBACK"] = str(args.debug_skip)
    # one context (= one HIP stream set) per batch: the nine merges are independent, their passes run side by side on the device
    # exactly as the worker contexts of the MSA above run sibling merges side by side
    def make_plans(layout):
        if layout == "one" and batches:
            # every subproblem of the rank's merges in ONE batch: the launch groups (one per kernel shape) hold nine merges' worth of subproblems
            # each, so a step is a dozen large launches instead of ~110 small ones that queue behind one another on the hardware queues
            prev = os.environ.get("CL_CTX_STREAMS")
            os.environ["CL_CTX_STREAMS"] = prev or "8"      # eight streams for the sixteen launches of the one plan (measured: 6: 3.0-3.1, 8: 2.5-2.85, 10: 3.5, 12: 3.3 ms per step)
            c = capi.Context(device)
            if prev is None:
                del os.environ["CL_CTX_STREAMS"]
            return [c], [("all %d merges" % len(batches), c.plan(capi.StitchBatch.concat([b for _, b in batches])))]
        cs = [capi.Context(device) for _ in batches]
        return cs, [(m, c.plan(b)) for (m, b), c in zip(batches, cs)]

    def run_steps(plans, warmup, steps):
        def one_pass(wait):
            for _, p in plans:
                p.execute()           # the plan's launches, dealt over the context's streams by their measured durations (cl_stitch_plan_execute)
            return max([p.sync() for _, p in plans] or [0.0]) if wait else 0.0   # HIP events around each plan's pass; the passes overlap
        for _ in range(warmup):
            one_pass(True)
        barrier()
        t0 = time.perf_counter()
        # the K timed passes are ENQUEUED one behind the other and waited for once, by the barrier + synchronize that closes the timed region: no host round trip
        # between two passes, and (second half of round 5) no join of the plan's streams either — a launch group follows its own launch of the pass before in stream
        # order, the passes overlap; the trace showed 0.4-0.6 ms of idle device per 2.2 ms step while every stream waited for the slowest one and two event hops
        for _ in range(steps - 1):
            one_pass(False)
        dev = one_pass(True) * steps      # (HIP events from the last pass's enqueue to its end, as an indication: see "device_ms_from_the_last_pass_enqueue_to_its_end")
        barrier()
        return time.perf_counter() - t0, dev

    # the other layout first, untimed for the line: it is reported beside the headline (config.other_plan_layout)
    other = "nine" if args.plans == "one" else "one"
    other_ctx, other_plans = make_plans(other)
    other_elapsed, _ = run_steps(other_plans, min(args.warmup, 2), max(3, args.steps // 4))
    other_ms = cd.max_over_ranks(other_elapsed, dist, device="cpu" if share else "cuda") / max(3, args.steps // 4) * 1e3
    for _, p in other_plans:
        p.destroy()
    for c in other_ctx:
        c.close()
    plan_ctx, plans = make_plans(args.plans)
    stats = [p.stats() for _, p in plans]
    my_cells = sum(st["dp_cells"] for st in stats)
    elapsed, dev_ms = run_steps(plans, args.warmup, args.steps)
    elapsed = cd.max_over_ranks(elapsed, dist, device="cpu" if share else "cuda")
    # round-5 verdict, weak #7: the timed region is 20 x 1.5 ms with a run-to-run spread of +-5 %.  `value` stays the contract's ONE block of K steps (above); four more
    # blocks of K steps follow, and the line carries the minimum and the median of the five so that a reader sees where the headline block fell
    more_blocks = []
    for _ in range(4):
        e2, _ = run_steps(plans, 0, args.steps)
        more_blocks.append(cd.max_over_ranks(e2, dist, device="cpu" if share else "cuda") / args.steps * 1e3)
    # HIP events round every launch ON ITS STREAM inside a concurrent pass (cl_stitch_plan_execute_evented), with plain passes enqueued in front of and behind it so that
    # the evented pass overlaps its neighbours exactly as a timed pass does: the durations roofline.frac is priced with (they agree with the rocprofv3 kernel trace of the
    # step up to the stream's launch gap; the kernels' own clocks, which round 5 priced with, start at the first workgroup and miss the dispatch's wait for compute units
    # and the end-of-kernel write-back: profiles/r06_clock_gap.json)
    event_acc, event_n = {}, 0
    for _ in range(max(3, min(8, args.steps // 2))):
        for _, p in plans:
            p.execute(); p.execute()
        for _, p in plans:
            p.execute_evented()
        for _, p in plans:
            p.execute(); p.execute()
        for m, p in plans:
            p.sync()
            for li in p.launches():
                if li["event_ms"] > 0:
                    key = (m, li["kernel"], li["n_problems"], li["dp_cells"])
                    event_acc[key] = event_acc.get(key, 0.0) + li["event_ms"]
        event_n += 1
    for _, p in plans:        # one plain pass, waited for: the launches' own clocks of a pass WITHOUT events (in_pass_ms below)
        p.execute()
    for _, p in plans:
        p.sync()
    # every launch's own clock in the LAST timed pass (launches side by side, as in production): kernel name + subproblem count identify a launch
    in_pass = {}
    for m, p in plans:
        for li in p.launches():
            in_pass[(m, li["kernel"], li["n_problems"], li["dp_cells"])] = li["in_pass_ms"]
    # beside the line: the same plans with a join of each plan's streams after EVERY pass (how rounds 1-4 and the first half of round 5 timed the step), a few passes
    os.environ["CL_STITCH_JOIN"] = "eager"
    try:
        j_steps = max(3, args.steps // 4)
        j_elapsed, _ = run_steps(plans, 1, j_steps)
        join_per_pass_ms = cd.max_over_ranks(j_elapsed, dist, device="cpu" if share else "cuda") / j_steps * 1e3
    finally:
        del os.environ["CL_STITCH_JOIN"]
    total_cells = my_cells
    if dist is not None:
        t = torch.tensor([float(my_cells)], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t)
        total_cells = int(t.item())

    # per-kernel durations: the same passes launched kernel by kernel with HIP events on each launch's stream
    launches = []
    prof_steps = max(3, min(10, args.steps))
    for (m, p), st in zip(plans, stats):
        acc = {}
        for _ in range(prof_steps):
            p.execute_profiled()
            p.sync()
            for li in p.launches():
                e = acc.setdefault((li["kernel"], li["n_problems"], li["dp_cells"]), dict(li, ms=0.0, merge=m, in_pass_ms=in_pass.get((m, li["kernel"], li["n_problems"], li["dp_cells"]), 0.0),
                                                                                          event_ms=event_acc.get((m, li["kernel"], li["n_problems"], li["dp_cells"]), 0.0) / max(1, event_n)))
                e["ms"] += li["ms"]
        for e in acc.values():
            e["ms"] /= prof_steps
            launches.append(e)

    if rank == 0:
        value = total_cells * args.steps / elapsed
        # the dominant KERNEL = the kernel (by name) whose launches take the most time inside a timed pass; its roofline figure is the algorithmic bytes of
        # ALL its launches over the sum of their durations (= bytes per launch / average launch duration); the single longest launch of that kernel
        # carries the latency model.  (Rounds 1-3 and the first round-4 lines picked the single longest launch of any kernel: two launches of
        # different kernels last about the same, and which of them came out on top changed from run to run.)
        by_kernel = {}
        for e in launches:
            k = by_kernel.setdefault(e["kernel"], {"ms": 0.0, "bytes": 0, "cells": 0, "problems": 0, "launches": 0, "longest": e})
            dur = e.get("event_ms") or e.get("in_pass_ms") or e["ms"]
            k["ms"] += dur; k["bytes"] += e["dp_bytes"]; k["cells"] += e["dp_cells"]; k["problems"] += e["n_problems"]; k["launches"] += 1
            k["clock_ms"] = k.get("clock_ms", 0.0) + (e.get("in_pass_ms") or e["ms"])
            if dur > (k["longest"].get("event_ms") or k["longest"].get("in_pass_ms") or k["longest"]["ms"]):
                k["longest"] = e
        dom_name = max(by_kernel, key=lambda n: by_kernel[n]["ms"]) if by_kernel else None
        dom = by_kernel[dom_name]["longest"] if dom_name else None
        # PMC traffic cannot be collected inside this run (rocprofv3 --pmc is its own pass): what profiles/ holds is quoted with its
        # provenance and never divided by this run's times
        traffic_profile = None
        tp = os.path.join(HERE, "profiles", "hbm_traffic_latest.json")
        if dom is not None and os.path.exists(tp):
            try:
                with open(tp) as f:
                    tj = json.load(f)
                if tj.get(dom["kernel"]) is not None:
                    traffic_profile = {"file": "profiles/hbm_traffic_latest.json", "kernel": dom["kernel"],
                                       "hbm_bytes_per_launch_mean_over_all_launches_of_that_kernel": tj.get(dom["kernel"]),
                                       "collected_with": tj.get("_command", "scripts/pmc_round2.sh (separate --pmc passes, FETCH_SIZE x 2 + WRITE_SIZE corrections of MI355X_MICROARCH.md)"),
                                       "tree": tj.get("_commit", "round 2"), "limiter": tj.get("_limiter", {}).get(dom["kernel"])}
            except Exception:
                traffic_profile = None
        # the digest every run of the headline workload must print, whatever N: the GFA the unmodified REFERENCE prints for this configuration (its eight
        # subproblems in the build container, its root merge — 36 minutes, 70 GB — on a GPU box's host: tests/golden/c3_10x1M_subproblems.json)
        expected_sha = None
        try:
            with open(os.path.join(HERE, "tests", "golden", "c3_10x1M_subproblems.json")) as f:
                expected_sha = json.load(f)["root_default_budget_reference"]["sha256"]   # the unmodified reference's GFA of this very configuration
        except (OSError, KeyError, ValueError):
            pass
        gfa_ok = None if (expected_sha is None or args.length != 1000000) else bool(gfa_sha == expected_sha)
        per_merge = res["stats"].get("per_merge", [])
        chain_ms = sum(m["chain_device_ms"] for m in per_merge)
        chain_pairs = sum(m["chain_match_pairs"] for m in per_merge)
        out = {
            "metric": "stitcher PO-POA DP cells/s (10 x 1 Mbp synthetic HOR MSA, every between-anchor subproblem of all nine merges, fill + traceback)",
            "value": value, "unit": "DP cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "timed_blocks": {"ms_per_step": [elapsed / args.steps * 1e3] + more_blocks, "min": min([elapsed / args.steps * 1e3] + more_blocks),
                             "median": sorted([elapsed / args.steps * 1e3] + more_blocks)[2],
                             "note": "five timed blocks of K steps each, every one bracketed by barrier + synchronize; value / ms_per_step are the FIRST block (the contract's K steps), the others show the run-to-run spread"},
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("INVALID (--debug-skip %d): " % args.debug_skip if args.debug_skip else "") + ("INVALID (the GFA differs from the reference's): " if gfa_ok is False else "") + WORKLOAD if args.length == 1000000 else "DRY RUN at %d bp per sequence, not the headline: " % args.length + WORKLOAD,
                       "sequences": len(names), "sequence_length": args.length, "merges": len(per_merge) if world == 1 else 9,
                       "subproblems": int(sum(st["n_problems"] for st in stats)), "dp_cells": int(total_cells),
                       "msa_wall_s": msa_wall, "gfa_sha256": gfa_sha, "gfa_bytes": gfa_bytes, "gfa_is_the_references": gfa_ok,
                       "plan_layout": "%s resident plan%s per rank (--plans %s)" % ("one" if args.plans == "one" else "nine", "" if args.plans == "one" else "s", args.plans),
                       "other_plan_layout": {"layout": other, "ms_per_step": other_ms, "value": total_cells / (other_ms * 1e-3) if other_ms else None,
                                             "note": "the same subproblems, %s; rounds 1-3 reported the nine-plan layout" % ("one plan per merge, nine contexts side by side" if other == "nine" else "all in one plan")},
                       "stitch_sharding": stitch_sharding,
                       "workspace_bytes": int(sum(st.get("workspace_bytes", 0) for st in stats)),
                       "merge_groups": None if world == 1 else (res["stats"].get("merge_groups") if res is not None and "stats" in res else None),
                       "merge_groups_retry": merge_groups_retry,
                       # rank 0's share of the merges that ran as groups: chaining DPs shared, far launches on its combinations, macro-blocks whose other combinations came from the other members
                       "merge_group_stats": None if world == 1 else dict(ctx.peer_stats(), shared_merges=res["stats"].get("shared_merges", 0) if res is not None and "stats" in res else 0),
                       "parallelism": "1 GPU, %d worker contexts in the MSA" % args.workers if world == 1 else
                                      "one MSA over %d ranks (sibling subtrees + leaf calibrations per rank, %d worker contexts inside a rank; merges across ranks run as merge groups of up to %d ranks: every member runs the merge, the far pass of its chaining DP is divided between their devices through peer stores, no collective), stitch batches on the rank that made them; msa_wall_s stays bounded by the serial walk of the spine's merges (DESIGN.md §5); no multi-GPU hardware curve has been measured by the builder (one-device runs of the multi-rank path only)" % (world, args.workers, args.share_merges)},
            "msa_wall_s": msa_wall,
            "msa_wall_s_second_run": msa_second,
            "msa": {"timeline_s": res["stats"].get("timeline_s") if world == 1 else None, "pipeline": "leaf graphs + 10 calibrations + 9 x (find_matches + Core::align + fuse) + write_gfa, no reference in the loop",
                    "warmup_outside_the_timed_region": "find_matches on two 10-base graphs and one merge of two 3-kbp arrays on the main context (first use of every kernel's code object); the MSA's worker contexts are created inside the timed region",
                    "gfa_sha256": gfa_sha, "gfa_bytes": gfa_bytes, "score_scale": res["scale"],
                    "match_s": res["stats"]["match_ms"] / 1e3, "align_s_summed_over_contexts": res["stats"]["align_ms"] / 1e3,
                    "per_merge": [{k: m[k] for k in ("merge", "paths1", "paths2", "match_sets", "chain_match_pairs", "chain_combinations", "match_ms",
                                                     "align_ms", "chain_ms", "chain_device_ms", "partition_ms", "stitch_ms", "fuse_ms")} for m in per_merge]},
            # (up to the first half of round 5 a pass joined its context's stream before the next one forked from it, and this was the device time of the last pass;
            # now the passes of a resident plan overlap — a stream starts pass i + 1 when ITS launches of pass i are done (cl_stitch_join, cl_api.cpp) — and the events
            # round the last pass span everything that was still running when it was enqueued: CL_STITCH_JOIN=eager restores the join per pass)
            "device_ms_from_the_last_pass_enqueue_to_its_end": dev_ms / args.steps,
            "with_a_join_per_pass": {"ms_per_step": join_per_pass_ms, "value": total_cells / (join_per_pass_ms * 1e-3) if join_per_pass_ms else None,
                                     "note": "CL_STITCH_JOIN=eager: the context's stream waits for all eight streams after every pass, the next pass forks from it — the timing of rounds 1-4"},
            "passes": "the K timed passes are enqueued back to back; a launch group runs on the same stream in every pass, so stream order is the only order between passes and they overlap (no join of the eight streams between passes); the barrier + synchronize that closes the timed region waits for all of them",
            "launches": sorted(launches, key=lambda e: -e["ms"])[:12],
        }
        if dom is not None:
            dom_ms = dom.get("event_ms") or dom.get("in_pass_ms") or dom["ms"]      # inside a concurrent pass, HIP events on the launch's stream; `kernel_ms_alone` beside it
            agg = by_kernel[dom_name]
            achieved = agg["bytes"] / (agg["ms"] * 1e-3) / 1e9
            achieved_clock = agg["bytes"] / (agg["clock_ms"] * 1e-3) / 1e9 if agg.get("clock_ms") else None
            # latency model of the same launch: its duration is the dependent sweep of its longest subproblem (rows on lanes, one column per
            # step), so what can be acted on is the time per step against what ONE wave can issue: ~195 instructions per step in the systolic
            # DAG kernel, 4 cycles per 64-wide VALU instruction, 2.4 GHz
            steps = int(dom.get("max_sweep") or 0)
            issue_floor_ns = 195 * 4 / 2.4
            latency = None
            if steps:
                ns_per_step = dom_ms * 1e6 / steps
                latency = {"model": "latency", "dependent_steps": steps, "ns_per_step": ns_per_step, "single_wave_issue_floor_ns_per_step": issue_floor_ns,
                           "floor_over_measured": issue_floor_ns / ns_per_step, "longest_subproblem": dom.get("longest"),
                           "note": "kernel_ms: the dominant launch by the KERNEL'S OWN CLOCK — first workgroup's start to last workgroup's end in s_memrealtime ticks (100 MHz), "
                                   "recorded by every launch of every pass — in the LAST TIMED PASS, i.e. with the plan's other launches beside it, which is what the rocprofv3 "
                                   "kernel trace of a step shows (profiles/r04_*); kernel_ms_alone: the same clock with the launch alone on an otherwise idle device, where these "
                                   "latency-bound launches take about twice as long (the device does not run a lone small launch at full speed).  Rounds 1-3 reported HIP event pairs "
                                   "round the lone launch"}
            # traffic: HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command (rocprofv3 --pmc is a pass of its own and
            # cannot run inside this one); beside it the algorithmic bytes per launch of THIS run — their ratio is the re-read factor
            traffic = traffic_profile["hbm_bytes_per_launch_mean_over_all_launches_of_that_kernel"] if traffic_profile else None
            traffic_unit = "HBM bytes per launch (FETCH_SIZE and WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes), mean over the kernel's launches in profiles/hbm_traffic_latest.json (NOT launch-matched)"
            # launch-matched (round 5): profiles/dominant_launches_latest.json holds one row per launch of the timed step — scripts/dominant_launches.sh: kernel trace + FETCH_SIZE +
            # WRITE_SIZE passes over the step alone — and the rows of THIS run's launches of the dominant kernel (same kernel, subproblems and cells) give the traffic
            traffic_rows = None
            try:
                with open(os.path.join(HERE, "profiles", "dominant_launches_latest.json")) as f:
                    dl = json.load(f)
                mine = {(e["n_problems"], e["dp_cells"]) for e in launches if e["kernel"] == dom["kernel"]}
                rows = [r for r in dl["launches"] if r["kernel"] == dom["kernel"] and (r["subproblems"], r["dp_cells"]) in mine and "hbm_bytes" in r]
                if rows and len(rows) == len(mine):
                    traffic = sum(r["hbm_bytes"] for r in rows) / len(rows)
                    traffic_unit = "HBM bytes per launch: FETCH_SIZE x 2 + WRITE_SIZE (MI355X_MICROARCH.md) of the SAME launches (kernel, subproblems, cells matched) in profiles/dominant_launches_latest.json"
                    traffic_rows = {"file": "profiles/dominant_launches_latest.json", "command": dl.get("command"), "tree": dl.get("tree"),
                                    "rows": [{k: r.get(k) for k in ("subproblems", "dp_cells", "algorithmic_bytes", "hbm_bytes", "duration_us_mean", "hbm_over_algorithmic")} for r in rows]}
            except (OSError, KeyError, ValueError):
                pass
            # the same figure from the committed rocprofv3 kernel trace of the step (scripts/dominant_launches.sh -> profiles/dominant_launches_latest.json), launch-matched:
            # what a reader recomputes from profiles/ alone
            frac_profile = None
            if traffic_rows and all(r.get("duration_us_mean") for r in traffic_rows["rows"]):
                pb = sum(r["algorithmic_bytes"] for r in traffic_rows["rows"]); pt = sum(r["duration_us_mean"] for r in traffic_rows["rows"]) * 1e-6
                frac_profile = {"file": "profiles/dominant_launches_latest.json", "algorithmic_bytes": pb, "kernel_seconds": pt, "achieved_GB_per_s": pb / pt / 1e9, "frac": pb / pt / 8e12,
                                "note": "sum of algorithmic bytes / sum of rocprofv3 kernel-trace durations of the SAME launches (kernel, subproblems, cells) in the committed profile"}
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                               "durations_from": "HIP events round each launch on its own stream inside a concurrent pass of THIS run (cl_stitch_plan_execute_evented; plain passes in front and behind), mean of %d evented passes" % event_n,
                               "frac_kernel_clock": achieved_clock / 8000.0 if achieved_clock else None,
                               "frac_kernel_clock_note": "the same bytes over the kernels' own s_memrealtime clocks (first workgroup's start to last workgroup's end) in a plain concurrent pass: what round 5 printed as frac; it leaves out the dispatch's wait for compute units behind the other launches and the end-of-kernel write-back (profiles/r06_clock_gap.json)",
                               "frac_from_committed_profile": frac_profile,
                               "traffic_unit": traffic_unit, "traffic_launch_matched": traffic_rows,
                               "algorithmic_bytes_per_launch": agg["bytes"] / agg["launches"],
                               "kernel": dom["kernel"], "merge": dom["merge"], "kernel_launches": agg["launches"], "kernel_ms_all_launches": agg["ms"],
                               "kernel_ms_average": agg["ms"] / agg["launches"], "kernel_bytes_all_launches": agg["bytes"], "kernel_cells_all_launches": agg["cells"],
                               "kernel_problems_all_launches": agg["problems"],
                               "kernel_ms": dom_ms, "kernel_ms_alone": dom["ms"], "kernel_cells": dom["dp_cells"],
                               "kernel_problems": dom["n_problems"], "latency_model": latency, "traffic_profile": traffic_profile,
                               "kernels_by_time_in_pass": sorted(({"kernel": n, "ms": v["ms"], "launches": v["launches"], "GB_per_s": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else None}
                                                                  for n, v in by_kernel.items()), key=lambda x: -x["ms"])[:6],
                               "limiter": "dependent chain of the largest matrix (n1 + n2 steps of one workgroup), not HBM: see latency_model",
                               "note": "achieved = ALGORITHMIC bytes (sizeof(cell_t<NumPW>) x cells, SURVEY.md §8d) of all launches of the dominant kernel / the sum of their durations by HIP events on the launches' streams inside concurrent passes of THIS run (kernel_ms, kernel_cells: its longest launch); "
                                       "traffic: HBM bytes per launch of the SAME launches (matched by kernel, subproblems, cells) from the committed per-launch profile (traffic_launch_matched / traffic_profile give its provenance), never divided by this run's times"}
        if elapsed > 0:
            # the whole step against the same roofline: what the plan's launches together stream per second if every cell's state moved once
            step_bytes = float(sum(st["dp_bytes"] for st in stats))
            if dist is not None:
                step_bytes *= total_cells / max(1.0, float(my_cells))   # (other ranks' batches: same bytes per cell on average)
            ach = step_bytes * args.steps / elapsed / 1e9
            out["roofline_pass"] = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                                    "note": "ALGORITHMIC bytes of all launches of a step / the step's wall-clock: the plan's launches run side by side on eight streams (and consecutive passes "
                                            "overlap), so the pass as a whole sits much closer to the roofline than its dominant kernel's launches, which `roofline` reports"}
        if chain_ms > 0:
            n_macro = sum(2 * -(-int(m["chain_match_pairs"]) // 1024) for m in per_merge)     # two whole-graph DPs per merge, 1024 pairs per macro-block
            out["chain_dp"] = {"model": "latency", "not_a_roofline": True, "device_ms": chain_ms, "match_pairs": int(chain_pairs), "macro_blocks": n_macro,
                               "us_per_macro_block": chain_ms * 1e3 / max(1, n_macro),
                               "kernels": "chain_walk2_kernel -> chain_inter_kernel (near) -> far_prune_kernel / far_seal(_big)_kernel, both whole-graph DPs of all nine merges",
                               "note": "the chaining DP is a chain of ~1 200 dependent macro-blocks per DP (walk + near pass on the serial stream, far pass two blocks "
                                       "behind on side streams); its kernels are priced per kernel against HBM bytes/s and VALU issue in profiles/r04_pmc_summary.json "
                                       "(rocprofv3 --pmc passes; command recorded in the file), not here: the far pass skips >98 % of the pair evaluations an "
                                       "all-pairs sweep would make, so an evaluations/s figure says nothing about the hardware"}
            # the chaining kernels against the hardware's bounds (last bullet; unit = match pair): the committed rocprofv3 passes of the chaining seam
            # (scripts/chain_roofline.sh: kernel trace + FETCH_SIZE + WRITE_SIZE + SQ counters over scripts/anchor_bench.py) per kernel — HBM bytes/s against 8 TB/s and
            # VALU wave-instructions/s against the chip's issue capacity — quoted with their provenance, never divided by this run's times
            try:
                with open(os.path.join(HERE, "profiles", "chain_roofline_latest.json")) as f:
                    cr = json.load(f)
                keep = ("kernel", "calls", "avg_us", "share_pct", "hbm_bytes_per_call", "hbm_bytes_per_match_pair", "hbm_GB_per_s", "hbm_frac", "valu_wave_insts_per_call",
                        "valu_wave_insts_per_match_pair", "valu_G_per_s", "valu_issue_frac", "wave_cycles_issuing_valu", "wave_cycles_parked_on_waitcnt_or_barrier", "bound")
                out["roofline_chain"] = {"file": "profiles/chain_roofline_latest.json", "command": cr.get("command"), "tree": cr.get("tree"), "peaks": cr.get("peaks"), "unit": cr.get("unit"),
                                         "kernels": [{k: e.get(k) for k in keep} for e in cr.get("kernels", [])[:8]],
                                         "this_run": {"chain_device_ms_all_merges": chain_ms, "match_pairs_all_merges": int(chain_pairs), "us_per_macro_block": chain_ms * 1e3 / max(1, n_macro)}}
            except (OSError, KeyError, ValueError):
                out["roofline_chain"] = None
        # in situ (round-5 verdict, weak #3): what ONE production pass of the stitcher costs inside the MSA — cl_core_align's stitch_ms = extraction + rank-space packing + upload +
        # DP + collect + translate — against the resident plan's timed step; and the same batches one-shot on an otherwise idle device (cl_stitch_batch_align: plan + H2D + DP + D2H)
        try:
            situ_ms = sum(m["stitch_ms"] for m in per_merge)
            situ_cells = sum(b.dp_cells() for _, b in batches) if world == 1 else None
            one_shot = None
            if world == 1:
                t_os = time.perf_counter()
                for _, b in batches:
                    ctx.stitch_batch_align(b)
                one_shot = time.perf_counter() - t_os
            out["in_situ"] = {"stitch_ms_summed_over_merges": situ_ms, "dp_cells": situ_cells,
                              "cells_per_s_inside_the_msa": (situ_cells / (situ_ms * 1e-3)) if situ_cells and situ_ms else None,
                              "one_shot_s_all_batches": one_shot, "cells_per_s_one_shot": (situ_cells / one_shot) if situ_cells and one_shot else None,
                              "note": "stitch_ms (cl_align_api.cpp) = subgraph extraction + rank-space packing + H2D + fill + traceback + D2H + translate, per merge, with the MSA's other worker contexts beside it; "
                                      "one_shot = cl_stitch_batch_align per batch (plan create + upload + execute + collect + translate) on an idle device, batches already extracted; `value` is the resident plan re-executed (inputs in HBM, K passes overlapping)"}
        except Exception as e:   # noqa: BLE001
            out["in_situ"] = {"error": repr(e)[:200]}
        # BASELINE.md §2 holds ONE figure for this metric, measured (not published) by the survey: the reference's po_poa at 38 M cells/s inside
        # its 2 x 1 Mbp run on one Xeon core; the same-host figures are in cpu_baseline (measured here, every run)
        # ... which is another configuration (2 x 1 Mbp) and not a published number: vs_baseline stays null (BASELINE.json `published` is {}), the ratio is printed under its own name
        out["vs_baseline"] = None
        out["vs_survey_cpu_rate"] = value / 38.0e6
        out["vs_baseline_note"] = ("vs_baseline is null: BASELINE.md publishes no number for this metric.  vs_survey_cpu_rate = value / 38 M cells/s, the reference's stitching rate the survey "
                                   "measured itself (BASELINE.md §2: 2 x 1 Mbp, one Xeon core); the same-host figure is cpu_baseline / vs_cpu_baseline_same_host")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batches)
            out["vs_cpu_baseline_same_host"] = value / out["cpu_baseline"]["value"]
            out["cpu_reference_wall"] = reference_leaf_merge(seqs, names)
            if "seconds" in out["cpu_reference_wall"]:
                out["cpu_reference_wall"]["whole_msa_here_s"] = msa_wall
                out["cpu_reference_wall"]["note"] = ("the reference needs %.1f s for the FIRST of the nine merges on this host; the whole ten-sequence MSA takes %.2f s here "
                                                     "(the reference needs 37.6 CPU-minutes for the eight lower merges and 36.1 for the root at the default budget, on two hosts: profiles/r04_c3_root_default_reference.json; its GFA is the one printed here, byte for byte)" % (out["cpu_reference_wall"]["seconds"], msa_wall))
        if world == 1 and not args.no_extras:
            out["chaining_c2"] = chaining_section(ctx, not args.no_cpu_baseline)
            out["pairwise_c2"] = pairwise_section(ctx, not args.no_cpu_baseline)
            out["big_dag_pair"] = big_dag_section(ctx)
        print(json.dumps(out))
    barrier()
    for _, p in plans:
        p.destroy()
    for c in plan_ctx:
        c.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
