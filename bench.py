#!/usr/bin/env python3
"""bench.py — stitch-path throughput on MI355X (driver contract: see the task statement / DESIGN.md §Measurement).

One "step" = one pass of the hot path (batched PO-POA fill + traceback of every between-anchor
subproblem) over the stitch batch of BASELINE.json configs[1]: the pairwise 2 x 1 Mbp synthetic HOR
centromere, seed 7 — exactly the 13 245 subproblems / 42 416 142 DP cells the reference extracts for that pair
(tests/golden/c2_pair_seed7_intervals.npz holds the subproblem intervals dumped from the reference; the two
sequences are regenerated from the seed).  Inputs are resident in HBM before the timed region starts.

N > 1: one process per GPU (torchrun), every rank stitches one such pairwise batch (independent sibling merges
of a guide tree shard with no data-path exchange) => weak scaling, no collective in the timed region except
the bracketing barriers.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402


def build_workload():
    from centrolign_amd import synth
    z = np.load(os.path.join(HERE, "tests", "golden", "c2_pair_seed7_intervals.npz"))
    seqs = synth.hor_sequences(7, 1000000, 2)
    return synth.batch_from_intervals(seqs[0], seqs[1], z["intervals"], z["only_del"])


def cpu_baseline(batch, min_seconds=10.0, max_reps=12):
    """single-thread CPU time of the same stitch batch: the compiled reference when oracle/_ref travelled
    with the repo ("reference"), else our C restatement ("port")"""
    from oracle import pyoracle as po
    cells = batch.dp_cells()
    kind = "reference" if po.have_ref() else "port"
    total, reps = 0.0, 0
    while total < min_seconds and reps < max_reps:
        if kind == "reference":
            _, secs = po.ref_stitch_batch(batch)
        else:
            t0 = time.perf_counter()
            po.oracle_stitch_batch(batch)
            secs = time.perf_counter() - t0
        total += secs
        reps += 1
    return {"value": cells * reps / total, "unit": "DP cells/s", "cores": 1, "kind": kind,
            "sample": "the whole 2x1Mbp stitch batch (13245 subproblems, %d cells) x %d passes, %.1f s of CPU, "
                      "time inside Stitcher::subalign only" % (cells, reps, total)}


def saturated_section(ctx, batch, copies=16, steps=10):
    """kernel throughput when the device is filled: the same batch replicated `copies` times in one plan (what a
    guide tree with that many sibling merges in flight on one GPU would submit).  Reported beside the single-batch
    pass, whose duration is set by the sweep of its few largest matrices rather than by throughput."""
    from centrolign_amd import capi
    big = capi.StitchBatch.concat([batch] * copies)
    plan = ctx.plan(big)
    st = plan.stats()
    for _ in range(2):
        plan.execute()
        plan.sync()
    ms = 0.0
    for _ in range(steps):
        plan.execute()
        ms += plan.sync()
    ms /= steps
    plan.destroy()
    return {"copies": copies, "subproblems": st["n_problems"], "dp_cells": st["dp_cells"], "device_ms_per_pass": ms,
            "cells_per_s": st["dp_cells"] / (ms * 1e-3), "algorithmic_GBps": st["dp_bytes"] / (ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": st["dp_bytes"] / (ms * 1e-3) / 1e9 / 8000.0}


def pipelined_section(batch, local_rank, depth=6, steps=48):
    """the same pass with `depth` independent passes in flight (one context + plan each, as the sibling merges of a guide
    tree provide them): fills the CUs that idle while the longest subproblem of a single batch finishes its sweep"""
    from centrolign_amd import capi
    ctxs = [capi.Context(local_rank) for _ in range(depth)]
    plans = [c.plan(batch) for c in ctxs]
    cells = plans[0].stats()["dp_cells"]
    for p in plans:
        p.execute(); p.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        plans[k % depth].execute()
    for p in plans:
        p.sync()
    dt = time.perf_counter() - t0
    for p in plans:
        p.destroy()
    for c in ctxs:
        c.close()
    return {"passes_in_flight": depth, "steps": steps, "ms_per_step": dt / steps * 1e3, "cells_per_s": cells * steps / dt}


def chaining_section(ctx, with_reference):
    """second half of the hot path, at its seam S3: Anchorer::anchor_chain in the CLI's default configuration (branch
    splitting, scale estimate by sparse_chain_dp, sparse_affine_chain_dp, fill-in re-anchoring, global anchoring) on the
    match sets of the same 2 x 1 Mbp pair (bench_data/c2_chain_input.npz: the reference's graphs and match sets; built by
    scripts/chain_bench.py's recipe in the build container, shipped with the repo snapshot, not committed)"""
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
        # no fixture on this box: the same inputs made here — the pair's leaf graphs, its match sets by cl_find_matches (identical
        # to the reference's, see match_finding) and the calibrated scale by cl_leaf_intrinsic_scale
        seqs = synth.hor_sequences(7, 1000000, 2)
        graphs = [synth.base_graph_from_sequence(seqs[0]), synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8))]
        ms = ctx.find_matches(graphs[0], graphs[1])
        scales = [ctx.leaf_intrinsic_scale(g) for g in graphs]
        score_scale = sum(scales) / len(scales)
        source = "made in place: cl_find_matches + cl_leaf_intrinsic_scale on the synthetic pair"

    def run():
        t0 = time.perf_counter()
        split = capi.split_branching_matches(graphs[0], graphs[1], ms)
        got = ctx.anchor_chain(graphs[0], graphs[1], split, score_scale=score_scale)
        return got, time.perf_counter() - t0
    run()  # warm-up
    got, wall = run()
    n = ms.n_pairs()
    out = {"seam": "Anchorer::anchor_chain (default configuration)", "input": source, "match_sets": ms.n_sets, "match_pairs": n,
           "chain_anchors": int(len(got["chain"])), "estimated_scale": got["scale"], "tie_resolutions": got["n_ties"],
           "fill_in_pairs": got["fill_in_pairs"], "max_num_match_pairs": 1250000, "wall_s": wall,
           "match_pairs_per_s": min(n, 1250000) / wall}
    # the inner DP alone (sparse_affine_chain_dp on every pair), with its device / host split
    if n <= 1300000:   # the fixture is already cut to the CLI's budget of 1.25 M pairs; the unbudgeted sets are several times that
        dp = ctx.chain_sparse_affine(graphs[0], graphs[1], ms, scale=got["scale"])
        out["affine_dp_only"] = {"match_pairs": dp["n_pairs"], "device_dp_ms": dp["device_ms"], "host_prep_ms": dp["prep_ms"],
                                 "value_index_ms": dp["index_ms"], "traceback_ms": dp["traceback_ms"],
                                 "pair_evaluations_per_s_device": dp["n_pairs"] ** 2 / 2 / (dp["device_ms"] * 1e-3)}
    # the whole merge: Core::align = anchor chain + partition + despecify + stitch (cl_core_align), same input
    t0 = time.perf_counter()
    al = ctx.core_align(graphs[0], graphs[1], ms, score_scale=score_scale)
    out["core_align"] = {"wall_s": time.perf_counter() - t0, "chain_ms": al["chain_ms"], "partition_ms": al["partition_ms"],
                         "stitch_ms": al["stitch_ms"], "segments": int(len(al["seg_off"]) - 1), "anchors": int(len(al["walk_off"]) - 1),
                         "aligned_pairs": int(len(al["alignment"]))}
    if with_reference:
        from oracle import pyoracle as po
        if po.have_ref():
            t0 = time.perf_counter()
            split = po.ref_split_branching_matches(graphs[0], graphs[1], ms)
            ref = po.ref_anchor_chain(graphs[0], graphs[1], split, score_scale=score_scale, fill_in=True)
            secs = time.perf_counter() - t0
            same = all(np.array_equal(ref[k], got[k]) for k in ("set_order", "chain", "walk1", "walk2", "gap_before", "gap_after",
                                                                 "gap_score_before", "gap_score_after", "score"))
            out["cpu_reference"] = {"seconds": secs, "match_pairs_per_s": min(n, 1250000) / secs, "cores": 1, "kind": "reference",
                                    "identical_result": bool(same and ref["scale"] == got["scale"])}
    return out


def match_section(ctx, with_reference):
    """the row before the hot path: PathMatchFinder::find_matches on the same 2 x 1 Mbp pair — suffix
    array + LCP on the device, the minimal-rare-match query on the host — checked against the digest of the compiled
    reference's output (tests/golden/match_finder.npz "c2.*") and timed beside the reference where oracle/_ref exists"""
    from centrolign_amd import capi, synth
    from tests import helpers as H
    seqs = synth.hor_sequences(7, 1000000, 2)
    g1 = synth.base_graph_from_sequence(seqs[0])
    g2 = synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8))
    ctx.find_matches(g1, g2)   # warm-up
    t0 = time.perf_counter()
    ms, st = ctx.find_matches(g1, g2, want_stats=True)
    wall = time.perf_counter() - t0
    z = np.load(os.path.join(H.GOLDEN, "match_finder.npz"))
    out = {"seam": "PathMatchFinder::find_matches", "text_length": st["text_length"], "match_sets": ms.n_sets, "wall_s": wall,
           "device_suffix_array_ms": st["sa_ms"], "device_lcp_ms": st["lcp_ms"], "doubling_rounds": st["doubling_rounds"],
           "host_tree_ms": st["tree_ms"], "host_query_ms": st["query_ms"], "host_walk_out_ms": st["walk_ms"],
           "lcp_intervals": st["n_internal_nodes"],
           "identical_to_reference_digest": bool(ms.n_sets == int(z["c2.n_sets"][0]) and H.match_sets_digest(ms) == str(z["c2.digest"][0]))}
    if with_reference:
        from oracle import pyoracle as po
        if po.have_ref():
            t0 = time.perf_counter()
            po.ref_find_matches(g1, g2, max_count=3000)
            out["cpu_reference"] = {"seconds": time.perf_counter() - t0, "cores": 1, "kind": "reference",
                                    "note": "includes flattening the reference's vectors into numpy"}
    return out


def end_to_end_section(ctx, with_reference):
    """BASELINE configs[1] as the CLI runs it, without the reference in the loop: FASTA-level sequences -> leaf graphs ->
    calibration -> match finding -> Core::align -> fuse -> explicit CIGAR (centrolign_amd/msa.py over the C ABI), wall-clock;
    beside it the compiled reference's whole pipeline on the same sequences (ref_msa_dump = Core::execute + explicit_cigar) and a
    byte comparison of the two outputs"""
    import hashlib
    import tempfile
    from centrolign_amd import msa, synth
    seqs = synth.hor_sequences(7, 1000000, 2)
    names = ["seq0", "seq1"]
    t0 = time.perf_counter()
    r = msa.progressive_msa(ctx, dict(zip(names, seqs)), msa.balanced_tree(names), workers=2)   # the two leaf calibrations side by side
    text = msa.output_text(r)
    wall = time.perf_counter() - t0
    out = {"pipeline": "leaf graphs + calibration (2 worker contexts) + find_matches + Core::align + fuse + explicit_cigar", "wall_s": wall,
           "score_scale": r["scale"], "match_ms": r["stats"]["match_ms"], "align_ms": r["stats"]["align_ms"], "fuse_ms": r["stats"]["fuse_ms"],
           "cigar_bytes": len(text), "cigar_sha256": hashlib.sha256(text).hexdigest()}
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
            out["cpu_reference"] = {"seconds": secs, "cores": 1, "kind": "reference", "phases_s": tm, "identical_output": bool(want == text)}
    return out


def main():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before HIP initialises: independent passes overlap on separate queues
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed passes (no saturated / pipelined / chaining sections): the command the rocprofv3 "
                         "summaries under profiles/ are taken with, so that their per-kernel averages are those of the timed passes")
    args = ap.parse_args()

    import torch
    from centrolign_amd import dist as cd
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank, world, dist = cd.init_distributed("nccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from centrolign_amd import capi
    ctx = capi.Context(local_rank)          # raises without a GPU / without the HIP library: no fallback
    batch = build_workload()
    plan = ctx.plan(batch)
    stats = plan.stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.execute()
        plan.sync()
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        plan.execute()            # replays the captured hipGraph of all kernel launches of the pass
        dev_ms += plan.sync()     # HIP events around the pass on the context's stream
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel durations: the same pass launched kernel by kernel with HIP events on each launch's stream
    launch_ms = {}
    prof_steps = max(5, min(20, args.steps))
    for _ in range(prof_steps):
        plan.execute_profiled()
        plan.sync()
        for li in plan.launches():
            e = launch_ms.setdefault(li["kernel"], dict(li, ms=0.0))
            e["ms"] += li["ms"]
    elapsed = cd.max_over_ranks(elapsed, dist, device="cuda")

    if rank == 0:
        cells = stats["dp_cells"]
        value = cells * args.steps * world / elapsed
        for e in launch_ms.values():
            e["ms"] /= prof_steps
        dom = max(launch_ms.values(), key=lambda e: e["ms"])
        peak = 8000.0
        achieved = dom["dp_bytes"] / (dom["ms"] * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(HERE, "profiles", "hbm_traffic_latest.json")
        if os.path.exists(tp):
            try:
                with open(tp) as f:
                    traffic = json.load(f).get(dom["kernel"])
            except Exception:
                traffic = None
        out = {
            "metric": "stitcher PO-POA DP cells/s (2x1Mbp synthetic HOR pair, all between-anchor subproblems, fill + traceback)",
            "value": value, "unit": "DP cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: pairwise 1 Mbp x 1 Mbp synthetic centromere (seed 7), "
                                   "stitch batch = %d subproblems / %d DP cells per GPU" % (stats["n_problems"], cells),
                       "subproblems": stats["n_problems"], "dp_cells": cells, "chain_problems": stats["n_linear"],
                       "parallelism": "1 stitch batch per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": traffic, "kernel": dom["kernel"], "kernel_ms": dom["ms"],
                         "kernel_cells": dom["dp_cells"], "kernel_problems": dom["n_problems"],
                         "note": "achieved = sizeof(cell_t<NumPW>) x cells of this launch / its HIP-event duration; "
                                 "scores stay in registers and only 1-2 B/cell traceback codes reach HBM, so traffic << algorithmic bytes"},
            "device_ms_per_step": dev_ms / args.steps,
            "whole_pass": {"algorithmic_GBps": stats["dp_bytes"] / (dev_ms / args.steps * 1e-3) / 1e9,
                           "cells_per_s_device": cells / (dev_ms / args.steps * 1e-3)},
            "launches": sorted(launch_ms.values(), key=lambda e: -e["ms"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch)
        if world == 1 and not args.no_extras:
            out["saturated"] = saturated_section(ctx, batch)
            out["pipelined"] = pipelined_section(batch, local_rank)
            ch = chaining_section(ctx, not args.no_cpu_baseline)
            if ch is not None:
                out["chaining"] = ch
            out["match_finding"] = match_section(ctx, not args.no_cpu_baseline)
            out["end_to_end"] = end_to_end_section(ctx, not args.no_cpu_baseline)
        print(json.dumps(out))
    barrier()
    plan.destroy()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
