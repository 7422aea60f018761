"""time-bounded randomised parity campaign on the GPU: the HIP paths against the CPU oracle (oracle/ — test infrastructure) on inputs the fixed tests do not hold.
  stitch: batches of chain pairs (tiny .. 5 000 rows: every chain kernel, the workgroup-spanning one included), near-chains, random DAGs, large branching pairs with far
          forks (strips), NumPW as the library picks it or forced 1 / 2 / 3 — aligned pairs, scores, routes against oracle/popoa_oracle.c;
  chain : two multi-path graphs out of the library's own merges of random HOR sequences, a random subset of their match sets, the affine and the gap-free DP, local and
          global anchoring, random scale — DP values as bit patterns and the chain against oracle/chain_oracle.cpp.
usage: python3 scripts/fuzz_gpu.py [--seconds T] [--seed S] [--what stitch,chain] [--json OUT]; exit code 1 on any mismatch (each printed with its seed)"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from centrolign_amd import capi, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def stitch_case(rng):
    kind = rng.choice(["linear_small", "linear_mixed", "linear_big", "near_chain", "dag_small", "dag_sized", "far_fork"], p=[0.12, 0.18, 0.12, 0.2, 0.18, 0.12, 0.08])
    seed = int(rng.integers(0, 1 << 30))
    if kind == "linear_small":
        sizes = [(int(rng.integers(1, 70)), int(rng.integers(1, 70))) for _ in range(int(rng.integers(1, 400)))]
        b = synth.linear_batch(sizes, seed=seed, divergence=float(rng.choice([0.0, 0.05, 0.3])))
    elif kind == "linear_mixed":
        sizes = [(int(rng.integers(1, 1300)), int(rng.integers(1, 1300))) for _ in range(int(rng.integers(1, 40)))]
        b = synth.linear_batch(sizes, seed=seed, divergence=float(rng.choice([0.01, 0.1, 0.4])))
    elif kind == "linear_big":
        sizes = []
        for _ in range(int(rng.integers(1, 4))):
            a, c = int(rng.integers(1025, 5000)), int(rng.integers(200, 5000))
            sizes.append((a, c) if rng.random() < 0.5 else (c, a))
        b = synth.linear_batch(sizes, seed=seed, divergence=float(rng.choice([0.01, 0.1, 0.4])))
    elif kind == "near_chain":
        sizes = [(int(rng.integers(20, 900)), int(rng.integers(20, 900))) for _ in range(int(rng.integers(1, 30)))]
        b = synth.near_chain_batch(sizes, seed=seed, p_snp=float(rng.choice([0.0, 0.04, 0.15])), p_del=float(rng.choice([0.0, 0.03, 0.1])),
                                   n_long=(0, int(rng.integers(0, 3))), long_min=int(rng.choice([70, 150])), long_max=400, related=bool(rng.random() < 0.7))
    elif kind == "dag_small":
        b = synth.random_dag_batch(int(rng.integers(1, 300)), seed=seed, max_n=int(rng.choice([6, 20, 40, 90])), extra_edge_p=float(rng.choice([0.0, 0.1, 0.3, 0.6])),
                                   skip_max=int(rng.choice([1, 2, 4, 8])), alphabet=int(rng.integers(2, 5)))
    elif kind == "dag_sized":
        sizes = [(int(rng.integers(64, 1500)), int(rng.integers(64, 1500))) for _ in range(int(rng.integers(1, 6)))]
        b = synth.sized_dag_batch(sizes, seed=seed, extra_edge_p=float(rng.choice([0.0, 0.02, 0.1, 0.3])), skip_max=int(rng.choice([1, 2, 4, 8, 20, 62])),
                                  n_alt=int(rng.integers(0, 4)), alphabet=int(rng.integers(2, 5)))
    else:
        sizes = []
        for _ in range(int(rng.integers(1, 3))):
            a, c = int(rng.integers(300, 2600)), int(rng.integers(1200, 4200))
            sizes.append((a, c) if rng.random() < 0.6 else (c, a))
        b = synth.far_fork_batch(sizes, seed=seed, n_far=int(rng.integers(1, 10)), far_min=int(rng.choice([70, 200, 600])), far_max=2000)
    npw = int(rng.choice([0, 0, 1, 2, 3]))
    return kind, seed, b, (None if npw == 0 else np.full(b.n_problems, npw, np.uint8)), npw


def fuzz_stitch(ctx, rng, seconds, log):
    seen, n, bad, cells = collections.Counter(), 0, 0, 0
    t_end = time.time() + seconds
    while time.time() < t_end:
        kind, seed, b, force, npw = stitch_case(rng)
        plan = ctx.plan(b, force_num_pw=force)
        for li in plan.launches():
            seen[li["kernel"].split(" x ")[0]] += 1
        plan.execute(); plan.sync()
        got = plan.collect()
        cells += int(plan.stats()["dp_cells"])
        plan.destroy()
        want = po.oracle_stitch_batch(b, force_num_pw=force)
        diff = got.same_as(want)
        n += 1
        if diff is not None:
            bad += 1
            log("MISMATCH stitch kind %s seed %d npw %d problems %d: %s" % (kind, seed, npw, b.n_problems, diff))
    return dict(cases=n, mismatches=bad, dp_cells=cells, launches_by_kernel=dict(seen))


def relabelled(g, src_label, snk_label):
    lab = g.label.copy()
    lab[g.src_id], lab[g.snk_id] = src_label, snk_label
    return capi.BaseGraph(lab, g.next_off, g.next_idx, g.prev_off, g.prev_idx, g.path_off, g.path_nodes, g.src_id, g.snk_id)


def merged(ctx, leaves, scale):
    g = leaves[0]
    for h in leaves[1:]:
        g = ctx.merge(g, h, score_scale=scale)["fused"]
    return g


def fuzz_chain(ctx, rng, seconds, log):
    n, bad, pairs, shapes = 0, 0, 0, collections.Counter()
    t_end = time.time() + seconds
    while time.time() < t_end:
        seed = int(rng.integers(0, 1 << 30))
        n1, n2 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        length = int(rng.choice([8000, 12000, 20000, 40000]))
        seqs = synth.hor_sequences(seed, length, n1 + n2, seq_div=float(rng.choice([0.002, 0.01, 0.03])))
        if min(len(s) for s in seqs) < 200:
            continue
        leaves = [capi.leaf_graph(s) for s in seqs]
        scale0 = 0.8
        g1, g2 = relabelled(merged(ctx, leaves[:n1], scale0), 5, 6), relabelled(merged(ctx, leaves[n1:], scale0), 7, 8)
        full = ctx.find_matches(g1, g2, max_count=int(rng.choice([50, 300, 3000])))
        if full.n_sets == 0:
            continue
        ms = po.budget_subset(full, int(rng.choice([300, 3000, 20000, 60000])), seed=seed)
        if ms.n_sets == 0:
            continue
        for algo in ("affine", "sparse"):
            glob = bool(rng.random() < 0.5)
            scale = float(rng.choice([1.0, 0.6, 0.05]))
            kw = dict(want_dp=True, params=capi.default_chain_params(global_anchoring=glob))
            if algo == "affine":
                got = ctx.chain_sparse_affine(g1, g2, ms, scale=scale, **kw)
                want_chain, want_dp = po.oracle_chain("affine", g1, g2, ms, scale=scale, want_dp=True, global_anchoring=glob)
            else:
                got = ctx.chain_sparse_affine(g1, g2, ms, sparse=True, **kw)
                want_chain, want_dp = po.oracle_chain("sparse", g1, g2, ms, want_dp=True, global_anchoring=glob)
            n += 1
            pairs += int(ms.n_pairs())
            shapes["%d x %d paths" % (n1, n2)] += 1
            ok_dp = np.array_equal(got["dp"].view(np.uint32), want_dp[:len(got["dp"])].view(np.uint32))
            ok_chain = np.array_equal(got["chain"], want_chain)
            if not (ok_dp and ok_chain):
                bad += 1
                log("MISMATCH chain %s seed %d paths %d+%d length %d pairs %d global %d scale %g: dp %s chain %s" % (
                    algo, seed, n1, n2, length, ms.n_pairs(), glob, scale, "same" if ok_dp else "DIFFERS", "same" if ok_chain else "DIFFERS"))
    return dict(cases=n, mismatches=bad, match_pairs=pairs, cases_by_shape=dict(shapes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0, help="per kind")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--what", default="stitch,chain")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    lines = []

    def log(s):
        lines.append(s)
        print(s, flush=True)
    ctx = capi.Context(0)
    out = dict(seed=args.seed, seconds_per_kind=args.seconds)
    for what in args.what.split(","):
        rng = np.random.default_rng([args.seed, len(what)])
        out[what] = (fuzz_stitch if what == "stitch" else fuzz_chain)(ctx, rng, args.seconds, log)
        print(what, json.dumps(out[what]), flush=True)
    out["mismatch_lines"] = lines
    out["fallbacks"] = capi.fallback_counters() if hasattr(capi, "fallback_counters") else None
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()
    sys.exit(1 if any(out[w]["mismatches"] for w in args.what.split(",")) else 0)


if __name__ == "__main__":
    main()
