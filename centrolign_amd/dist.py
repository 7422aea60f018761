"""Multi-GPU plumbing for the stitch path: one process per GPU (torch.distributed, RCCL on GPUs, gloo in CPU tests).

The path shards by subproblem with no data exchange, so the only collectives are the barrier that
brackets a timed region, a MAX over ranks of the elapsed time, and — for callers that split ONE batch over ranks —
a gather of the per-rank results on rank 0.
"""
import os

import numpy as np


def init_distributed(backend=None):
    """returns (rank, world, dist or None); reads RANK / WORLD_SIZE / MASTER_* from the environment"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, dist


def max_over_ranks(value, dist, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_problems(batch, world):
    """Longest-processing-time assignment of the subproblems of ONE batch to `world` ranks by DP cells
    ((n1+1)(n2+1)); returns a list of index arrays, each in ascending problem order."""
    n1, n2 = batch.sizes()
    cells = (n1 + 1) * (n2 + 1)
    order = np.argsort(-cells, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    bins = [[] for _ in range(world)]
    for k in order:
        r = int(np.argmin(load))
        bins[r].append(int(k))
        load[r] += int(cells[k])
    return [np.array(sorted(b), dtype=np.int64) for b in bins]


def exchange_objects(obj, members, rank, dist, group=None):
    """every member of `members` (ranks of the host group, the same list on each) gets every member's object, in the list's order — point to point, no collective
    (a subset of the ranks must not need a process group of its own): pairs meet in rank order, the lower rank sends first"""
    import pickle
    import torch
    out = []
    blob = torch.from_numpy(np.frombuffer(pickle.dumps(obj, protocol=4), np.uint8).copy())

    def send(dst):
        dist.send(torch.tensor([blob.numel()], dtype=torch.int64), dst, group=group)
        dist.send(blob, dst, group=group)

    def recv(src):
        n = torch.zeros(1, dtype=torch.int64)
        dist.recv(n, src, group=group)
        buf = torch.zeros(int(n.item()), dtype=torch.uint8)
        dist.recv(buf, src, group=group)
        return pickle.loads(buf.numpy().tobytes())
    for m in members:
        if m == rank:
            out.append(obj)
        elif rank < m:
            send(m); out.append(recv(m))
        else:
            got = recv(m); send(m); out.append(got)
    return out


def assemble_results(parts, n_problems):
    """parts: [(problem indices, aln_off, pairs, score, route, num_pw)] covering every problem exactly once -> the batch-order StitchResult"""
    from .capi import StitchResult
    lens = np.zeros(n_problems, dtype=np.int64)
    for pidx, aln_off, _, _, _, _ in parts:
        lens[pidx] = np.diff(np.asarray(aln_off).astype(np.int64))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    pairs = np.zeros((int(off[-1]), 2), dtype=np.uint64)
    score = np.zeros(n_problems, np.int64)
    route = np.zeros(n_problems, np.uint8)
    num_pw = np.zeros(n_problems, np.uint8)
    for pidx, aln_off, p, sc, ro, pw in parts:
        p = np.asarray(p, np.uint64).reshape(-1, 2)
        for j, k in enumerate(pidx):
            pairs[int(off[k]):int(off[k + 1])] = p[int(aln_off[j]):int(aln_off[j + 1])]
        if len(pidx):
            score[pidx], route[pidx], num_pw[pidx] = sc, ro, pw
    return StitchResult(off, pairs, score, route, num_pw)


def gather_results(result, idx, n_problems, dist, rank):
    """rank 0 receives every rank's (problem indices, StitchResult) and reassembles the batch-order result"""
    from .capi import StitchResult
    payload = (idx, result.aln_off, result.pairs, result.score, result.route, result.num_pw)
    if dist is None:
        parts = [payload]
    else:
        parts = [None] * dist.get_world_size() if rank == 0 else None
        dist.gather_object(payload, parts, dst=0)
        if rank != 0:
            return None
    lens = np.zeros(n_problems, dtype=np.int64)
    for pidx, aln_off, _, _, _, _ in parts:
        lens[pidx] = np.diff(aln_off.astype(np.int64))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    pairs = np.zeros((int(off[-1]), 2), dtype=np.uint64)
    score = np.zeros(n_problems, np.int64)
    route = np.zeros(n_problems, np.uint8)
    num_pw = np.zeros(n_problems, np.uint8)
    for pidx, aln_off, p, sc, ro, pw in parts:
        for j, k in enumerate(pidx):
            pairs[int(off[k]):int(off[k + 1])] = p[int(aln_off[j]):int(aln_off[j + 1])]
        score[pidx], route[pidx], num_pw[pidx] = sc, ro, pw
    return StitchResult(off, pairs, score, route, num_pw)


def steal_chunks(batch, chunk_cells=2000000):
    """the subproblems of ONE batch in LPT order (descending DP cells, stable) cut into chunks of about chunk_cells cells: the list every rank of a
    work-stealing pass holds.  A chunk is an index array in ascending problem order; chunk 0 holds the largest subproblems."""
    n1, n2 = batch.sizes()
    cells = (n1 + 1) * (n2 + 1)
    order = np.argsort(-cells, kind="stable")
    chunks, cur, load = [], [], 0
    for k in order:
        cur.append(int(k))
        load += int(cells[k])
        if load >= chunk_cells:
            chunks.append(np.array(sorted(cur), dtype=np.int64))
            cur, load = [], 0
    if cur:
        chunks.append(np.array(sorted(cur), dtype=np.int64))
    return chunks


def stitch_by_stealing(ctx, batch, steal, chunk_cells=2000000, params=None, run=None):
    """One rank's part of a work-stealing stitch pass (north_star: subproblems shard across the GPUs "for work-stealing only"): every rank holds the same
    chunk list (steal_chunks) and pulls chunk numbers from ONE atomic counter until they run out — `steal()` returns the next number: in production
    ctx.peer_steal(job) (a 64-bit word in member 0's exported device memory, system-scope atomics over xGMI: cl_context_peer_steal), in the gloo tests
    the rendezvous store's fetch-add.  Returns (problem indices this rank ran, StitchResult over them in that order, chunk numbers taken); the caller
    gathers with gather_results.  No collective and no rank-to-rank exchange while the pass runs (the reference runs them one after the other:
    stitcher.hpp:157-203)."""
    from .capi import StitchResult
    chunks = steal_chunks(batch, chunk_cells)
    run = run or (lambda sub: ctx.stitch_batch_align(sub, params))
    took, idx_parts, res_parts = [], [], []
    while True:
        c = int(steal())
        if c >= len(chunks):
            break
        took.append(c)
        idx = chunks[c]
        idx_parts.append(idx)
        res_parts.append(run(batch.subset(idx)))
    if not idx_parts:
        return np.zeros(0, np.int64), StitchResult(np.zeros(1, np.uint64), np.zeros((0, 2), np.uint64), np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.uint8)), took
    idx = np.concatenate(idx_parts)
    lens = np.concatenate([np.diff(r.aln_off.astype(np.int64)) for r in res_parts])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    pairs = np.concatenate([np.asarray(r.pairs, np.uint64).reshape(-1, 2) for r in res_parts]) if int(off[-1]) else np.zeros((0, 2), np.uint64)
    cat = lambda name, dt: np.concatenate([np.asarray(getattr(r, name), dt) for r in res_parts])
    return idx, StitchResult(off, pairs, cat("score", np.int64), cat("route", np.uint8), cat("num_pw", np.uint8)), took
