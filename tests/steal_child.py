"""child of tests/test_gpu_merge_group.py: one rank of a work-stealing stitch pass on ONE device (started under torch.distributed.run, gloo, before any GPU call
of its own): the ranks' contexts form a group (handles exchanged over gloo), every rank pulls chunks of the same LPT-ordered list from the ONE counter word in
member 0's exported device memory (cl_context_peer_steal), rank 0 gathers and compares with the unsharded pass and with the static LPT sharding.
usage: steal_child.py n_problems job"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import capi, dist as cd, synth  # noqa: E402


def main():
    n_problems, job = int(sys.argv[1]), int(sys.argv[2])
    rank, world, dist = cd.init_distributed("gloo")
    batch = synth.random_dag_batch(n_problems, seed=21, max_n=60)
    ctx = capi.Context(0)
    handles = [None] * world
    dist.all_gather_object(handles, ctx.peer_export())
    ctx.peer_group(handles, rank, 1)
    dist.barrier()
    took_all = []
    for round_ in range(2):                     # a second job on the same counter: job numbers grow, nothing is reset
        idx, res, took = cd.stitch_by_stealing(ctx, batch, lambda: ctx.peer_steal(job + round_), chunk_cells=6000)
        full = cd.gather_results(res, idx, batch.n_problems, dist, rank)
        parts = [None] * world
        dist.all_gather_object(parts, took)
        took_all.append(parts)
        if rank == 0:
            want = ctx.stitch_batch_align(batch)
            assert full.same_as(want) is None, full.same_as(want)
            n_chunks = len(cd.steal_chunks(batch, 6000))
            assert sorted(c for t in parts for c in t) == list(range(n_chunks)), parts
    # a stale job number is refused
    stale = False
    try:
        ctx.peer_steal(job - 1)
    except capi.ClError:
        stale = True
    assert stale
    # static LPT sharding of the same batch gives the same gathered result
    shards = cd.shard_problems(batch, world)
    res = ctx.stitch_batch_align(batch.subset(shards[rank]))
    full = cd.gather_results(res, shards[rank], batch.n_problems, dist, rank)
    if rank == 0:
        assert full.same_as(ctx.stitch_batch_align(batch)) is None
        print("STEAL OK world=%d chunks_per_rank=%s steals=%d" % (world, [[len(t) for t in p] for p in took_all], ctx.peer_stats()["steals"]), flush=True)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
