"""CPU suite: the N > 1 path with world_size 2 over gloo — LPT sharding of one stitch batch, per-rank work (the
oracle stands in for the device here: this test is about the sharding / gather / max-reduce plumbing), gather on
rank 0, identical to the unsharded result."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from centrolign_amd import synth, dist as cd
from oracle import pyoracle as po
rank, world, dist = cd.init_distributed("gloo")
assert world == 2
batch = synth.random_dag_batch(120, seed=5, max_n=30)
shards = cd.shard_problems(batch, world)
assert sorted(np.concatenate(shards).tolist()) == list(range(batch.n_problems))
mine = shards[rank]
t0 = time.perf_counter()
res = po.oracle_stitch_batch(batch.subset(mine))
dist.barrier()
elapsed = cd.max_over_ranks(time.perf_counter() - t0 + rank, dist)   # rank 1 reports >= 1 s more
assert elapsed >= 1.0
full = cd.gather_results(res, mine, batch.n_problems, dist, rank)
if rank == 0:
    want = po.oracle_stitch_batch(batch)
    assert full.same_as(want) is None, full.same_as(want)
    n1, n2 = batch.sizes()
    cells = (n1 + 1) * (n2 + 1)
    loads = [int(cells[s].sum()) for s in shards]
    assert max(loads) <= 1.2 * (sum(loads) / 2) + int(cells.max())
    print("DIST OK", loads)
dist.destroy_process_group()
''' % ROOT


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "DIST OK" in p.stdout


STEAL_WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as tdist
from centrolign_amd import synth, dist as cd
from oracle import pyoracle as po
rank, world, dist = cd.init_distributed("gloo")
assert world == 3
batch = synth.random_dag_batch(150, seed=9, max_n=36)
chunks = cd.steal_chunks(batch, chunk_cells=4000)
assert sorted(np.concatenate(chunks).tolist()) == list(range(batch.n_problems)) and len(chunks) > 6
n1, n2 = batch.sizes()
cells = (n1 + 1) * (n2 + 1)
assert all(cells[chunks[i]].max() >= cells[chunks[i + 1]].max() for i in range(len(chunks) - 1))   # LPT order: largest first
# ONE atomic counter, every rank pulls from it: here the rendezvous store's fetch-add (the device word of cl_context_peer_steal needs a GPU)
store = tdist.distributed_c10d._get_default_store()
steal = lambda: store.add("steal/job7", 1) - 1
if rank == 2:
    time.sleep(0.5)   # a rank that arrives late finds fewer chunks left: nobody waits for it
idx, res, took = cd.stitch_by_stealing(None, batch, steal, chunk_cells=4000, run=lambda sub: po.oracle_stitch_batch(sub))
full = cd.gather_results(res, idx, batch.n_problems, dist, rank)
all_took = [None] * world
dist.all_gather_object(all_took, took)
if rank == 0:
    flat = sorted(c for t in all_took for c in t)
    assert flat == list(range(len(chunks))), (flat, len(chunks))          # every chunk exactly once
    want = po.oracle_stitch_batch(batch)
    assert full.same_as(want) is None, full.same_as(want)
    print("STEAL OK", [len(t) for t in all_took])
dist.destroy_process_group()
''' % ROOT


def test_three_rank_gloo_work_stealing(tmp_path):
    """north_star's work stealing over the stitch subproblems: one counter, chunks of the LPT-ordered list, every chunk run exactly once, the gathered result is
    the unsharded one (the counter's production transport — a device word reached through hipIpc — is covered by tests/test_gpu_merge_group.py)"""
    script = tmp_path / "steal_worker.py"
    script.write_text(STEAL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "STEAL OK" in p.stdout
