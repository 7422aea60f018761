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
