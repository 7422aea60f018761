"""CPU suite: the MSA driver over 2 and 3 ranks (gloo): leaf work round-robin, sibling subtrees on different ranks, the fused
graph of the right child sent to the owner of the left one.  A stub stands in for the device context (this test is about the
schedule and the graph transport): its merge is the real host-side cl_fuse along a trivial alignment, so real graphs travel."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from centrolign_amd import capi, msa

class StubCtx:
    """deterministic stand-ins: the scale is a function of the leaf alone; merge = cl_fuse along 'first bases aligned'"""
    def leaf_intrinsic_scale(self, g, **kw):
        return float(int(g.label.astype(np.int64).sum()) %% 9973) / 7.0
    def merge(self, g1, g2, score_scale=1.0, **kw):
        k = min(5, len(g1.label) - 2, len(g2.label) - 2)
        p1 = g1.path_nodes[int(g1.path_off[0]):int(g1.path_off[0]) + k].astype(np.uint64)
        p2 = g2.path_nodes[int(g2.path_off[0]):int(g2.path_off[0]) + k].astype(np.uint64)
        pairs = np.stack([p1, p2], 1)
        return dict(fused=capi.fuse(g1, g2, pairs), alignment=pairs, match_ms=0.0, align_ms=0.0, fuse_ms=0.0, n_match_sets=0)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(4)
names = ["s%%d" %% i for i in range(7)]
seqs = {nm: "".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(20, 60)))) for nm in names}
for tree in (msa.balanced_tree(names), msa.balanced_tree(names[:2]), ((("s0", "s1"), "s2"), ("s3", ("s4", ("s5", "s6"))))):
    got = msa.progressive_msa_distributed(StubCtx(), seqs, tree, dist, rank, world)
    # three worker contexts inside every rank: a rank's independent merges and calibrations side by side, same result
    par = msa.progressive_msa_distributed(StubCtx(), seqs, tree, dist, rank, world, workers=3, make_context=StubCtx)
    if rank == 0:
        assert capi.graphs_equal(got["root"], par["root"]) and got["scales"] == par["scales"] and got["stats"]["merges"] == par["stats"]["merges"]
        want = msa.progressive_msa(StubCtx(), seqs, tree)
        assert capi.graphs_equal(got["root"], want["root"]) and got["paths"] == want["paths"]
        assert got["scale"] == want["scale"] and got["scales"] == want["scales"]
        assert got["stats"]["merges"] <= want["stats"]["merges"] and got["stats"]["graphs_received"] >= 1
        assert len(names) < 7 or got["stats"]["merges"] < want["stats"]["merges"] or tree == msa.balanced_tree(names[:2])
    else:
        assert got is None
dist.barrier()
if rank == 0:
    print("MSA DIST OK")
dist.destroy_process_group()
''' % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_msa_over_ranks_equals_serial(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29640 + world), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "MSA DIST OK" in p.stdout


def test_rank_split_is_proportional():
    from centrolign_amd import msa
    t = msa.balanced_tree(["a", "b", "c", "d", "e", "f", "g", "h"])
    assert msa.split_ranks(t, [0, 1, 2, 3]) == ([0, 1], [2, 3])
    assert msa.split_ranks((("a", "b"), "c"), [0, 1, 2]) == ([0, 1], [2])
    assert msa.split_ranks(("a", ("b", ("c", "d"))), [0, 1]) == ([0], [1])
