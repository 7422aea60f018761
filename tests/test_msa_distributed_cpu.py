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


GROUP_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from centrolign_amd import capi, msa

class GroupStub:
    """the stub context of the test above + the merge-group calls: records every group it was put into"""
    def __init__(self):
        self.groups, self.in_group, self.merges_in_group = [], None, 0
    def leaf_intrinsic_scale(self, g, **kw):
        return float(int(g.label.astype(np.int64).sum()) %% 9973) / 7.0
    def merge(self, g1, g2, score_scale=1.0, **kw):
        self.merges_in_group += self.in_group is not None
        k = min(5, len(g1.label) - 2, len(g2.label) - 2)
        p1 = g1.path_nodes[int(g1.path_off[0]):int(g1.path_off[0]) + k].astype(np.uint64)
        p2 = g2.path_nodes[int(g2.path_off[0]):int(g2.path_off[0]) + k].astype(np.uint64)
        pairs = np.stack([p1, p2], 1)
        return dict(fused=capi.fuse(g1, g2, pairs), alignment=pairs, match_ms=0.0, align_ms=0.0, fuse_ms=0.0, n_match_sets=0)
    def peer_export(self):
        return b"rank%%03d" %% dist.get_rank() + bytes(121)
    def peer_group(self, members, my_index, epoch_base):
        self.in_group = None if not members else ([int(m[4:7]) for m in members], my_index, epoch_base)
        if members:
            self.groups.append(self.in_group)
    def peer_selftest(self, token, timeout_ms=5000):
        return os.environ.get("FAIL_SELFTEST_ON") != str(dist.get_rank())

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(4)
names = ["s%%d" %% i for i in range(8)]
seqs = {nm: "".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(20, 60)))) for nm in names}
tree = msa.balanced_tree(names)
ctx = GroupStub()
got = msa.progressive_msa_distributed(ctx, seqs, tree, dist, rank, world, all_ranks=True, share_merges=3, min_shared_combos=5, make_context=GroupStub)
want = msa.progressive_msa(GroupStub(), seqs, tree)
mine = [g for g in ctx.groups if g[2] != 0]               # (epoch base 0 = the self-test round over all ranks)
every = [None] * world
dist.all_gather_object(every, (mine, ctx.merges_in_group, got["stats"].get("merge_groups"), got["stats"].get("shared_merges", 0)))
if rank == 0:
    assert capi.graphs_equal(got["root"], want["root"]) and got["paths"] == want["paths"] and got["scale"] == want["scale"]
    if os.environ.get("FAIL_SELFTEST_ON"):
        assert all(not e[0] and e[1] == 0 and "OFF" in e[2] for e in every), every
    else:
        # 8 leaves over `world` ranks: the root (4 x 4 = 16 combinations) is shared by min(3, world) ranks, the 2 x 2 merges are not
        # (below min_shared_combos); every member saw the same member list and epoch, its own index, and ran one merge inside the group
        members = every[0][0][-1][0]
        assert members[0] == 0 and len(members) == min(3, world) and len(set(members)) == len(members)
        for r, (groups, n_in, note, n_shared) in enumerate(every):
            if r in members:
                assert len(groups) == 1 and groups[0][0] == members and groups[0][1] == members.index(r) and groups[0][2] == every[0][0][-1][2] and n_in == 1 and n_shared == 1
            else:
                assert not groups and n_in == 0
dist.barrier()
if rank == 0:
    print("MSA GROUPS OK")
dist.destroy_process_group()
''' % ROOT


@pytest.mark.parametrize("world,fail_on", [(2, None), (4, None), (3, "1")])
def test_merge_groups_are_formed_consistently(tmp_path, world, fail_on):
    """the driver's side of one-merge-over-several-GPUs (share_merges) with stub contexts: who is in a group, what every member is told, that
    both children reach every member and the result is the serial one; and that a failed self-test on any rank switches every rank back"""
    script = tmp_path / "worker.py"
    script.write_text(GROUP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if fail_on is not None:
        env["FAIL_SELFTEST_ON"] = fail_on
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29660 + world), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "MSA GROUPS OK" in p.stdout
