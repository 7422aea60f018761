"""GPU suite: one merge over several contexts (cl_peer_api.cpp).  The members of a merge group all run the same merge and
share the far pass of its affine chaining DP by chain combination, storing what they find into one another's device memory; every member
must return the single-context result.  Here on ONE device: two contexts in one process (two threads), and two processes (ranks of a gloo
group started before any GPU call) running the distributed MSA driver with shared merges — the same code path N GPUs take, with peer
stores that happen to stay on the device."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from centrolign_amd import capi, msa, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def two_plus_two(gpu_ctx, length=100000, seed=11):
    seqs = synth.hor_sequences(seed, length, 4)
    leaves = [capi.leaf_graph(s) for s in seqs]
    scale = sum(gpu_ctx.leaf_intrinsic_scale(g) for g in leaves) / 4
    left = gpu_ctx.merge(leaves[0], leaves[1], score_scale=scale)["fused"]
    right = gpu_ctx.merge(leaves[2], leaves[3], score_scale=scale)["fused"]
    return left, right, scale


def test_two_contexts_share_the_far_pass_of_one_merge(gpu_ctx):
    left, right, scale = two_plus_two(gpu_ctx)
    want = gpu_ctx.merge(left, right, score_scale=scale)
    assert want["align"]["chain_combinations"] == 4
    members = [capi.Context(0), capi.Context(0)]
    try:
        handles = [c.peer_export() for c in members]
        for rounds in range(2):                                   # a second merge of the same group: the epochs go on
            for i, c in enumerate(members):
                c.peer_group(handles, i, 16 * (rounds + 1))
            got, errors = [None, None], []

            def run(i):
                try:
                    got[i] = members[i].merge(left, right, score_scale=scale)
                except Exception as e:   # noqa: BLE001
                    errors.append(e)
            threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
            [t.start() for t in threads]
            [t.join(timeout=300) for t in threads]
            assert not errors, errors
            assert all(not t.is_alive() for t in threads), "a member is still waiting for the other"
            for r in got:
                assert capi.graphs_equal(r["fused"], want["fused"])
                assert np.array_equal(r["alignment"], want["alignment"])
        for c in members:
            st = c.peer_stats()
            assert st["shared_dps"] >= 2 and st["shared_far_launches"] > 100 and st["merged_blocks"] > 100, st
        # the arrival words are never reset: an epoch base below what this context has already used is refused (it would pass every wait at once and
        # fold the previous run's slots in), and so is a self-test token that is not above the last one
        assert members[0].peer_stats()["epoch_mark"] > 32
        with pytest.raises(capi.ClError):
            members[0].peer_group(handles, 0, 16)
        for i, c in enumerate(members):
            c.peer_group(handles, i, 64)
        t = [threading.Thread(target=lambda c=c: c.peer_selftest(5)) for c in members]
        [x.start() for x in t]
        [x.join(timeout=60) for x in t]
        assert members[0].peer_stats()["selftest_mark"] == 5
        assert not members[0].peer_selftest(5)
        # out of the group again: an ordinary merge
        st0 = members[0].peer_stats()
        members[0].peer_group([], 0, 0)
        assert capi.graphs_equal(members[0].merge(left, right, score_scale=scale)["fused"], want["fused"])
        assert members[0].peer_stats() == st0
    finally:
        for c in members:
            c.close()


def test_two_ranks_on_one_device_print_the_reference_gfa(tmp_path):
    """the distributed MSA driver with shared merges, two ranks (child processes under torch.distributed.run, gloo) on the one device: the
    root merge of the ten-sequence golden (5 + 5 paths, 25 chain combinations) is run by both ranks as a merge group — IPC-mapped inboxes, stream memory operations —
    and rank 0 prints the reference's GFA"""
    import hashlib
    import re
    z = np.load(os.path.join(H.GOLDEN, "msa_text_big.npz"))
    want = hashlib.sha256(bytes(z["msa10_30k.gfa"])).hexdigest()
    port = 29500 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "merge_group_child.py"), "30000", "200000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = {int(m.group(1)): m for m in re.finditer(r"RANK (\d) shared_merges=(\d+) shared_dps=(\d+) far_launches=(\d+) merged_blocks=(\d+) sha=(\S+)", r.stdout)}
    assert set(lines) == {0, 1}, r.stdout[-2000:]
    for k in (0, 1):
        assert int(lines[k].group(2)) == 1 and int(lines[k].group(3)) >= 1 and int(lines[k].group(4)) > 16 and int(lines[k].group(5)) > 16, lines[k].group(0)
    assert lines[0].group(6) == want
    # ... and the root merge's stitch subproblems were shared too: both members pulled chunks from the one counter, together every subproblem exactly once
    steal = {int(m.group(1)): [int(x) for x in m.groups()[1:]] for m in re.finditer(r"STEAL (\d) batches=(\d+) chunks=(\d+) problems=(\d+) of=(\d+) steals=(\d+)", r.stdout)}
    assert set(steal) == {0, 1}, r.stdout[-2000:]
    assert steal[0][0] == steal[1][0] >= 1 and steal[0][3] == steal[1][3] and steal[0][2] + steal[1][2] == steal[0][3], steal
    assert steal[0][1] + steal[1][1] >= 2 and steal[0][4] >= steal[0][1], steal


def test_two_contexts_steal_chunks_from_one_counter(gpu_ctx):
    """cl_context_peer_steal inside ONE process: two member contexts (threads) pull the chunks of a stitch batch from the counter word in member 0's memory;
    every chunk is run exactly once and the pieces together are the unsharded result"""
    from centrolign_amd import dist as cd
    batch = synth.random_dag_batch(160, seed=33, max_n=50)
    want = gpu_ctx.stitch_batch_align(batch)
    n_chunks = len(cd.steal_chunks(batch, 5000))
    assert n_chunks > 8
    members = [capi.Context(0), capi.Context(0)]
    try:
        handles = [c.peer_export() for c in members]
        for i, c in enumerate(members):
            c.peer_group(handles, i, 1)
        out, errors = [None, None], []

        def run(i):
            try:
                out[i] = cd.stitch_by_stealing(members[i], batch, lambda: members[i].peer_steal(3), chunk_cells=5000)
            except Exception as e:   # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in threads]
        [t.join(timeout=300) for t in threads]
        assert not errors, errors
        assert sorted(out[0][2] + out[1][2]) == list(range(n_chunks))
        got = {}
        for idx, res, _ in out:
            for j, k in enumerate(idx):
                got[int(k)] = res.alignment(j)
        assert len(got) == batch.n_problems
        for k in range(batch.n_problems):
            assert np.array_equal(got[k], want.alignment(k)), k
        assert members[0].peer_stats()["steals"] + members[1].peer_stats()["steals"] == n_chunks + 2   # (each member's last pull finds the list exhausted)
        # without a group the count is local: one context takes every chunk
        gpu_ctx.peer_group([], 0, 0)
        idx, res, took = cd.stitch_by_stealing(gpu_ctx, batch, lambda: gpu_ctx.peer_steal(1), chunk_cells=5000)
        assert took == list(range(n_chunks)) and sorted(idx.tolist()) == list(range(batch.n_problems))
    finally:
        for c in members:
            c.close()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_device_steal_stitch_chunks(world):
    """north_star's work stealing with its production transport: `world` processes on the one device, one atomic counter in rank 0's hipIpc-exported memory,
    chunks of the LPT-ordered subproblem list; compared with the unsharded pass and the static LPT sharding (tests/steal_child.py)"""
    port = 29500 + ((os.getpid() + 7 * world) % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "steal_child.py"), "240", "5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "STEAL OK world=%d" % world in r.stdout, r.stdout[-2000:]
