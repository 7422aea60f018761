"""The exact `bench.py --gpus 2` command line on ONE device (round-5 verdict, item 8): bench.py starts its two ranks itself (torch.distributed.run as a child), the ranks find
fewer devices than ranks and take the gloo "share" path (bench.py: `share`), split the MSA's subtrees, shard the stitch batches by LPT, run one stealing pass, and rank 0 prints
ONE JSON line.  A dry-run length keeps it inside the suite's time; what is checked is the plumbing of the multi-rank path — the line's shape, the sharding records, and that a
wrong GFA of a merge-group run is a hard failure (exit code 3, no line) instead of the silent second run of round 5."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--length", "30000", "--no-cpu-baseline", "--no-extras"] + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_bench_with_two_ranks_on_one_device():
    r = run_bench([])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"] == "DP cells/s" and out["value"] > 0
    assert out["config"]["workload"].startswith("DRY RUN at 30000 bp")            # (not the headline: says so)
    sh = out["config"]["stitch_sharding"]
    assert sh and sh["batches"] == 9 and sh["dp_cells_all_ranks"] == out["config"]["dp_cells"]
    assert "work_stealing_pass" in sh and "error" not in sh["work_stealing_pass"], sh
    assert len(out["timed_blocks"]["ms_per_step"]) == 5
    assert out["roofline"]["frac"] > 0 and out["roofline"]["frac_kernel_clock"] is not None and "HIP events" in out["roofline"]["durations_from"]


@pytest.mark.gpu
def test_bench_refuses_to_time_a_wrong_merge_group_result():
    # merge groups of two ranks, the first GFA declared wrong by the test hook: every rank leaves with exit code 3 and nothing is printed.  (The hook is read at the headline
    # length only — a dry-run length has no reference digest —, so this case runs the 10 x 1 Mbp MSA once; ~40 s on two ranks that share a device.)
    env = {"CL_BENCH_FORCE_RETRY": "1"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-merges", "2", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert r.returncode != 0, r.stdout[-500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "wrong multi-GPU result" in r.stderr
