"""the ROOT merge of the 10 x 1 Mbp MSA (5 + 5 paths, 25 chain combinations) alone on the device: the MSA once to get its two inputs, then the merge K times
(with CL_CHAIN_TIMING=1: the phases on stderr).  usage: python scripts/dev/root_merge.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centrolign_amd import capi, msa, synth  # noqa: E402

names, seqs, tree = synth.c3_workload(1000000)
ctx = capi.Context(0)
r = msa.progressive_msa(ctx, seqs, tree, workers=4)
g1, g2 = r["root_inputs"]
print("MARK root merges begin", file=sys.stderr, flush=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    m = ctx.merge(g1, g2, score_scale=r["scale"])
    al = m.get("align") or {}
    print("root merge: wall %.0f ms; match %.0f align %.0f fuse %.0f; chain %.0f (device %.0f) partition %.0f stitch %.0f" % (
        (time.perf_counter() - t0) * 1e3, m["match_ms"], m["align_ms"], m["fuse_ms"], al.get("chain_ms", 0), al.get("chain_device_ms", 0), al.get("partition_ms", 0), al.get("stitch_ms", 0)), flush=True)
