"""phase times of one leaf merge at BASELINE configs[1] scale (2 x 1 Mbp): CL_CHAIN_TIMING=1 python scripts/dev/leaf_merge_timing.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centrolign_amd import capi, synth  # noqa: E402

seqs = synth.hor_sequences(7, 1000000, 2)
ctx = capi.Context(0)
leaves = [capi.leaf_graph(s) for s in seqs]
scale = sum(ctx.leaf_intrinsic_scale(g) for g in leaves) / 2
ctx.merge(leaves[0], leaves[1], score_scale=scale)   # warm-up (pools, pinned area)
sys.stderr.write("==== timed merge ====\n")
t0 = time.perf_counter()
r = ctx.merge(leaves[0], leaves[1], score_scale=scale)
print("merge %.1f ms: match %.1f align %.1f fuse %.1f" % ((time.perf_counter() - t0) * 1e3, r["match_ms"], r["align_ms"], r["fuse_ms"]))
