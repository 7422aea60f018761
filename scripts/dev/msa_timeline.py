"""timeline of the 10 x 1 Mbp MSA by worker count: when the calibrations and each wave of merges are done.  usage: python scripts/dev/msa_timeline.py [workers ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centrolign_amd import capi, msa, synth  # noqa: E402

names, seqs, tree = synth.c3_workload(1000000)
ctx = capi.Context(0)
msa.progressive_msa(ctx, {k: seqs[k] for k in names[:2]}, (names[0], names[1]), workers=1)   # warm-up
for w in [int(x) for x in sys.argv[1:]] or [4, 4, 2, 8]:
    t0 = time.perf_counter()
    r = msa.progressive_msa(ctx, seqs, tree, workers=w)
    wall = time.perf_counter() - t0
    tl = r["stats"]["timeline_s"]
    print("workers %d: wall %.2f s; " % (w, wall) + "; ".join("%s %.2f" % (a, b) for a, b in tl), flush=True)
    pm = r["stats"]["per_merge"]
    print("   align ms per merge: %s" % [round(m["align_ms"]) for m in pm], flush=True)
