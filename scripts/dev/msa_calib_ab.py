"""10 x 1 Mbp MSA, four merge workers: calibrations on the merge workers' contexts (rounds 2-3) against contexts of their own.  usage: python scripts/dev/msa_calib_ab.py"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centrolign_amd import capi, msa, synth  # noqa: E402

names, seqs, tree = synth.c3_workload(1000000)
ctx = capi.Context(0)
for cc in (0, 8, 0, 8, 10):
    t0 = time.perf_counter()
    r = msa.progressive_msa(ctx, seqs, tree, workers=4, calibration_contexts=cc)
    wall = time.perf_counter() - t0
    tl = dict(r["stats"]["timeline_s"])
    print("calibration contexts %2d: wall %.2f s, calibrations done at %.2f s (leaf graphs at %.2f), gfa %s" %
          (cc, wall, tl.get("calibrations done", 0), tl.get("leaf graphs done", 0), hashlib.sha256(msa.output_text(r)).hexdigest()[:12]), flush=True)
