"""the stitch batches of the nine merges of the 10 x 1 Mbp MSA (BASELINE configs[2]) concatenated into ONE batch, saved as an .npz (the arrays of the two
cl_graph_side + only_deletion_alns + the first problem of every merge) — and the kernel each problem is routed to — so that their structure can be studied, and
new kernels tried against the oracle, without a device.  usage (GPU box): python scripts/dev/dump_c3_batches.py OUT.npz [length]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth  # noqa: E402
from bench import stitch_batches  # noqa: E402

ctx = capi.Context(0)
names, seqs, tree = synth.c3_workload(int(sys.argv[2]) if len(sys.argv) > 2 else 1000000)
r = msa.progressive_msa(ctx, seqs, tree, workers=4, keep_merges=True)
bs = stitch_batches(r["stats"]["kept"])
big = capi.StitchBatch.concat([b for _, b in bs])
out = {}
for si, s in enumerate(big.side):
    for k in capi._SIDE_DTYPES:
        a = getattr(s, k)
        if a is not None and k != "back_translation":
            out["side%d.%s" % (si, k)] = a
out["only_deletion_alns"] = big.only_deletion_alns if big.only_deletion_alns is not None else np.zeros(big.n_problems, np.uint8)
out["merge_first_problem"] = np.cumsum([0] + [b.n_problems for _, b in bs])
out["merge_names"] = np.array([m for m, _ in bs])
np.savez_compressed(sys.argv[1], **out)
print("saved %d problems, %d cells" % (big.n_problems, big.dp_cells()))
