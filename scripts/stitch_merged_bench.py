"""the stitch work of the 10 x 1 Mbp MSA (all nine merges' subproblems) from resident plans: nine plans side by side, one per merge, as
bench.py times it — against ONE plan over the concatenated batch (subproblems of different merges in the same launches).
usage: python scripts/stitch_merged_bench.py [length]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from bench import stitch_batches  # noqa: E402
from centrolign_amd import capi, msa, synth  # noqa: E402

length = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
names, seqs, tree = synth.c3_workload(length)
ctx = capi.Context(0)
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
res = msa.progressive_msa(ctx, seqs, tree, workers=4, keep_merges=True)
batches = stitch_batches(res["stats"]["kept"])
cells = sum(b.dp_cells() for _, b in batches)


def timed(plans, steps=20, warmup=3):
    def one_pass():
        for p in plans:
            p.execute()
        return max(p.sync() for p in plans)
    for _ in range(warmup):
        one_pass()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    return (time.perf_counter() - t0) / steps * 1e3


ctxs = [capi.Context(0) for _ in batches]
nine = [c.plan(b) for (_, b), c in zip(batches, ctxs)]
ms9 = timed(nine)
print("nine plans side by side: %.2f ms per pass, %.1f G cells/s" % (ms9, cells / ms9 / 1e6), flush=True)
merged = capi.concat_stitch_batches([b for _, b in batches])
one = ctx.plan(merged)
st = one.stats()
ms1 = timed([one])
print("one plan over the concatenated batch (%d problems, %d launches): %.2f ms per pass, %.1f G cells/s" % (merged.n_problems, st["n_launches"], ms1, cells / ms1 / 1e6), flush=True)
# the same alignments?
r9 = [p.collect() for p in nine]
r1 = one.collect()
a = np.concatenate([np.asarray(r.pairs).reshape(-1) for r in r9])
print("identical alignments:", np.array_equal(a, np.asarray(r1.pairs).reshape(-1)), flush=True)
for li in sorted(one.launches() if hasattr(one, "launches") else [], key=lambda e: -e.get("ms", 0))[:6]:
    print("  ", {k: li[k] for k in ("kernel", "n_problems", "dp_cells", "max_sweep") if k in li})
