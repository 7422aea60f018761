"""leaf merges (2 x 1 Mbp) side by side: N threads in one process, each with its own cl_context — or run several copies of this script at once
(N = 1) for N processes.  Prints every merge's wall-clock and the device time of its chaining DPs.
usage: python scripts/dev/contention.py n_threads [n_merges] [tag]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centrolign_amd import capi, synth  # noqa: E402

n_threads = int(sys.argv[1])
n_merges = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tag = sys.argv[3] if len(sys.argv) > 3 else ""
seqs = synth.hor_sequences(7, 1000000, 2)
leaves = [capi.leaf_graph(s) for s in seqs]
ctxs = [capi.Context(0) for _ in range(n_threads)]
for c in ctxs:
    c.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
barrier = threading.Barrier(n_threads)


def work(i):
    for k in range(n_merges):
        barrier.wait()
        t0 = time.perf_counter()
        r = ctxs[i].merge(leaves[0], leaves[1], score_scale=0.64)
        print("%s thread %d merge %d: %.0f ms wall, align %.0f, chain device %.0f" % (tag, i, k, 1e3 * (time.perf_counter() - t0), r["align_ms"], r["align"]["chain_device_ms"]), flush=True)


ts = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
[t.start() for t in ts]
[t.join() for t in ts]
