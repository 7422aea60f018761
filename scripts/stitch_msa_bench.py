"""per-launch times of the stitch pass on REAL graph x graph batches: the merges of a 10-sequence MSA (200 kbp per sequence by default),
replayed from resident plans; a second argument "fill" times the fill alone, CL_NO_SYS=1 the older kernels"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import stitch_batches  # noqa: E402
from centrolign_amd import capi, msa, synth  # noqa: E402


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    names, seqs, tree = synth.c3_workload(length)
    ctx = capi.Context(0)
    r = msa.progressive_msa(ctx, seqs, tree, max_num_match_pairs=400000, workers=4, keep_merges=True)
    if len(sys.argv) > 2 and sys.argv[2] == "fill":   # read when a plan is made: the MSA above needed its tracebacks
        os.environ["CL_DEBUG_SKIP_TRACEBACK"] = "1"
    for m, b in stitch_batches(r["stats"]["kept"])[4:]:
        plan = ctx.plan(b)
        for _ in range(2):
            plan.execute(); plan.sync()
        best = {}
        for _ in range(4):
            plan.execute_profiled(); plan.sync()
            for li in plan.launches():
                k = (li["kernel"], li["n_problems"], li["dp_cells"], li["lds_bytes"], li["longest"])
                best[k] = min(best.get(k, 1e9), li["ms"])
        n1, n2 = b.sizes()
        big = sorted(zip((n1 + 1) * (n2 + 1), n1, n2), reverse=True)[:3]
        print(m[:40], "problems", b.n_problems, "largest", [(int(x[1]), int(x[2])) for x in big])
        for (kern, npb, cells, lds, longest), ms in sorted(best.items(), key=lambda kv: -kv[1])[:8]:
            print("    %-30s %5d problems %10d cells %8.3f ms   LDS %6d B, longest sweep %d x %d = %.2f us per step" %
                  (kern, npb, cells, ms, lds, longest[0], longest[1], 1e3 * ms / max(1, sum(longest))))
        plan.destroy()


if __name__ == "__main__":
    main()
