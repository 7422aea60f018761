"""SURVEY.md §8(d)'s kernel-only stress set: batches of synthetic linear and bubble graph pairs at 32^2, 128^2, 512^2, 2 048^2 and 6 300^2 (≈ 40 M cells, the
reference's ceiling for one pair, src/parameters.cpp:79), each batch ≈ 40 M cells: DP cells/s, the kernels that ran, algorithmic bytes (sizeof(cell_t<NumPW>) per cell)
against the 8 TB/s of HBM.  usage: python scripts/stress_set.py [--json OUT]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import capi, synth  # noqa: E402


def main():
    ctx = capi.Context(0)
    rows = []
    for n, count in ((32, 36000), (128, 2400), (512, 150), (2048, 10), (6300, 1)):
        for kind, kw in (("linear", dict(extra_edge_p=0.0, n_alt=0)), ("bubbles", dict(extra_edge_p=0.02, skip_max=2))):
            b = synth.sized_dag_batch([(n, n)] * count, seed=9, **kw)
            plan = ctx.plan(b)
            for _ in range(2):
                plan.execute(); plan.sync()
            best = 1e30
            for _ in range(5):
                t0 = time.perf_counter()
                plan.execute(); plan.sync()
                best = min(best, time.perf_counter() - t0)
            st = plan.stats()
            kernels = sorted({li["kernel"] for li in plan.launches()})
            plan.destroy()
            row = dict(size=n, graphs=kind, problems=count, dp_cells=st["dp_cells"], dp_bytes=st["dp_bytes"], ms=best * 1e3, g_cells_per_s=st["dp_cells"] / best / 1e9,
                       algorithmic_GB_per_s=st["dp_bytes"] / best / 1e9, frac_of_8TBps=st["dp_bytes"] / best / 8e12, kernels=kernels)
            rows.append(row)
            print("%5d^2 x %5d %-8s %8.2f ms  %7.2f G cells/s  %7.1f GB/s algorithmic (%.3f of HBM)  %s" %
                  (n, count, kind, row["ms"], row["g_cells_per_s"], row["algorithmic_GB_per_s"], row["frac_of_8TBps"], ", ".join(kernels)), flush=True)
    if "--json" in sys.argv:
        json.dump(dict(rows=rows, note="wall clock of execute + sync of a resident plan (inputs in HBM), best of five; NumPW chosen by the library per size as Stitcher::subalign does"),
                  open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
