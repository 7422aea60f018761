"""A progressive MSA WITHOUT the reference in the loop (BASELINE configs 3-4 shape): n sequences + a binary guide tree -> leaf graphs
-> calibration (every leaf's intrinsic scale, their mean) -> one cl_merge per internal tree node (match finding, Core::align, fuse)
-> GFA of the root graph, all through the C ABI; then, where oracle/_ref is present, the compiled reference's whole pipeline on
the same FASTA + Newick (ref_msa_dump = Core::execute + write_gfa) and a byte comparison of the two GFA texts.
The guide tree is the caller's (a balanced binary tree over the sequences in order); Tree / Execution are not rebuilt: node ids follow
the Newick text (src/tree.cpp:75-132), so leaves are calibrated in order of appearance and the first child is graph 1."""
import hashlib
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, synth  # noqa: E402


from centrolign_amd import msa  # noqa: E402


def main():
    n, length, budget, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    names = ["seq%d" % i for i in range(n)]
    tree = msa.balanced_tree(names)
    ctx = capi.Context(0)
    ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))   # first-use initialisation outside the timed region
    t0 = time.perf_counter()
    workers = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5].isdigit() else 1
    r = msa.progressive_msa(ctx, dict(zip(names, seqs)), tree, max_num_match_pairs=budget, workers=workers)
    root, paths, scale, stats = r["root"], r["paths"], r["scale"], r["stats"]
    t1 = t2 = time.perf_counter()
    gfa = capi.write_gfa(root, paths)
    t3 = time.perf_counter()
    print("native (%d worker context%s): leaf graphs + calibration + %d merges %.2f s (scale %.17g; in the merges: matches %.2f, align %.2f, fuse %.2f s), GFA %.2f s; "
          "total %.2f s; root graph %d nodes, GFA %d bytes sha256 %s" % (workers, "" if workers == 1 else "s", stats["merges"], t1 - t0, scale, stats["match_ms"] / 1e3, stats["align_ms"] / 1e3,
                                                                         stats["fuse_ms"] / 1e3, t3 - t2, t3 - t0, len(root.label), len(gfa),
                                                                         hashlib.sha256(gfa).hexdigest()[:16]), flush=True)
    from oracle import pyoracle as po
    if po.have_ref() and "--no-ref" not in sys.argv:
        with tempfile.TemporaryDirectory() as d:
            fa, nwk, out = os.path.join(d, "in.fa"), os.path.join(d, "t.nwk"), os.path.join(d, "out.gfa")
            synth.write_fasta(fa, seqs, names)
            open(nwk, "w").write(msa.newick(tree) + ";")
            t = time.perf_counter()
            tm = po.ref_msa_dump(fa, newick_path=nwk, out_path=out, max_num_match_pairs=budget)
            wall = time.perf_counter() - t
            want = open(out, "rb").read()
        print("reference (1 core): %.1f s wall; calibration %.1f, match finding %.1f, chaining %.1f, subalign %.1f, fuse %.1f s" %
              (wall, tm["calibration"], tm["match_finding"], tm["chaining"], tm["subalign"], tm["fuse"]))
        print("GFA identical to the reference's:", want == gfa, "(%d bytes)" % len(gfa))


if __name__ == "__main__":
    main()
