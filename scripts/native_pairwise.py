"""BASELINE configs[1] end to end WITHOUT the reference in the loop: two sequences -> leaf graphs -> calibration (both leaves'
intrinsic scales, their mean) -> one merge (match finding, Core::align, fuse) -> explicit CIGAR, all through the C ABI; then, where
oracle/_ref is present, the compiled reference's whole pipeline on the same FASTA (ref_msa_dump = Core::execute + explicit_cigar)
and a byte comparison of the two CIGAR strings."""
import hashlib
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from centrolign_amd import capi, synth  # noqa: E402


def decode(seq):
    return seq if isinstance(seq, str) else "".join("ACGT"[int(b)] for b in seq)


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    seqs = [decode(s) for s in synth.hor_sequences(seed, length, 2)]
    ctx = capi.Context(0)
    ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))   # first-use initialisation outside the timed region
    t0 = time.perf_counter()
    leaves = [capi.leaf_graph(s) for s in seqs]
    t1 = time.perf_counter()
    if "--workers2" in sys.argv:   # the two leaf calibrations side by side, each on its own cl_context
        from concurrent.futures import ThreadPoolExecutor
        ctx2 = capi.Context(0)
        with ThreadPoolExecutor(2) as pool:
            scales = list(pool.map(lambda a: a[0].leaf_intrinsic_scale(a[1]), zip((ctx, ctx2), leaves)))
        ctx2.close()
    else:
        scales = [ctx.leaf_intrinsic_scale(g) for g in leaves]
    scale = sum(scales) / len(scales)                      # src/core.cpp:169-184
    t2 = time.perf_counter()
    r = ctx.merge(leaves[0], leaves[1], score_scale=scale)
    t3 = time.perf_counter()
    cigar = capi.explicit_cigar(leaves[0], leaves[1], r["alignment"])
    t4 = time.perf_counter()
    print("native: leaf graphs %.3f s, calibration %.3f s (scale %.17g), merge %.3f s (matches %.0f ms, align %.0f ms, fuse %.0f ms), "
          "CIGAR %.3f s; total %.3f s; %d match sets, %d aligned pairs, CIGAR sha256 %s" %
          (t1 - t0, t2 - t1, scale, t3 - t2, r["match_ms"], r["align_ms"], r["fuse_ms"], t4 - t3, t4 - t0, r["n_match_sets"],
           len(r["alignment"]), hashlib.sha256(cigar).hexdigest()[:16]), flush=True)
    from oracle import pyoracle as po
    if po.have_ref():
        with tempfile.TemporaryDirectory() as d:
            fa, out = os.path.join(d, "in.fa"), os.path.join(d, "out.txt")
            synth.write_fasta(fa, seqs)
            t = time.perf_counter()
            tm = po.ref_msa_dump(fa, out_path=out)
            wall = time.perf_counter() - t
            want = open(out, "rb").read().rstrip(b"\n")
        print("reference (1 core): %.1f s wall; calibration %.1f, match finding %.1f, chaining %.1f, subalign %.1f s" %
              (wall, tm["calibration"], tm["match_finding"], tm["chaining"], tm["subalign"]))
        print("CIGAR identical to the reference's:", want == cigar, "(%d bytes)" % len(cigar))


if __name__ == "__main__":
    main()
