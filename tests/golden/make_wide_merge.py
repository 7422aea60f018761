"""Goldens for merges with many chain combinations (CASES below): sequences over a balanced guide tree whose root merge pairs 12 + 12 paths (144
combinations of path-merge chains: the walk kernel with its reduction exchange) and 25 + 25 paths (625: beyond the walk kernel, on the
per-block kernels) — the GFA the UNMODIFIED compiled reference prints (oracle/_ref/ref_cli, default parameters), as text size + sha256 + the
text itself (gzip).  One core, 26 and 34 minutes.

usage (build container only):  python tests/golden/make_wide_merge.py CASE [--from-dir DIR]   # DIR = an earlier run's directory (in.fa, out.gfa)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {   # name: sequences, seed, length, name prefix, reference minutes
    # root merge 12 + 12 paths = 144 chain combinations: the walk kernel with its reduction exchange
    "wide_merge_24x7k": (24, 91, 7000, "q", 26),
    # root merge 25 + 25 paths = 625 combinations: beyond the walk kernel's 256, on the per-block kernels (BASELINE configs[4]'s shape without -c)
    "wide_merge_50x5k": (50, 92, 5000, "r", 34),
}


def workload(case):
    from centrolign_amd import msa, synth
    n, seed, length, prefix, _ = CASES[case]
    seqs = synth.hor_sequences(seed, length, n, indel_hor=1)
    names = ["%s%02d" % (prefix, i) for i in range(n)]
    return names, seqs, msa.newick(msa.balanced_tree(names)) + ";"


def main():
    from centrolign_amd import synth
    case = sys.argv[1]
    n, seed, length, prefix, minutes = CASES[case]
    names, seqs, newick = workload(case)
    if "--from-dir" in sys.argv:
        d = sys.argv[sys.argv.index("--from-dir") + 1]
    else:
        d = tempfile.mkdtemp(prefix="wide_ref_")
        synth.write_fasta(os.path.join(d, "in.fa"), seqs, names)
        open(os.path.join(d, "t.nwk"), "w").write(newick + "\n")
        subprocess.check_call(["bash", "-c", "cd %s && %s in.fa t.nwk - out.gfa 0 2 > ref.log 2>&1" % (d, os.path.join(ROOT, "oracle", "_ref", "ref_cli"))])
    # the run's input is the workload of this script
    recs = open(os.path.join(d, "in.fa")).read().split(">")[1:]
    assert [r.split("\n", 1)[0] for r in recs] == names and ["".join(r.split("\n")[1:]) for r in recs] == list(seqs)
    assert open(os.path.join(d, "t.nwk")).read().strip() == newick
    gfa = open(os.path.join(d, "out.gfa"), "rb").read()
    import gzip
    with open(os.path.join(HERE, case + ".gfa.gz"), "wb") as f:
        f.write(gzip.compress(gfa, 9, mtime=0))
    out = {"workload": "hor_sequences(seed %d, %d, %d, indel_hor=1), names %s00.., balanced tree" % (seed, length, n, prefix), "newick": newick,
           "input_sha256": hashlib.sha256("".join(seqs).encode()).hexdigest(),
           "reference": "oracle/_ref/ref_cli (the unmodified reference, default parameters), build container, 1 core, %d minutes" % minutes,
           "gfa": {"sha256": hashlib.sha256(gfa).hexdigest(), "bytes": len(gfa)}}
    with open(os.path.join(HERE, case + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
