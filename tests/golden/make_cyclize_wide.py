"""Goldens for the CLI's -c flow on MANY sequences (BASELINE configs[4] scaled down): HOR arrays some of which carry a recent tandem duplication,
balanced guide tree, -c with a small min_cyclizing_length and match-pair budget (CASES below): the GFA the UNMODIFIED compiled reference
prints (oracle/_ref/ref_cli), as text size + sha256 + the text itself (gzip).

usage (build container only):  python tests/golden/make_cyclize_wide.py CASE [--from-dir DIR]   # DIR = an earlier run's directory (in.fa, out.gfa)
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {   # name: sequences, seed, length, duplicated bases, carriers, hor_div, min_cyclizing_length, max_num_match_pairs, name prefix, reference minutes
    "cyclize_16x12k": (16, 41, 12000, 4000, [0, 3, 5, 8, 9, 13], 0.08, 2500, 60000, "c", 1.8),
    # BASELINE configs[4]'s shape at small length: 50 sequences, root merge 25 + 25 paths = 625 chain combinations (beyond the walk kernel's 256: the
    # per-block kernels), 133 bonds, 550 polished regions
    "cyclize_50x8k": (50, 43, 8000, 3000, [1, 4, 7, 12, 18, 23, 29, 31, 36, 40, 44, 48], 0.08, 2000, 40000, "d", 5.1),
    # (round 6 tried the same width at five times the length — 50 x 40 kbp, 10-kbp duplications in six sequences, budget 250 000: the unmodified reference was stopped
    # after 3.4 hours on one core when its resident set passed 46 GB of this container's 62; no golden of that size can be made here)
}


def workload(case):
    from centrolign_amd import msa, synth
    n, seed, length, dup, carriers, hor_div = CASES[case][:6]
    seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=hor_div)
    names = ["%s%02d" % (CASES[case][8], i) for i in range(n)]
    return names, seqs, msa.newick(msa.balanced_tree(names)) + ";"


def main():
    from centrolign_amd import synth
    case = sys.argv[1]
    n, seed, length, dup, carriers, hor_div, min_len, budget, prefix, minutes = CASES[case]
    names, seqs, newick = workload(case)
    if "--from-dir" in sys.argv:
        d = sys.argv[sys.argv.index("--from-dir") + 1]
    else:
        d = tempfile.mkdtemp(prefix="cyc_ref_")
        synth.write_fasta(os.path.join(d, "in.fa"), seqs, names)
        open(os.path.join(d, "t.nwk"), "w").write(newick + "\n")
        subprocess.check_call(["bash", "-c", "cd %s && %s in.fa t.nwk - out.gfa 0 2 0 'b:cyclize_tandem_duplications=1;i:min_cyclizing_length=%d;i:max_num_match_pairs=%d' > ref.log 2>&1"
                               % (d, os.path.join(ROOT, "oracle", "_ref", "ref_cli"), min_len, budget)])
    recs = open(os.path.join(d, "in.fa")).read().split(">")[1:]
    assert [r.split("\n", 1)[0] for r in recs] == names and ["".join(r.split("\n")[1:]) for r in recs] == list(seqs)
    gfa = open(os.path.join(d, "out.gfa"), "rb").read()
    with open(os.path.join(HERE, case + ".gfa.gz"), "wb") as f:
        f.write(gzip.compress(gfa, 9, mtime=0))
    out = {"workload": "tandem_dup_sequences(seed %d, %d, %d, dup %d, carriers %s, hor_div %g), names %s00.., balanced tree" % (seed, length, n, dup, carriers, hor_div, prefix),
           "newick": newick, "min_cyclizing_length": min_len, "max_num_match_pairs": budget,
           "input_sha256": hashlib.sha256("".join(seqs).encode()).hexdigest(),
           "reference": "oracle/_ref/ref_cli -c (the unmodified reference), build container, 1 core, %.1f minutes" % minutes,
           "gfa": {"sha256": hashlib.sha256(gfa).hexdigest(), "bytes": len(gfa)}}
    with open(os.path.join(HERE, case + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
