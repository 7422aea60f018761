"""Golden for the CLI's -c flow on MANY sequences (a scaled BASELINE configs[4]): 16 HOR arrays of ~12 kbp, six of them with a recent 4-kbp tandem
duplication, balanced guide tree (root merge 8 + 8 paths = 64 chain combinations), -c with min_cyclizing_length 2 500 and 60 000 match
pairs: the GFA the UNMODIFIED compiled reference prints (oracle/_ref/ref_cli), as text size + sha256 + the text itself (gzip).  1.8 minutes.

usage (build container only):  python tests/golden/make_cyclize_wide.py [--from-dir DIR]   # DIR = an earlier run's directory (in.fa, out.gfa)
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

N, SEED, LENGTH, DUP, CARRIERS, HOR_DIV, MIN_LEN, BUDGET = 16, 41, 12000, 4000, [0, 3, 5, 8, 9, 13], 0.08, 2500, 60000


def workload():
    from centrolign_amd import msa, synth
    seqs = synth.tandem_dup_sequences(SEED, LENGTH, N, DUP, carriers=CARRIERS, hor_div=HOR_DIV)
    names = ["c%02d" % i for i in range(N)]
    return names, seqs, msa.newick(msa.balanced_tree(names)) + ";"


def main():
    from centrolign_amd import synth
    names, seqs, newick = workload()
    if "--from-dir" in sys.argv:
        d = sys.argv[sys.argv.index("--from-dir") + 1]
    else:
        d = tempfile.mkdtemp(prefix="cyc_ref_")
        synth.write_fasta(os.path.join(d, "in.fa"), seqs, names)
        open(os.path.join(d, "t.nwk"), "w").write(newick + "\n")
        subprocess.check_call(["bash", "-c", "cd %s && %s in.fa t.nwk - out.gfa 0 2 0 'b:cyclize_tandem_duplications=1;i:min_cyclizing_length=%d;i:max_num_match_pairs=%d' > ref.log 2>&1"
                               % (d, os.path.join(ROOT, "oracle", "_ref", "ref_cli"), MIN_LEN, BUDGET)])
    recs = open(os.path.join(d, "in.fa")).read().split(">")[1:]
    assert [r.split("\n", 1)[0] for r in recs] == names and ["".join(r.split("\n")[1:]) for r in recs] == list(seqs)
    gfa = open(os.path.join(d, "out.gfa"), "rb").read()
    with open(os.path.join(HERE, "cyclize_16x12k.gfa.gz"), "wb") as f:
        f.write(gzip.compress(gfa, 9, mtime=0))
    out = {"workload": "tandem_dup_sequences(seed %d, %d, %d, dup %d, carriers %s, hor_div %g), names c00..c15, balanced tree" % (SEED, LENGTH, N, DUP, CARRIERS, HOR_DIV),
           "newick": newick, "min_cyclizing_length": MIN_LEN, "max_num_match_pairs": BUDGET,
           "input_sha256": hashlib.sha256("".join(seqs).encode()).hexdigest(),
           "reference": "oracle/_ref/ref_cli -c (the unmodified reference), build container, 1 core, 1.8 minutes",
           "gfa": {"sha256": hashlib.sha256(gfa).hexdigest(), "bytes": len(gfa)}}
    with open(os.path.join(HERE, "cyclize_16x12k.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
