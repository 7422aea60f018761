"""Golden for a merge with more chain combinations than one walk launch holds: 24 sequences of 7 kbp over a balanced guide tree, whose root
merge pairs 12 + 12 paths (144 combinations of path-merge chains; the chaining DP's walk kernel takes 96, cl_chain_api.cpp) — the GFA the
UNMODIFIED compiled reference prints (oracle/_ref/ref_cli, default parameters), as text size + sha256.  One core, 26 minutes.

usage (build container only):  python tests/golden/make_wide_merge.py [--from-dir DIR]   # DIR = an earlier run's directory (in.fa, out.gfa)
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

NEWICK = "((((q00,(q01,q02)),(q03,(q04,q05))),((q06,(q07,q08)),(q09,(q10,q11)))),(((q12,(q13,q14)),(q15,(q16,q17))),((q18,(q19,q20)),(q21,(q22,q23)))));"
SEED, LENGTH, N = 91, 7000, 24


def workload():
    from centrolign_amd import synth
    seqs = synth.hor_sequences(SEED, LENGTH, N, indel_hor=1)
    return ["q%02d" % i for i in range(N)], seqs


def main():
    from centrolign_amd import synth
    names, seqs = workload()
    if "--from-dir" in sys.argv:
        d = sys.argv[sys.argv.index("--from-dir") + 1]
    else:
        d = tempfile.mkdtemp(prefix="wide_ref_")
        synth.write_fasta(os.path.join(d, "in.fa"), seqs, names)
        open(os.path.join(d, "t.nwk"), "w").write(NEWICK + "\n")
        subprocess.check_call(["bash", "-c", "cd %s && %s in.fa t.nwk - out.gfa 0 2 > ref.log 2>&1" % (d, os.path.join(ROOT, "oracle", "_ref", "ref_cli"))])
    # the run's input is the workload of this script
    recs = open(os.path.join(d, "in.fa")).read().split(">")[1:]
    assert [r.split("\n", 1)[0] for r in recs] == names and ["".join(r.split("\n")[1:]) for r in recs] == list(seqs)
    gfa = open(os.path.join(d, "out.gfa"), "rb").read()
    out = {"workload": "hor_sequences(seed %d, %d, %d, indel_hor=1), names q00..q23" % (SEED, LENGTH, N), "newick": NEWICK,
           "input_sha256": hashlib.sha256("".join(seqs).encode()).hexdigest(),
           "reference": "oracle/_ref/ref_cli (the unmodified reference, default parameters), build container, 1 core, 26 minutes",
           "gfa": {"sha256": hashlib.sha256(gfa).hexdigest(), "bytes": len(gfa)}}
    with open(os.path.join(HERE, "wide_merge_24x7k.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
