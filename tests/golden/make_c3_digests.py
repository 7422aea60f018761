"""Golden digests for BASELINE configs[2] (10 x 1 Mbp, seed 7, guide tree of SURVEY.md §8(d)) from the UNMODIFIED compiled reference.

The reference cannot finish this configuration in the build container: its root merge (5 + 5 paths) runs out of memory
(std::bad_alloc under the 40 GB address-space limit used here; killed by the kernel at 65 GB without a limit).  With -S it writes
every subproblem it does finish as a GFA file (Core::emit_subproblem, /root/reference/src/core.cpp:392-422): this script runs
oracle/_ref/ref_cli (oracle/ref_cli.cpp, the reference's own CLI flow) that way and records the sha256 of each file, keyed by the
subproblem's sorted leaf names, plus the reference's wall-clock at every fuse.  tests/test_c3_full.py (-m gpu) reproduces the digests
natively.

usage (build container only):  python tests/golden/make_c3_digests.py [--from-dir DIR]   # DIR = an earlier run's output directory
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    from centrolign_amd import synth
    if "--from-dir" in sys.argv:
        d = sys.argv[sys.argv.index("--from-dir") + 1]
    else:
        d = tempfile.mkdtemp(prefix="c3_ref_")
        names, seqs, _ = synth.c3_workload()
        synth.write_fasta(os.path.join(d, "in.fa"), [seqs[n] for n in names], names)
        open(os.path.join(d, "t.nwk"), "w").write(synth.C3_NEWICK + "\n")
        cmd = "ulimit -v 40000000; cd %s && %s in.fa t.nwk %s/sub %s/out.gfa 0 4 > ref.log 2>&1" % (d, os.path.join(ROOT, "oracle", "_ref", "ref_cli"), d, d)
        rc = subprocess.call(["bash", "-c", cmd])
        print("ref_cli exit code", rc, "(3 = the expected std::bad_alloc at the root merge)")
    subs = {}
    with open(os.path.join(d, "sub_info.txt")) as f:
        next(f)
        for ln in f:
            path, leaves = ln.rstrip("\n").split("\t")
            data = open(os.path.join(d, os.path.basename(path)), "rb").read()
            subs[leaves] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data)}
    fuse_min = [float(m.group(1)) for m in re.finditer(r"elapsed: ([0-9.]+) m wall / [0-9.]+ m cpu\] Fusing MSAs", open(os.path.join(d, "ref.log")).read())]
    log_tail = open(os.path.join(d, "ref.log")).read().strip().splitlines()[-2:]
    names, seqs, _ = synth.c3_workload()
    out = {"workload": "hor_sequences(seed 7, 1 000 000, 10), names s0..s9, " + synth.C3_NEWICK,
           "input_sha256": hashlib.sha256("".join(seqs[n] for n in names).encode()).hexdigest(),
           "reference": "oracle/_ref/ref_cli (the unmodified reference, -S), build container, 1 core, address space limited to 40 GB",
           "subproblems": subs,
           "reference_wall_minutes_at_each_fuse": fuse_min,
           "reference_end": log_tail,
           "unfinished": "the root merge (s0..s9): " + log_tail[-1]}
    with open(os.path.join(HERE, "c3_10x1M_subproblems.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
