"""The ROOT merge of BASELINE configs[2] (10 x 1 Mbp) at the DEFAULT budget (1.25 M match pairs) from the UNMODIFIED compiled reference — the one
output of the headline configuration the build container cannot produce (the reference's 36 sets of search trees need more memory than it has:
std::bad_alloc at 40 GB, killed at 65 GB).  A GPU box has the host RAM: this script (run THERE, through gpurun, with oracle/_ref travelling with
the repo) writes the eight subproblem files with this implementation (-S; each is pinned byte for byte against the reference's own file by
tests/golden/c3_10x1M_subproblems.json — checked again here), lets oracle/_ref/ref_cli restart from them (-R, src/core.cpp:1071-1081; the
reference recomputes the ten calibrations and the root merge only) and records the sha256 of the GFA it prints, its wall-clock and its peak
resident memory in gpurun_out/c3_root_default.json.  The digest goes into c3_10x1M_subproblems.json ("root_default_budget_reference") by hand
from that file; tests/test_c3_full.py compares the device's root with it.

usage (GPU box): python tests/golden/make_c3_root_default.py [time limit in seconds, default 4500]"""
import hashlib
import json
import os
import resource
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    limit = int(sys.argv[1]) if len(sys.argv) > 1 else 4500
    from centrolign_amd import capi, msa, synth
    gold = json.load(open(os.path.join(HERE, "c3_10x1M_subproblems.json")))
    out_path = os.path.join(ROOT, "gpurun_out", "c3_root_default.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    d = tempfile.mkdtemp(prefix="c3_root_")
    names, seqs, tree = synth.c3_workload()
    synth.write_fasta(os.path.join(d, "in.fa"), [seqs[n] for n in names], names)
    open(os.path.join(d, "t.nwk"), "w").write(synth.C3_NEWICK + "\n")
    ctx = capi.Context(0)
    prefix = os.path.join(d, "sub")
    t0 = time.perf_counter()
    r = msa.progressive_msa(ctx, seqs, tree, workers=4, subproblems_prefix=prefix)
    ours = capi.write_gfa(r["root"], r["paths"])
    ours_sha = hashlib.sha256(ours).hexdigest()
    print("this implementation: %.1f s, root GFA sha256 %s (%d bytes)" % (time.perf_counter() - t0, ours_sha, len(ours)), flush=True)
    ctx.close()
    # the subproblem files: the eight the reference finished must be its own files byte for byte; the root's file and its line go away
    info = prefix + "_info.txt"
    lines = open(info).read().splitlines()
    keep = [lines[0]]
    for ln in lines[1:]:
        fname, leaves = ln.split("\t")
        data = open(fname, "rb").read()
        if leaves in gold["subproblems"]:
            assert hashlib.sha256(data).hexdigest() == gold["subproblems"][leaves]["sha256"], leaves
            keep.append(ln)
        else:
            assert len(leaves.split(",")) == 10, leaves
            os.remove(fname)
    assert len(keep) == 9, keep
    open(info, "w").write("\n".join(keep) + "\n")
    res = {"what": "the ROOT merge of 10 x 1 Mbp at the default budget (1 250 000 match pairs) by the unmodified reference, restarted (-R) from its eight "
                   "subproblem files (reproduced here byte for byte and checked against their pinned digests)",
           "this_implementation_sha256": ours_sha, "this_implementation_bytes": len(ours), "time_limit_s": limit}
    cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_cli"), os.path.join(d, "in.fa"), os.path.join(d, "t.nwk"), prefix, os.path.join(d, "out.gfa"), "0", "4", "1"]
    t0 = time.perf_counter()
    with open(os.path.join(d, "ref.log"), "w") as log:
        p = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT)
        try:
            rc = p.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            rc = "killed after the time limit"
    res["reference_wall_minutes"] = (time.perf_counter() - t0) / 60
    res["reference_exit"] = rc
    res["reference_peak_rss_gb"] = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6
    res["reference_log_tail"] = open(os.path.join(d, "ref.log")).read().strip().splitlines()[-6:]
    try:
        res["host_mem_total_gb"] = int(open("/proc/meminfo").readline().split()[1]) / 1e6
    except Exception:
        pass
    if rc == 0 and os.path.exists(os.path.join(d, "out.gfa")):
        ref = open(os.path.join(d, "out.gfa"), "rb").read()
        res["reference_sha256"] = hashlib.sha256(ref).hexdigest()
        res["reference_bytes"] = len(ref)
        res["identical"] = bool(ref == ours or ref.rstrip(b"\n") == ours.rstrip(b"\n"))
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
