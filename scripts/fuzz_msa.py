"""time-bounded randomised END-TO-END parity campaign on the GPU box: the library's cl_msa (FASTA + Newick -> GFA / CIGAR) against the compiled, unmodified reference
(oracle/_ref/ref_cli — test infrastructure; it travels to the box as a prebuilt binary) run LIVE on the same random input: 2-7 HOR-array sequences of 4.5-14 kbp, random
binary guide trees, random divergence, match-pair budgets that bite or do not, with and without tandem duplications under -c (bond search, apply_bonds, polishing), one
or three worker contexts.  Byte-for-byte comparison of the output text.  A -c case that differs while the library's counter bond_trims_past_the_end is non-zero is run
through the unmodified reference again with MALLOC_PERTURB_=1 / 85 / 170 / 255 (glibc's fill byte for freed and fresh heap memory) and once more as it is: if the reference's own text
changes (or, failing that, if the recorded flow of oracle/ref_driver.cpp — another program round the same unmodified objects, another heap — prints another text), its result on that input is undefined (Bonder::trim_partition_ends reads one element past a vector, src/bonder.cpp:753-757) and the case is listed apart.
usage: python3 scripts/fuzz_msa.py [--seconds T] [--seed S] [--json OUT]; exit code 1 on any difference (each printed with its parameters)"""
import argparse
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centrolign_amd import capi, msa, synth  # noqa: E402

REF_CLI = os.path.join(ROOT, "oracle", "_ref", "ref_cli")


def random_tree(rng, names):
    nodes = list(names)
    while len(nodes) > 1:
        i = rng.randrange(len(nodes) - 1)
        nodes[i:i + 2] = [(nodes[i], nodes[i + 1])]
    return nodes[0]


def one_case(rng, more=False, big=False):
    n = rng.choice([2, 3, 3, 4, 5, 6, 7])
    length = rng.choice([4500, 6000, 9000, 14000])
    if big:   # (--big: the regime of the far pass, several macro-blocks per DP, helpers of the walk, larger stitch problems; the reference needs minutes per case)
        n = rng.choice([2, 3, 3, 4, 5])
        length = rng.choice([30000, 50000, 80000])
    seed = rng.randrange(1 << 30)
    cyc = rng.random() < 0.4
    kw = dict(seq_div=rng.choice([0.002, 0.01, 0.03]), hor_div=rng.choice([0.02, 0.08]))
    p = dict(n=n, length=length, seed=seed, cyclize=cyc, budget=rng.choice([60000, 150000, 400000] if big else [2000, 8000, 30000]), workers=rng.choice([1, 3]), **kw)
    if more:   # (--more-switches: the CLI's -m, the hidden -g 1 = Anchorer::Sparse over ChainMerge structures, and the developer switch that skips the calibration)
        p["max_count"] = rng.choice([3000, 3000, 300, 50])
        if not cyc:
            p["skip_calibration"] = rng.random() < 0.2
            p["chaining_algorithm"] = 1 if rng.random() < 0.2 else None
    if cyc:
        p["dup"] = rng.choice([8000, 15000] if big else [1500, 3000, 5000])
        p["carriers"] = sorted(rng.sample(range(n), rng.randrange(1, n + 1)))
        p["min_cyclizing_length"] = rng.choice([4000, 6000] if big else [1000, 2500])
        seqs = synth.tandem_dup_sequences(seed, length, n, p["dup"], carriers=p["carriers"], **kw)
    else:
        seqs = synth.hor_sequences(seed, length, n, **kw)
    if min(len(s) for s in seqs) < 300:
        return None
    names = ["q%02d" % i for i in range(n)]
    tree = random_tree(rng, names) if rng.random() < 0.6 else msa.balanced_tree(names)
    return p, names, seqs, msa.newick(tree) + ";"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--json", default=None)
    ap.add_argument("--more-switches", action="store_true", help="also vary max_count, skip_calibration and the chaining algorithm")
    ap.add_argument("--big", action="store_true", help="2-5 sequences of 30-80 kbp, budgets 60 000-400 000 (minutes of reference time per case)")
    ap.add_argument("--jobs", type=int, default=max(2, min(24, (os.cpu_count() or 4) - 4)), help="reference runs side by side (one core each)")
    ap.add_argument("--ref-timeout", type=float, default=400.0, help="a reference run that takes longer is dropped (counted)")
    args = ap.parse_args()
    if not os.path.exists(REF_CLI):
        sys.exit("oracle/_ref/ref_cli is missing (python __graft_entry__.py build in the build container)")
    rng = random.Random(args.seed)
    ctx = capi.Context(0)
    cases, bad, undefined, ref_s, own_s, bases, n_cyc, bonds, regions, ref_timeouts, stood = 0, [], [], 0.0, 0.0, 0, 0, 0, 0, 0, 0
    ref_failed = []
    t_end = time.time() + args.seconds

    def reference(c, perturb=None):
        """the reference on one case, in its own directory (a worker thread of the pool below: the CLI is one core each)"""
        p, names, seqs, newick = c
        env = dict(os.environ)
        if perturb:
            env["MALLOC_PERTURB_"] = perturb
        d = tempfile.mkdtemp(prefix="fuzz_msa_")
        synth.write_fasta(os.path.join(d, "in.fa"), seqs, names)
        open(os.path.join(d, "t.nwk"), "w").write(newick + "\n")
        over = "i:max_num_match_pairs=%d" % p["budget"]
        if "max_count" in p:
            over += ";i:max_count=%d" % p["max_count"]
        if p.get("skip_calibration"):
            over += ";b:skip_calibration=1"
        if p.get("chaining_algorithm") is not None:
            over += ";i:chaining_algorithm=%d" % p["chaining_algorithm"]
        if p["cyclize"]:
            over += ";b:cyclize_tandem_duplications=1;i:min_cyclizing_length=%d" % p["min_cyclizing_length"]
        t0 = time.time()
        try:
            r = subprocess.run([REF_CLI, "in.fa", "t.nwk", "-", "out.txt", "0", "0", "0", over], cwd=d, capture_output=True, text=True, timeout=args.ref_timeout, env=env)
            rc, err = r.returncode, r.stderr[-300:]
        except subprocess.TimeoutExpired:
            rc, err = -9, "timeout"
        return c, d, rc, err, time.time() - t0

    from concurrent.futures import ThreadPoolExecutor, FIRST_COMPLETED, wait
    pool = ThreadPoolExecutor(args.jobs)
    pending = set()
    while True:
        while len(pending) < args.jobs and time.time() < t_end:
            c = one_case(rng, args.more_switches, args.big)
            if c is not None:
                pending.add(pool.submit(reference, c))
        if not pending:
            break
        done, pending = wait(pending, return_when=FIRST_COMPLETED)
        for fut in done:
            (p, names, seqs, newick), d, rc, err, secs = fut.result()
            ref_s += secs
            if rc == 0:
                want = open(os.path.join(d, "out.txt"), "rb").read()
                fasta = open(os.path.join(d, "in.fa")).read()
            for f in os.listdir(d):
                os.remove(os.path.join(d, f))
            os.rmdir(d)
            if rc == -9:
                ref_timeouts += 1
                continue
            if rc != 0:
                print("reference failed (%d) on %s: %s" % (rc, json.dumps(p), err), flush=True)
                rec = dict(p, returncode=rc, newick=newick)
                try:   # (what the library does with the input the reference died on)
                    fa_text = "".join(">%s\n%s\n" % (nm, sq) for nm, sq in zip(names, seqs))
                    got, st = ctx.msa(fa_text, newick=newick, max_num_match_pairs=p["budget"], workers=p["workers"], cyclize=p["cyclize"], min_cyclizing_length=p.get("min_cyclizing_length"),
                                      max_count=p.get("max_count", 3000))
                    rec["library"] = dict(completed=True, bytes=len(got), n_bonds=int(st.get("n_bonds", 0)), n_polished_regions=int(st.get("n_polished_regions", 0)))
                except Exception as e:   # noqa: BLE001
                    rec["library"] = dict(completed=False, error=repr(e))
                ref_failed.append(rec)
                continue
            capi.fallback_counters(reset=True)
            t0 = time.time()
            try:
                got, st = ctx.msa(fasta, newick=newick, max_num_match_pairs=p["budget"], workers=p["workers"], cyclize=p["cyclize"],
                                  min_cyclizing_length=p.get("min_cyclizing_length"), max_count=p.get("max_count", 3000), skip_calibration=bool(p.get("skip_calibration")),
                                  chaining_algorithm=p.get("chaining_algorithm"))
            except Exception as e:   # noqa: BLE001
                got, st = b"", {}
                print("library failed on %s: %r" % (json.dumps(p), e), flush=True)
            own_s += time.time() - t0
            cases += 1
            bases += sum(len(s) for s in seqs)
            n_cyc += int(p["cyclize"])
            bonds += int(st.get("n_bonds", 0))
            regions += int(st.get("n_polished_regions", 0))
            stood += int(capi.fallback_counters()["bond_trims_past_the_end"] > 0)
            if got != want:
                p["want_sha256"], p["got_sha256"], p["newick"] = hashlib.sha256(want).hexdigest(), hashlib.sha256(got).hexdigest(), newick
                trims = int(capi.fallback_counters()["bond_trims_past_the_end"])
                verdict = "MISMATCH"
                if p["cyclize"] and trims:
                    # the run stood where the reference's result is undefined (Bonder::trim_partition_ends reads one element past a vector, include/centrolign_amd.h:
                    # cl_fallback_stats): does the REFERENCE's own text depend on what the heap holds?  The unmodified binary again, glibc filling freed / fresh heap
                    # memory with other bytes (MALLOC_PERTURB_), and once more as it is
                    shas = {p["want_sha256"]}
                    for perturb in ("1", "85", "170", "255", None):
                        (_, d2, rc2, _, secs2) = reference((p, names, seqs, newick), perturb)
                        ref_s += secs2
                        if rc2 == 0:
                            shas.add(hashlib.sha256(open(os.path.join(d2, "out.txt"), "rb").read()).hexdigest())
                        for f in os.listdir(d2):
                            os.remove(os.path.join(d2, f))
                        os.rmdir(d2)
                    if len(shas) == 1:
                        # the same text under every fill byte: what the reference reads past its vector may be LIVE data of its neighbours on the heap.  Another program
                        # round the same unmodified objects — the recorded flow of oracle/ref_driver.cpp, whose dumps make other allocations in between — has another heap
                        d3 = tempfile.mkdtemp(prefix="fuzz_msa_")
                        synth.write_fasta(os.path.join(d3, "in.fa"), seqs, names)
                        open(os.path.join(d3, "t.nwk"), "w").write(newick + "\n")
                        over3 = "i:min_cyclizing_length=%d;i:max_num_match_pairs=%d" % (p["min_cyclizing_length"], p["budget"]) + (";i:max_count=%d" % p["max_count"] if "max_count" in p else "")
                        code = ("import sys; sys.path.insert(0, %r); from oracle import pyoracle as po; po.ref_cyclize_dump('in.fa', 't.nwk', 'd.bin', 'out.txt', %r)" % (ROOT, over3))
                        try:
                            r3 = subprocess.run([sys.executable, "-c", code], cwd=d3, capture_output=True, text=True, timeout=args.ref_timeout)
                            if r3.returncode == 0:
                                shas.add(hashlib.sha256(open(os.path.join(d3, "out.txt"), "rb").read()).hexdigest())
                                p["reference_recorded_flow_text"] = hashlib.sha256(open(os.path.join(d3, "out.txt"), "rb").read()).hexdigest()
                        except subprocess.TimeoutExpired:
                            pass
                        for f in os.listdir(d3):
                            os.remove(os.path.join(d3, f))
                        os.rmdir(d3)
                    p["reference_texts_under_other_heap_fill_bytes"] = sorted(shas)
                    p["library_text_is_one_of_them"] = p["got_sha256"] in shas
                    if len(shas) > 1:
                        verdict = "REFERENCE UNDEFINED (its own text changes with the heap's fill byte / from run to run)"
                        undefined.append(p)
                if verdict == "MISMATCH":
                    bad.append(p)
                print("%s %s" % (verdict, json.dumps(p)), flush=True)
    pool.shutdown()
    out = dict(seed=args.seed, seconds=args.seconds, cases=cases, mismatches=len(bad), bases=bases, cases_with_c=n_cyc, bonds=bonds, polished_regions=regions,
               reference_cpu_s=round(ref_s, 1), reference_runs_dropped_at_their_time_limit=ref_timeouts, reference_jobs=args.jobs, library_s=round(own_s, 1),
               cases_that_stood_where_the_reference_reads_past_a_vector=stood, cases_in_which_the_reference_is_undefined=len(undefined), mismatch_cases=bad, reference_undefined_cases=undefined,
               reference_runs_that_ended_in_a_signal_or_error=ref_failed)
    print(json.dumps(out), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
