"""BASELINE configs[4] (50 sequences x 5 Mbp, -c) walked up from small sizes: one step per invocation, one JSON record per step.

  python scripts/configs4_walk.py N LENGTH [--cyclize] [--workers W] [--budget B] [--dup D] [--json OUT] [--log LOG]

A step is the whole flow on N synthetic HOR arrays of LENGTH bases (seed 43; with --cyclize a third of them carry a recent tandem duplication of D bases,
as tests/golden/make_cyclize_wide.py's cases), balanced guide tree, default parameters unless --budget is given:
  * without --cyclize: msa.progressive_msa merge by merge (the Python driver keeps per-merge times), under CL_CHAIN_TIMING=1 with stderr in LOG;
  * with --cyclize: cl_msa (the library's whole CLI flow: calibration with the tandem-duplication rounds, merges, apply_bonds + polishing).
What the record holds: wall-clock by phase and per merge; the device memory the contexts held per chaining DP against scripts/memory_model.py (from the log);
cl_fallback_counters (strips re-run anti-diagonal-wise, walks re-run per block) — a clean run has zeros; the property checks of tests/test_c3_full.py
(every path of the root spells its input; paths == names; the GFA parses back to a graph with the same paths); with --twice the same text from W and
from 1 worker.  The script never touches the oracle: nothing here is a parity claim beyond those properties (the reference does not finish these sizes
on a box's host within a lease: DESIGN.md section 6b)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("length", type=int)
    ap.add_argument("--cyclize", action="store_true")
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--budget", type=int, default=1250000)
    ap.add_argument("--dup", type=int, default=0, help="duplicated bases of the carriers (default: 3 %% of the length, at least 3000)")
    ap.add_argument("--min-cyclizing-length", type=int, default=None)
    ap.add_argument("--carriers", type=int, default=0, help="how many sequences carry the tandem duplication (default: every third)")
    ap.add_argument("--twice", action="store_true", help="run again with one worker and compare the text")
    ap.add_argument("--json", default=None)
    ap.add_argument("--log", default=None, help="stderr of the library (CL_CHAIN_TIMING=1) goes here; default gpurun_out/configs4_N_LENGTH[_c].log")
    args = ap.parse_args()
    tag = "%dx%d%s" % (args.n, args.length, "_c" if args.cyclize else "")
    log = args.log or os.path.join(ROOT, "gpurun_out", "configs4_%s.log" % tag)
    os.makedirs(os.path.dirname(log), exist_ok=True)
    os.environ["CL_CHAIN_TIMING"] = "1"
    if args.cyclize and args.workers == 1:
        os.environ["CL_POLISH_TIMING"] = "1"
    # the library prints with fprintf(stderr): point fd 2 at the log before it is loaded, keep Python's own messages on the old stderr
    sys.stderr.flush()
    keep = os.dup(2)
    fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 2)
    say = os.fdopen(keep, "w", buffering=1)

    from centrolign_amd import capi, msa, synth
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import memory_model

    rec = dict(step=tag, n=args.n, length=args.length, cyclize=bool(args.cyclize), workers=args.workers, max_num_match_pairs=args.budget, log=os.path.relpath(log, ROOT))
    t0 = time.time()
    dup = args.dup or max(3000, args.length * 3 // 100)
    carriers = sorted(set(range(1, args.n, 3)))
    if args.carriers:
        carriers = carriers[:args.carriers]
    if args.cyclize:
        seqs = synth.tandem_dup_sequences(43, args.length, args.n, dup, carriers=carriers, hor_div=0.08)
        rec["workload"] = "tandem_dup_sequences(seed 43, %d, %d, dup %d, carriers %s, hor_div 0.08), balanced tree" % (args.length, args.n, dup, carriers)
    else:
        seqs = synth.hor_sequences(43, args.length, args.n)
        rec["workload"] = "hor_sequences(seed 43, %d, %d), balanced tree" % (args.length, args.n)
    names = ["q%02d" % i for i in range(args.n)]
    tree = msa.balanced_tree(names)
    newick = msa.newick(tree) + ";"
    rec["input_sha256"] = hashlib.sha256("".join(seqs).encode()).hexdigest()
    rec["synth_s"] = time.time() - t0
    rec["predicted"] = memory_model.predict(args.n, args.length, args.budget)
    say.write("[configs4] %s: inputs made in %.1f s; model says the root's DPs hold %s\n" % (tag, rec["synth_s"], {k: "%.1f GB" % (v / 1e9) for k, v in rec["predicted"]["dp_bytes"].items()}))

    ctx = capi.Context(0)
    capi.fallback_counters(reset=True)
    by_name = dict(zip(names, seqs))

    def run(workers):
        t = time.time()
        if args.cyclize:
            fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
            text, st = ctx.msa(fasta, newick=newick, max_num_match_pairs=args.budget, cyclize=True, min_cyclizing_length=args.min_cyclizing_length, workers=workers)
            return text, dict(wall_s=time.time() - t, **{k: (float(v) if isinstance(v, float) else int(v)) for k, v in st.items()}), None
        r = msa.progressive_msa(ctx, by_name, tree, max_num_match_pairs=args.budget, workers=workers, verbose=False)
        wall = time.time() - t
        text = capi.write_gfa(r["root"], r["paths"])
        st = dict(wall_s=wall, scale=r["scale"], timeline_s=r["stats"]["timeline_s"], per_merge=r["stats"]["per_merge"],
                  match_ms=r["stats"]["match_ms"], align_ms=r["stats"]["align_ms"], fuse_ms=r["stats"]["fuse_ms"])
        return text, st, r

    text, st, r = run(args.workers)
    rec["run"] = st
    rec["gfa"] = dict(bytes=len(text), sha256=hashlib.sha256(text).hexdigest())
    rec["memory"] = ctx.memory_stats()
    rec["fallbacks"] = capi.fallback_counters()
    say.write("[configs4] %s: %.1f s, GFA %d bytes, fallbacks %s, context peak %.1f GB\n" % (tag, st["wall_s"], len(text), rec["fallbacks"], rec["memory"]["peak_bytes"] / 1e9))

    # ---- property checks (tests/test_c3_full.py:62-75) ----
    checks = {}
    # on the TEXT that was printed: read it back as a restart would (cl_read_gfa) and walk every path
    from tests.test_c3_full import spelled
    g, paths = capi.read_gfa(text)
    checks["paths_are_the_names"] = sorted(paths) == sorted(names)
    checks["every_path_spells_its_input"] = all(spelled(g, i) == by_name[nm] for i, nm in enumerate(paths))
    if r is not None:
        checks["gfa_round_trip_same_node_count"] = len(g.label) == len(r["root"].label)
    rec["checks"] = checks
    if args.twice and args.workers != 1:
        text1, st1, _ = run(1)
        rec["run_one_worker"] = dict(wall_s=st1["wall_s"], same_text=text1 == text)
        rec["fallbacks_after_both"] = capi.fallback_counters()
    ctx.close()

    # ---- the memory model against what the contexts held (from the log) ----
    sys.stderr.flush()
    try:
        rows = memory_model.check([log])
        big = [x for x in rows if x.get("held", 0) > 256e6]
        rec["memory_model"] = dict(n_dps=len(rows), n_above_256MB=len(big),
                                   ratio_min=min((x["model_over_held"] for x in big), default=None), ratio_max=max((x["model_over_held"] for x in big), default=None),
                                   largest=max(big, key=lambda x: x["held"]) if big else None)
    except Exception as e:   # noqa: BLE001
        rec["memory_model"] = dict(error=repr(e))
    with open(log, "rb") as f:
        body = f.read().decode("latin1")
    rec["log_lines"] = body.count("\n")
    rec["log_mentions"] = dict(walk_stalled=body.count("walk kernel stalled"), strip_fallback=body.count("anti-diagonal kernel"), error=body.lower().count("error"))
    rec["ok"] = all(checks.values()) and rec["fallbacks"]["strip_fallbacks"] == 0 and rec["fallbacks"]["walk_stalls"] == 0
    out = json.dumps(rec, indent=1, default=str)
    if args.json:
        with open(args.json, "w") as f:
            f.write(out + "\n")
    say.write(out + "\n")
    print(json.dumps(dict(step=tag, ok=rec["ok"], wall_s=st["wall_s"], gfa=rec["gfa"], fallbacks=rec["fallbacks"], checks=checks, memory_model=rec["memory_model"])))


if __name__ == "__main__":
    main()
