"""Where the -c flow's polishing step spends its time: cl_msa(cyclize) on one of the wide goldens (tests/golden/cyclize_16x12k.json / cyclize_50x8k.json) with one worker under
CL_POLISH_TIMING=1 CL_CHAIN_TIMING=1, the phase lines of the thousands of small chaining DPs summed per phase.   usage: python scripts/polish_profile.py [case] [workers]"""
import collections
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"cyclize_16x12k": (16, 41, 12000, 4000, [0, 3, 5, 8, 9, 13], "c"), "cyclize_50x8k": (50, 43, 8000, 3000, [1, 4, 7, 12, 18, 23, 29, 31, 36, 40, 44, 48], "d")}


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "cyclize_50x8k"
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    os.environ["CL_POLISH_TIMING"] = "1"
    os.environ["CL_CHAIN_TIMING"] = "1"
    log = "/tmp/polish_profile_%d.log" % os.getpid()
    sys.stderr.flush()
    keep = os.dup(2)
    fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(fd, 2)
    from centrolign_amd import capi, msa, synth
    n, seed, length, dup, carriers, prefix = CASES[case]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", case + ".json")))
    seqs = synth.tandem_dup_sequences(seed, length, n, dup, carriers=carriers, hor_div=0.08)
    names = ["%s%02d" % (prefix, i) for i in range(n)]
    fasta = "".join(">%s\n%s\n" % (a, b) for a, b in zip(names, seqs))
    tree = msa.newick(msa.balanced_tree(names)) + ";"
    ctx = capi.Context(0)
    ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
    t0 = time.time()
    text, st = ctx.msa(fasta, newick=tree, max_num_match_pairs=gold["max_num_match_pairs"], cyclize=True, min_cyclizing_length=gold["min_cyclizing_length"], workers=workers)
    wall = time.time() - t0
    ctx.close()
    os.dup2(keep, 2)
    print("%s, %d worker(s): %.2f s; cyclize %.2f s, align %.2f s, bonds %.2f s, %d regions" % (case, workers, wall, st["cyclize_s"], st["align_s"], st["bonds_s"], st["n_polished_regions"]))
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for ln in open(log, errors="replace"):
        if ln.startswith("[cl_polish]"):
            print(ln.strip()[:220])
        m = re.match(r"\[(\w+)\]\s+(.*?)\s+([\d.]+) ms\s*$", ln)
        if m:
            k = m.group(1) + " " + re.sub(r"\d+", "#", m.group(2).strip())[:58]
            acc[k] += float(m.group(3)); cnt[k] += 1
    # the small DPs by their number of chain combinations: how many there are, and where the waits for the device go
    buckets = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    cur = None
    for ln in open(log, errors="replace"):
        m = re.match(r"\[chain_dp_batch\]\s+(\d+) pairs, (\d+) combinations, (\d+) records", ln)
        if m:
            cur = (int(m.group(1)), int(m.group(2)), int(m.group(3)))
        m = re.match(r"\[chain_dp_batch\]\s+sync\s+([\d.]+) ms", ln)
        if m and cur:
            c = cur[1]
            b = "1" if c == 1 else "2-8" if c <= 8 else "9-64" if c <= 64 else "65-256" if c <= 256 else "257-768" if c <= 768 else "> 768"
            e = buckets[b]; e[0] += 1; e[1] += float(m.group(1)); e[2] += cur[0]; e[3] += cur[2]
            cur = None
    for b in ("1", "2-8", "9-64", "65-256", "257-768", "> 768"):
        if b in buckets:
            n_, t_, p_, r_ = buckets[b]
            print("  DPs of %-8s combinations: %6d, waiting for the device %8.1f ms (%.3f each), %.0f pairs and %.0f records each" % (b, n_, t_, t_ / n_, p_ / n_, r_ / n_))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:18]:
        print("  %-72s %9.1f ms  x %6d  (%.3f each)" % (k, v, cnt[k], v / cnt[k]))
    os.remove(log)


if __name__ == "__main__":
    main()
