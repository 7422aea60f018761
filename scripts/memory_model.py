"""Device-memory model of the chaining DP — the only part of a merge whose device footprint grows with the number of paths — and its check
against what a context really held (cl_context_memory / the line CL_CHAIN_TIMING=1 prints per DP).

  python scripts/memory_model.py predict [n_seq length_bp budget]      # the root merge of an n-sequence MSA, no GPU needed
  python scripts/memory_model.py check LOG [LOG ...] [--json OUT]      # logs of runs under CL_CHAIN_TIMING=1: model against measurement for every DP

All sizes in bytes.  M match pairs (at most max_num_match_pairs: the budget caps it whatever the sequence length), R records (a pair has one record
in the combination of its own two chains... or more on graphs whose nodes lie on several paths), C chain combinations (paths of graph 1 x paths of graph 2),
T1 / T2 chains."""
import json
import re
import sys

MACRO, BLOCK = 1024, 256


def chain_dp_bytes(M, R, C, T1, T2, sparse, far_pad=0, far_levels=0, walk="1", factored=True):
    """device bytes a chaining DP holds between its uploads and its traceback (cl_chain_api.cpp: chain_dp_batch)"""
    n_blocks = (M + BLOCK - 1) // BLOCK
    per_pair = 12 + 4 + 16                       # weight, start value, dp; record offsets; group, group end, group base, group total
    per_pair += 8 if (C > 16 or walk == "fold") and C > 1 else 0   # atomic maximum + arrival count of the walk's exchange
    per_record = 16 + 28 + 8                      # pair, insertion index, offset, shift; 7 stored values; combination and position of the record
    per_combo = 4 * (n_blocks + 1) + 104 + 8 * MACRO + (8 * MACRO + 64 * MACRO if walk == "2" else 0)
    per_pair_combo = 12 + 28 + 4                  # query (insertion bound, offset bound, shift); 7 running maxima; own record
    b = M * per_pair + R * per_record + C * per_combo + M * C * per_pair_combo
    if factored:
        b += (T1 + T2) * M * 4 * (1 if sparse else 2) + 8 * C
    if far_pad:
        sides = 1 if sparse else 2
        b += 48 * far_pad + 7 * 4 * far_pad       # record image of the far pass; seven scratch arrays of the sorts
        b += far_levels * sides * 4 * far_pad     # the static orders per level
        arena = 0
        for lv in range(far_levels):
            arena += sides * (2 * far_pad + sum((far_pad >> (3 * (j + 1))) + 8 for j in range(lv + 1)))
        b += 4 * arena
        b += 8 * far_pad + (1 << 20)              # radix-sort scratch
    return b


def predict(n_seq, length, budget):
    half1, half2 = (n_seq + 1) // 2, n_seq // 2
    C, M = half1 * half2, budget
    rows = []
    for sparse in (True, False):
        # records: the gap-free DP keeps one per pair; the affine DP one per pair and chain combination the pair's two nodes lie on.  On HOR arrays of 0.5 %
        # divergence most nodes of a merged graph lie on most of its paths: measured 50 x 100 kbp (profiles/r05_configs4.json) records / (pairs x combinations) =
        # 0.95 (4 combinations), 0.90 (9-12), 0.84 (36-42), 0.75 (156), 0.64 (625) ~ 1.03 x combinations^-0.075
        # (at 50 x 1 Mbp the root keeps more: 0.79-0.89 of pairs x combinations, 617-696 M records in four runs — longer arrays, fewer pairs per node under the same budget)
        R = M if sparse else int(M * C * min(1.0, (1.03 * C ** -0.075) if length < 500000 else max(0.89, 1.03 * C ** -0.075) if C > 100 else 1.03 * C ** -0.075))
        levels = 1
        while levels < 4 and (64 << (3 * levels)) <= max(1, R // max(1, C)) * 4:   # (kFarMaxLevels = 4)
            levels += 1
        pad = R + C * (64 << (3 * (levels - 1))) // 2
        # the far pass runs whenever its arena fits 2^35 words and the device has the room (cl_chain_api.cpp; up to round 4 the arena was addressed in words: off beyond
        # 2^32, i.e. at the 625-combination root of 50 sequences, which then swept all pairs: 196 s instead of 36)
        far = True   # (taken at a wide root only where the context's DPs prune — cl_chain_api.cpp, far_last_choice — and where the device has the room: this is the upper figure)
        rows.append(("gap-free" if sparse else "affine", chain_dp_bytes(M, R, C, half1, half2, sparse, pad if far else 0, levels if far else 0, "fold" if C > 256 else "1")))
    return dict(n_seq=n_seq, length=length, budget=budget, combinations=C, dp_bytes=dict(rows),
                host_tables_bytes=2 * 2 * 4 * int(length * 1.2) * half1 + 2 * 4 * int(length * 1.2) * half1)   # PathMerge index + table per side, post-switch distances


PAIRS = re.compile(r"\[chain_dp_batch\]\s+(\d+) pairs, (\d+) combinations, (\d+) records")
HELD = re.compile(r"device memory held by the context: ([\d.]+) MB \((\w+) queries, far pass (\w+): (\d+) padded records in (\d+) levels, (\d+) \+ (\d+) tags, walk (\w+)\)")
KIND = re.compile(r"\[cl_anchor_chain\] (sparse|affine): chain DP")


def check(paths):
    out = []
    for path in paths:
        cur = None
        for ln in open(path):
            m = PAIRS.search(ln)
            if m:
                cur = dict(pairs=int(m.group(1)), combinations=int(m.group(2)), records=int(m.group(3)))
                continue
            m = HELD.search(ln)
            if m and cur:
                cur.update(held=float(m.group(1)) * 1048576, factored=m.group(2) == "factored", far=m.group(3) == "on", far_pad=int(m.group(4)), far_levels=int(m.group(5)),
                           t1=int(m.group(6)), t2=int(m.group(7)), walk=m.group(8))
                continue
            m = KIND.search(ln)
            if m and cur and "held" in cur:
                cur["sparse"] = m.group(1) == "sparse"
                if cur["factored"] and cur["walk"] != "off":
                    cur["model"] = chain_dp_bytes(cur["pairs"], cur["records"], cur["combinations"], cur["t1"], cur["t2"], cur["sparse"], cur["far_pad"], cur["far_levels"], cur["walk"])
                    cur["model_over_held"] = cur["model"] / cur["held"]
                    cur["log"] = path
                    out.append(cur)
                cur = None
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        args = [a for a in sys.argv[2:] if not a.startswith("--")]
        logs = [a for a in args if a not in sys.argv[sys.argv.index("--json") + 1:sys.argv.index("--json") + 2]] if "--json" in sys.argv else args
        rows = check(logs)
        big = sorted(rows, key=lambda r: -r["held"])[:12]
        for r in big:
            print("%9d pairs %4d combinations %s: held %8.1f MB, model %8.1f MB (x %.3f)" % (r["pairs"], r["combinations"], "gap-free" if r["sparse"] else "affine  ",
                                                                                       r["held"] / 2**20, r["model"] / 2**20, r["model_over_held"]))
        res = dict(n_dps=len(rows), largest=big, worst_ratio_among_dps_over_256MB=[min((r["model_over_held"] for r in rows if r["held"] > 2**28), default=None),
                                                                                  max((r["model_over_held"] for r in rows if r["held"] > 2**28), default=None)])
        print(json.dumps(res["worst_ratio_among_dps_over_256MB"]))
        if "--json" in sys.argv:
            json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    else:
        a = [int(x) for x in sys.argv[2:5]] if len(sys.argv) > 4 else None
        for cfg in ([a] if a else [(10, 1000000, 1250000), (50, 5000, 1250000), (50, 5000000, 1250000)]):
            p = predict(*cfg)
            print("%2d x %8d bp, budget %d: root %4d combinations; DP device bytes gap-free %.1f GB, affine %.1f GB; host reachability tables of the root ~%.1f GB" %
                  (p["n_seq"], p["length"], p["budget"], p["combinations"], p["dp_bytes"]["gap-free"] / 1e9, p["dp_bytes"]["affine"] / 1e9, p["host_tables_bytes"] / 1e9))
