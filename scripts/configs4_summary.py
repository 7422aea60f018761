"""profiles/r05_configs4.json from the step records of scripts/configs4_walk.py (gpurun_out/r5*/c4_*.json): one entry per step with the wall-clock by phase, the merges by
shape (paths x paths: count, mean match / chain / device / stitch ms), the device memory of the largest chaining DP against scripts/memory_model.py, the second attempts the
device path made by itself (cl_fallback_counters) and the property checks.  usage: python scripts/configs4_summary.py OUT.json step.json [step.json ...]"""
import collections
import json
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    steps = []
    for f in files:
        r = json.load(open(f))
        run = r["run"]
        e = dict(step=r["step"], workload=r["workload"], input_sha256=r["input_sha256"], workers=r["workers"], max_num_match_pairs=r["max_num_match_pairs"],
                 ok=r["ok"], wall_s=run["wall_s"], gfa=r["gfa"], checks=r["checks"], fallbacks=r["fallbacks"],
                 context_peak_GB=r["memory"]["peak_bytes"] / 1e9, device_total_GB=r["memory"]["device_total_bytes"] / 1e9, pinned_host_GB=r["memory"]["pinned_host_bytes"] / 1e9)
        if "per_merge" in run:
            agg = collections.OrderedDict()
            for m in run["per_merge"]:
                a = agg.setdefault("%d+%d" % (m["paths1"], m["paths2"]), dict(merges=0, match_ms=0.0, chain_ms=0.0, chain_device_ms=0.0, stitch_ms=0.0, combinations=m["chain_combinations"], nodes_max=0))
                a["merges"] += 1
                for k in ("match_ms", "chain_ms", "chain_device_ms", "stitch_ms"):
                    a[k] += m[k]
                a["nodes_max"] = max(a["nodes_max"], m["nodes"])
            for a in agg.values():
                for k in ("match_ms", "chain_ms", "chain_device_ms", "stitch_ms"):
                    a[k] = round(a[k] / a["merges"], 1)
            e["merges_by_shape_mean_ms"] = agg
            e["timeline_s"] = run["timeline_s"]
        else:
            e["phases"] = {k: run[k] for k in ("calibration_s", "bonds_s", "match_s", "align_s", "fuse_s", "cyclize_s", "total_s") if k in run}
            e["n_bonds"], e["n_polished_regions"] = run.get("n_bonds"), run.get("n_polished_regions")
        mm = r.get("memory_model") or {}
        if mm.get("largest"):
            lg = mm["largest"]
            e["largest_dp"] = dict(pairs=lg["pairs"], combinations=lg["combinations"], records=lg["records"], held_GB=lg["held"] / 1e9, model_GB=lg["model"] / 1e9,
                                   model_over_held=lg["model_over_held"], walk=lg["walk"], far_pass=lg["far"])
            if r["workers"] == 1:
                e["model_over_held_range_above_256MB"] = [mm["ratio_min"], mm["ratio_max"]]
        e["predicted_root_dp_GB"] = {k: v / 1e9 for k, v in r["predicted"]["dp_bytes"].items()}
        steps.append(e)
    json.dump(dict(steps=steps, script="scripts/configs4_walk.py (one invocation per step; logs under CL_CHAIN_TIMING=1 are not kept: hundreds of MB)",
                   note="BASELINE configs[4] (50 x 5 Mbp, -c) walked up on ONE MI355X: the chaining DP's footprint and time are set by the match-pair budget (1.25 M), not by the "
                        "sequence length, so 50 x 1 Mbp already holds the root merge of the full size: 625 chain combinations, ~0.7 G records, ~100 GB on the device"),
              open(out, "w"), indent=1)
    for e in steps:
        print("%-16s ok=%s %7.1f s  fallbacks %s  largest DP %s" % (e["step"], e["ok"], e["wall_s"], e["fallbacks"], e.get("largest_dp")))


if __name__ == "__main__":
    main()
