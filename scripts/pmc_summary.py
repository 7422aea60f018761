"""summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*) into per-kernel averages: HBM bytes per launch and the VALU / LDS
issue counters.  Writes gpurun_out/hbm_traffic_latest.json (copied to profiles/), which bench.py reads for `roofline.traffic`."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
command = sys.argv[2] if len(sys.argv) > 2 else None
res = defaultdict(dict)
# average duration per kernel from the --stats pass of the same command (prof_bench/*kernel_stats.csv)
duration_ns = {}
for f in glob.glob(os.path.join(out, "prof_bench", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        duration_ns[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]), float(r["Percentage"]))
SQ = ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")
# waves of ONE workgroup per SIMD (launch bounds / 256): what a wave can issue at most is one VALU instruction per 4 cycles divided by the
# waves that share its SIMD, so VALU wave-instructions per resident wave-cycle are measured against 0.25 / that (round 3 divided every kernel by
# 0.25).  Kernels of 256 threads or fewer share a SIMD with waves of OTHER workgroups: for them the counters below (SQ_WAIT_*) say what the
# waves wait for, not this ceiling
WAVES_PER_SIMD = {"chain_walk_kernel": 4, "chain_walk2_kernel": 4, "far_seal_big_kernel": 4, "popoa_sys_kernel<1, 1024>": 4, "popoa_sys_kernel<2, 1024>": 4,
                  "popoa_sys_kernel<3, 1024>": 4, "popoa_linear_kernel<16>": 4}
for d, counters in (("pmc_FETCH_SIZE", ("FETCH_SIZE",)), ("pmc_WRITE_SIZE", ("WRITE_SIZE",)),
                    ("pmc_SQ", SQ)):
    files = glob.glob(os.path.join(out, d, "*counter_collection.csv"))
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] in counters:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, per in acc.items():
        for c, v in per.items():
            res[k][c] = sum(v) / len(v)
            res[k]["launches"] = len(v)
summary, traffic, limiter = {}, {}, {}
for k, v in res.items():
    short = k.split("::")[-1].split("(")[0]
    # FETCH_SIZE / WRITE_SIZE are in KiB of 64-B requests as counted at the L2's memory side; on gfx950 a wide coalesced read stream is
    # tallied at 1/2 (MI355X_MICROARCH.md "HBM"), so the read side is doubled here
    fetch = v.get("FETCH_SIZE", 0.0) * 1024 * 2
    write = v.get("WRITE_SIZE", 0.0) * 1024
    e = {"launches": v.get("launches"), "fetch_bytes_per_launch_x2": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    if "SQ_WAVE_CYCLES" in v:
        # SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles summed over the waves / the SQs; instructions are wave-instructions
        e.update({c.lower(): v[c] for c in SQ if c in v})
        wc = v["SQ_WAVE_CYCLES"] * 4.0
        if wc > 0 and "SQ_INSTS_VALU" in v:
            e["valu_insts_per_wave_cycle"] = v["SQ_INSTS_VALU"] / wc
            e["lds_insts_per_wave_cycle"] = v.get("SQ_INSTS_LDS", 0.0) / wc
            wps = next((n for k2, n in WAVES_PER_SIMD.items() if short.replace("void ", "").startswith(k2)), None)
            if wps:
                e["waves_per_simd"] = wps
                e["valu_frac_of_issue_ceiling"] = e["valu_insts_per_wave_cycle"] / (0.25 / wps)
            # where a resident wave's cycles go (MI355X_MICROARCH.md "rocprofv3 PMC slots": the three are disjoint and add up to SQ_WAVE_CYCLES):
            # issuing (ACTIVE_INST_ANY), ready but not issued (WAIT_INST_ANY: the issue ports are the limit), parked on s_waitcnt / a barrier (WAIT_ANY)
            wcq = v["SQ_WAVE_CYCLES"]
            for name, c in (("wave_cycles_issuing", "SQ_ACTIVE_INST_ANY"), ("wave_cycles_issuing_valu", "SQ_ACTIVE_INST_VALU"), ("wave_cycles_ready_not_issued", "SQ_WAIT_INST_ANY"),
                            ("wave_cycles_parked_on_waitcnt_or_barrier", "SQ_WAIT_ANY")):
                if c in v and wcq > 0:
                    e[name] = v[c] / wcq
            limiter[short] = ("%.3f VALU and %.3f LDS wave-instructions per resident wave-cycle%s; of a resident wave's cycles %s issuing, %s ready but not issued, %s parked on "
                              "s_waitcnt or a barrier" % (e["valu_insts_per_wave_cycle"], e["lds_insts_per_wave_cycle"],
                                                          (" = %.0f %% of what a wave can issue with %d waves per SIMD" % (100 * e["valu_frac_of_issue_ceiling"], wps)) if wps else "",
                                                          *["%.0f %%" % (100 * e[n]) if n in e else "?" for n in ("wave_cycles_issuing", "wave_cycles_ready_not_issued", "wave_cycles_parked_on_waitcnt_or_barrier")]))
    if k in duration_ns:
        avg_ns, calls, pct = duration_ns[k]
        e.update({"avg_duration_us": avg_ns / 1e3, "calls_in_stats_pass": calls, "share_of_kernel_time_pct": pct,
                  "hbm_GBps": (fetch + write) / avg_ns if avg_ns else None, "hbm_frac_of_8TBps": (fetch + write) / avg_ns / 8000.0 if avg_ns else None})
    summary[short] = e
    traffic[short] = fetch + write
print(json.dumps(summary, indent=1))
traffic["_limiter"] = limiter
if command:
    traffic["_command"] = command
    summary["_command"] = command
try:
    import subprocess
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    head = ""
traffic["_commit"] = head or os.environ.get("CL_TREE", "round 4 working tree (no .git on the GPU box)")
summary["_commit"] = traffic["_commit"]
json.dump(traffic, open(os.path.join(out, "hbm_traffic_latest.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
