#!/bin/bash
# VALU / SALU / LDS / store instructions and wave-cycles per kernel of the timed stitch step (scripts/step_launches.py): which kernels spend the device's issue slots.
# rocprofv3 --pmc with --kernel-trace only; the program follows "--".  usage (gpurun): bash scripts/dev/step_valu.sh [env assignments are inherited]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/valu
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc -o p -- python3 $R/scripts/step_launches.py --steps 4 --warmup 2 --json $OUT/step.json > /dev/null 2>$OUT/err.txt
cd $R
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "valu")
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("::")[-1].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
print("kernel, launches, VALU wave-instr per launch (M), share of all VALU, SALU/VALU, wave-cycles x4 per VALU instr")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    print("%-34s %4d  %9.2f M  %5.1f %%  salu/valu %.2f  cycles per VALU instr %.1f  active-VALU share of wave cycles %.2f" % (k, n[k], v["SQ_INSTS_VALU"] / max(1, n[k]) / 1e6, 100 * v["SQ_INSTS_VALU"] / tot,
          v["SQ_INSTS_SALU"] / max(1.0, v["SQ_INSTS_VALU"]), 4 * v["SQ_WAVE_CYCLES"] / max(1.0, v["SQ_INSTS_VALU"]), v["SQ_ACTIVE_INST_VALU"] / max(1.0, v["SQ_WAVE_CYCLES"])))
PY
