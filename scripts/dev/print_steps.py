"""print ms per step and the longest launches of scripts/step_launches.py records: python scripts/dev/print_steps.py a.json [b.json ...]"""
import json
import sys
for f in sys.argv[1:]:
    try:
        r = json.load(open(f))
    except OSError as e:
        print(f, e)
        continue
    print("%s: %.3f ms per step, %.1f G cells/s" % (f, r["ms_per_step"], r["cells_per_s"] / 1e9))
    for li in sorted(r["launches"], key=lambda x: -x["in_pass_ms"])[:10]:
        print("    %-30s %6d problems %10d cells %8.3f ms  longest %s" % (li["kernel"], li["n_problems"], li["dp_cells"], li["in_pass_ms"], li["longest"]))
