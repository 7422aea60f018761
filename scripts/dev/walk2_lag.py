"""device time of the 2 x 1 Mbp chaining DPs over far lag / window / near split (every variant in a process of its own)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = []
for qpt in ("1", "2"):
    for lag in ("2", "3", "4"):
        for split in ("1", "0"):
            V.append(("q%s lag%s split%s" % (qpt, lag, split), {"CL_CHAIN_WALK2_QPT": qpt, "CL_CHAIN_FAR_LAG": lag, "CL_CHAIN_NEAR_SPLIT": split}))
V.append(("walk1 lag2 split1", {"CL_CHAIN_WALK2": "0"}))
V.append(("walk1 lag2 split0", {"CL_CHAIN_WALK2": "0", "CL_CHAIN_NEAR_SPLIT": "0"}))
for kind in ("affine", "sparse"):
    for name, env in V:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "far_ab_child.py"), "/tmp/walk2_ab_input.npz", kind], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        m = re.search(r"dp=(\w{8}).*chain=(\w{8}).*device_ms=([\d.]+)", r.stdout)
        print(kind, name, m.groups() if m else "FAILED " + (r.stdout + r.stderr)[-500:], flush=True)
