"""A/B of the walk kernels on the 2 x 1 Mbp chaining input (bench_data/c2_chain_input.npz): every variant in a process of its own (the
switches are read once), same digest expected everywhere, device time of the affine and the gap-free DP printed.
usage (GPU box): python scripts/dev/walk2_ab.py [reps]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tests import far_ab_child  # noqa: E402
from centrolign_amd import capi  # noqa: E402

VARIANTS = [("walk1", {"CL_CHAIN_WALK2": "0"}), ("walk2", {}), ("walk2_w128", {"CL_CHAIN_WALK2_QPT": "1"}), ("walk2_h0", {"CL_CHAIN_WALK2_HELPERS": "0"}),
            ("walk2_h3", {"CL_CHAIN_WALK2_HELPERS": "3"})]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    z = np.load(os.path.join(ROOT, "bench_data", "c2_chain_input.npz"))
    graphs = []
    for side in ("parent1.", "parent2."):
        t = z[side + "tableau"]
        graphs.append(capi.BaseGraph(*[z[side + k] for k in far_ab_child.GRAPH_KEYS], t[0], t[1]))
    ms = capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT})
    path = "/tmp/walk2_ab_input.npz"
    far_ab_child.save_input(path, graphs[0], graphs[1], ms, float(z["score_scale"][0]))
    for kind in ("affine", "sparse"):
        for name, env_extra in VARIANTS:
            env = dict(os.environ, **env_extra)
            for rep in range(reps):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "far_ab_child.py"), path, kind], env=env, capture_output=True, text=True, timeout=900)
                m = re.search(r"RESULT .*", r.stdout)
                print(kind, name, rep, m.group(0) if m else "FAILED rc=%d %s" % (r.returncode, (r.stdout + r.stderr)[-1500:]), flush=True)


if __name__ == "__main__":
    main()
