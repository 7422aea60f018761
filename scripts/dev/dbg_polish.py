import sys, os, re
sys.path.insert(0, os.getcwd())
import numpy as np
from centrolign_amd import capi
z = np.load("tests/golden/cyclize_flow.npz")
name = "tri16k"
d = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + ".")}
t = d["simplified.tableau"]
g = capi.BaseGraph(*[d["simplified." + k] for k in capi.GRAPH_KEYS], int(t[0]), int(t[1]))
text = d["output"].tobytes()
path_names = re.findall(r"^P\t(\S+)", text.decode(), re.M)
ctx = capi.Context(0)
got, n = ctx.polish_cyclized_graph(g, path_names, ["s0", "s1", "s2"], float(d["score_scale"][0]), max_num_match_pairs=40000)
print("regions", n, "nodes", len(got.label), "want", len(d["polished.label"]))
