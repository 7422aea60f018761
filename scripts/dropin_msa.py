"""a deeper MSA through the Core::align drop-in demo: n sequences, balanced guide tree, graphs with up to n/2 + n/2 paths"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "oracle", "_ref", "adapter_demo")
n, L, budget, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])


def balanced(names):
    if len(names) == 1:
        return names[0]
    h = len(names) // 2
    return "(" + balanced(names[:h]) + "," + balanced(names[h:]) + ")"


with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "in.fa")
    synth.write_fasta(fa, synth.hor_sequences(seed, L, n, seq_div=0.01, hor_div=0.03, indel_hor=2))
    nwk = os.path.join(d, "t.nwk")
    open(nwk, "w").write(balanced(["seq%d" % i for i in range(n)]) + ";")
    t0 = time.time()
    p = subprocess.run([DEMO, fa, nwk, str(budget), "core"], capture_output=True, text=True, timeout=2400)
    print(p.stdout, p.stderr[-500:], "total %.0f s" % (time.time() - t0))
