"""BASELINE configs[1] end to end at the Core::align seam with the real reference: 2 x 1 Mbp synthetic HOR pair (seed 7),
the reference's own match finding and calibration, then Core::align by the unmodified reference vs cl_core_align"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "oracle", "_ref", "adapter_demo")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "in.fa")
    synth.write_fasta(fa, synth.hor_sequences(7, L, 2))
    t0 = time.time()
    p = subprocess.run([DEMO, fa, "-", "1250000", "core"], capture_output=True, text=True, timeout=1500)
    print(p.stdout, p.stderr[-500:], "total %.0f s" % (time.time() - t0))
