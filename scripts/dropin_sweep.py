"""fuzzing with the real reference: oracle/_ref/adapter_demo in 'core' mode (Core::align of the unmodified reference vs
cl_core_align, merge after merge) on fresh synthetic inputs; prints every disagreement"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrolign_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "oracle", "_ref", "adapter_demo")
lo, hi = int(sys.argv[1]), int(sys.argv[2])
hard = len(sys.argv) > 3 and sys.argv[3] == "hard"   # diverged sequences: several segments, unalignable gaps, heuristic routes
bad = 0
merges = 0
t0 = time.time()
for seed in range(lo, hi):
    n = 2 if seed % 3 == 0 else 4
    length = (15000, 25000, 40000, 60000)[seed % 4]
    budget = (20000, 60000, 150000)[seed % 3]
    kw = dict(seq_div=(0.005, 0.01, 0.03)[seed % 3], hor_div=(0.02, 0.03, 0.05)[(seed // 3) % 3], indel_hor=(1, 2, 4)[(seed // 2) % 3])
    if hard:
        n = (2, 4, 3)[seed % 3]
        length = (30000, 50000, 80000)[(seed // 3) % 3]
        budget = (60000, 20000, 150000)[(seed // 2) % 3]
        kw = dict(seq_div=(0.06, 0.1, 0.15, 0.03)[seed % 4], hor_div=(0.05, 0.1, 0.02)[(seed // 4) % 3], indel_hor=(3, 6, 10)[(seed // 5) % 3])
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "in.fa")
        synth.write_fasta(fa, synth.hor_sequences(seed, length, n, **kw))
        nwk = "-"
        if n == 3:
            nwk = os.path.join(d, "t.nwk")
            open(nwk, "w").write("((seq0,seq1),seq2);")
        if n == 4:
            nwk = os.path.join(d, "t.nwk")
            open(nwk, "w").write("((seq0,seq1),(seq2,seq3));" if seed % 2 else "(((seq0,seq1),seq2),seq3);")
        try:
            p = subprocess.run([DEMO, fa, nwk, str(budget), "core"], capture_output=True, text=True, timeout=900)
        except subprocess.TimeoutExpired:
            print("seed", seed, "TIMEOUT", flush=True)
            continue
    ok = p.returncode == 0 and "DROP-IN OK" in p.stdout
    merges += p.stdout.count("GPU Core::align")
    if not ok:
        bad += 1
        print("seed", seed, n, length, budget, kw, "FAILED rc", p.returncode, "\n", p.stdout[-600:], p.stderr[-400:], flush=True)
    else:
        print("seed", seed, n, length, budget, "ok:", p.stdout.strip().splitlines()[-2], flush=True)
print("runs", hi - lo, "merges", merges, "failures", bad, "in %.0f s" % (time.time() - t0))
