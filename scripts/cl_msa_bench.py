import sys, time, hashlib
sys.path.insert(0, "/root/repo")
from centrolign_amd import capi, synth
names, seqs, _ = synth.c3_workload(1000000)
fasta = "".join(">%s\n%s\n" % (nm, seqs[nm]) for nm in names)
ctx = capi.Context(0)
ctx.find_matches(capi.leaf_graph("ACGTACGTAC"), capi.leaf_graph("ACGTTCGTAC"))
for w in (4, 1):
    t0 = time.time(); text, st = ctx.msa(fasta, synth.C3_NEWICK, workers=w); dt = time.time() - t0
    print("cl_msa workers %d: %.2f s, sha %s, stats %s" % (w, dt, hashlib.sha256(text).hexdigest()[:16], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}))
