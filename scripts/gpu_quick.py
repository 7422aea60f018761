"""quick GPU sanity: product vs oracle on random DAG pairs and chains (developer tool, not a test)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centrolign_amd import synth, capi
from oracle import pyoracle as po

ctx = capi.Context(0)
print("device:", ctx.device_name())
ok = True
for seed, max_n in ((1, 12), (2, 40), (3, 90), (4, 300)):
    b = synth.random_dag_batch(400 if max_n < 100 else 60, seed=seed, max_n=max_n)
    r_o = po.oracle_stitch_batch(b)
    r_g = ctx.stitch_batch_align(b)
    d = r_g.same_as(r_o)
    print("subalign seed", seed, "max_n", max_n, "->", d)
    ok &= d is None
    for npw in (1, 2, 3):
        f = np.full(b.n_problems, npw, np.uint8)
        d = ctx.po_poa_batch(b, f, capi.default_stitch_params().alignment_params).same_as(po.oracle_stitch_batch(b, force_num_pw=f))
        print("  po_poa npw", npw, "->", d)
        ok &= d is None
lb = synth.linear_batch([(5, 7), (40, 33), (100, 120), (1, 1), (0, 5), (6, 0), (700, 650), (300, 1500), (2000, 40)], seed=3)
d = ctx.stitch_batch_align(lb).same_as(po.oracle_stitch_batch(lb))
print("linear ->", d); ok &= d is None
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/c2_pair_seed7_intervals.npz"))
t0 = time.time()
seqs = synth.hor_sequences(7, 1000000, 2)
hb = synth.batch_from_intervals(seqs[0], seqs[1], z["intervals"], z["only_del"])
print("C2 batch", hb.n_problems, hb.dp_cells(), "gen %.1fs" % (time.time() - t0))
t0 = time.time(); r_o = po.oracle_stitch_batch(hb); t_or = time.time() - t0
t0 = time.time(); plan = ctx.plan(hb); print("plan create %.3fs" % (time.time() - t0))
print(plan.stats())
for i in range(5):
    plan.execute(); ms = plan.sync()
    print("execute ms", ms, "cells/s %.3g" % (plan.stats()["dp_cells"] / ms * 1e3))
t0 = time.time(); r_g = plan.collect(); print("collect %.3fs" % (time.time() - t0))
d = r_g.same_as(r_o)
print("C2 ->", d, "oracle %.2fs" % t_or); ok &= d is None
if os.environ.get("CL_FORCE_GENERAL") != "1":
    big = synth.linear_batch([(3000, 2900), (1500, 5000), (6000, 300), (257, 257), (1025, 64), (64, 1025), (129, 1), (1, 129)], seed=5)
    d = ctx.stitch_batch_align(big).same_as(po.oracle_stitch_batch(big))
    print("big linear ->", d); ok &= d is None
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
