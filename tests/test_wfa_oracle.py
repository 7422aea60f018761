"""CPU suite: the pure-Python restatement of the reference's pruned graph-graph WFA (oracle/wfa_oracle.py; route "w" of Stitcher::do_alignment) against (a) the compiled
reference's own outputs for that route (tests/golden/host_routes.npz, made from Stitcher::subalign) — this is what pins the restatement — and (b) the product's host code
(centrolign_amd/csrc/wfa_host.hpp through cl_host_route_align) on fresh random pairs, which until round 5 was compared with the reference only where oracle/_ref exists
(missing #4)."""
import numpy as np
import pytest

from centrolign_amd import capi, synth
from oracle import pyoracle as po
from oracle import wfa_oracle as wo
from tests.test_host_routes import _golden

ROUTE_W = 5


def oracle_w(batch, k, sp):
    g1, g2 = wo.Side(batch.side[0], k), wo.Side(batch.side[1], k)
    ap = sp.alignment_params
    npw = po.oracle_lib().clo_choose_num_pw(g1.n, g2.n, ap)
    costs = wo.wfa_costs(int(ap.match), int(ap.mismatch), [int(x) for x in ap.gap_open], [int(x) for x in ap.gap_extend], npw)
    pairs = wo.pwfa_po_poa(g1, g2, costs, 2 * int(sp.wfa_pruning_dist))
    out = np.array(pairs, np.uint64).reshape(-1, 2)
    for s in (0, 1):                                                    # translate (src/alignment.cpp:26-39)
        bt = batch.side[s].back_translation
        if bt is not None and len(out):
            base = int(batch.side[s].node_off[k])
            real = out[:, s] != wo.GAP
            out[real, s] = bt[base + out[real, s].astype(np.int64)]
    return out


@pytest.mark.parametrize("tag", ["w_linear", "w_dags"])
def test_the_restatement_gives_the_references_alignments(tag):
    z, batches = _golden()
    batch, params = batches[tag]
    aln_off, pairs = z[tag + ".aln_off"], z[tag + ".pairs"]
    n = 0
    for k in range(batch.n_problems):
        try:
            route, _ = capi.host_route_align(batch, k, params)
        except capi.ClError:
            continue
        if route != ROUTE_W:
            continue
        got = oracle_w(batch, k, params)
        assert np.array_equal(got, pairs[int(aln_off[k]):int(aln_off[k + 1])]), (tag, k)
        n += 1
    assert n >= 3, n


def test_the_products_host_code_gives_the_restatements_alignments():
    sp = capi.default_stitch_params()
    sp.min_wfa_size, sp.max_wfa_size, sp.max_wfa_ratio = 80, 10 ** 9, 3.0
    n = 0
    for seed in (41, 42, 43):
        b = synth.random_dag_batch(80, seed=seed, max_n=70)
        for k in range(b.n_problems):
            try:
                route, got = capi.host_route_align(b, k, sp)
            except capi.ClError:
                continue
            if route != ROUTE_W:
                continue
            assert np.array_equal(got, oracle_w(b, k, sp)), (seed, k)
            n += 1
    assert n >= 20, n


def oracle_ad(batch, k, sp, route):
    g1, g2 = wo.Side(batch.side[0], k), wo.Side(batch.side[1], k)
    ap = sp.alignment_params
    npw = po.oracle_lib().clo_choose_num_pw(g1.n, g2.n, ap)
    costs = wo.wfa_costs(int(ap.match), int(ap.mismatch), [int(x) for x in ap.gap_open], [int(x) for x in ap.gap_extend], npw)
    if route == 3:
        pairs = wo.deletion_wfa_po_poa(g1, g2, costs)
    else:
        pairs = [(b, a) for a, b in wo.deletion_wfa_po_poa(g2, g1, costs)]       # swap_graphs (src/alignment.cpp:41-45)
    out = np.array(pairs, np.uint64).reshape(-1, 2)
    for s in (0, 1):
        bt = batch.side[s].back_translation
        if bt is not None and len(out):
            base = int(batch.side[s].node_off[k])
            real = out[:, s] != wo.GAP
            out[real, s] = bt[base + out[real, s].astype(np.int64)]
    return out


@pytest.mark.parametrize("tag", ["ad_linear", "mixed_dags"])
def test_the_two_sided_deletion_search_gives_the_references_alignments(tag):
    z, batches = _golden()
    batch, params = batches[tag]
    aln_off, pairs = z[tag + ".aln_off"], z[tag + ".pairs"]
    seen = {3: 0, 4: 0}
    for k in range(batch.n_problems):
        try:
            route, _ = capi.host_route_align(batch, k, params)
        except capi.ClError:
            continue
        if route not in (3, 4):
            continue
        got = oracle_ad(batch, k, params, route)
        assert np.array_equal(got, pairs[int(aln_off[k]):int(aln_off[k + 1])]), (tag, k, route)
        seen[route] += 1
    assert seen[3] + seen[4] >= 2, seen


def test_the_containers_iteration_order_is_reproduced():
    """oracle/std_unordered_order.py against the real std::unordered_map: the committed fixture (40 insertion sequences of up to 1 500 keys, several rehashes) and,
    where g++ is present, the same program compiled now"""
    import json
    import os
    import shutil
    from oracle.std_unordered_order import UnorderedKeys
    from tests import helpers as H
    cases = [json.load(open(os.path.join(H.GOLDEN, "std_unordered_order.json")))]
    if shutil.which("g++"):
        from tests.golden import make_std_unordered_order as mk
        cases.append(json.loads(mk.run()))
    for cs in cases:
        assert len(cs) == 40
        for c in cs:
            u = UnorderedKeys()
            for k in c["keys"]:
                u.insert(k)
            assert list(u) == c["order"] and u.n_buckets == c["buckets"]


def test_the_products_deletion_search_gives_the_restatements_alignments():
    sp = capi.default_stitch_params()
    sp.max_trivial_size, sp.deletion_alignment_ratio = 40, 4
    sp.deletion_alignment_short_max_size, sp.deletion_alignment_long_min_size = 60, 50
    n = {3: 0, 4: 0}
    for seed in (51, 52, 53, 54):
        b = synth.random_dag_batch(120, seed=seed, max_n=130)
        b.only_deletion_alns[:] = 1
        for k in range(b.n_problems):
            try:
                route, got = capi.host_route_align(b, k, sp)
            except capi.ClError:
                continue
            if route not in (3, 4):
                continue
            assert np.array_equal(got, oracle_ad(b, k, sp, route)), (seed, k, route)
            n[route] += 1
    assert n[3] + n[4] >= 10, n
