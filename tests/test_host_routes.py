"""CPU suite: the routes of Stitcher::do_alignment (include/centrolign/stitcher.hpp:268-360) that are host algorithms —
pure deletion, greedy_partial_alignment, deletion_wfa_po_poa, pwfa_po_poa — through the product's host-only entry
cl_host_route_align, against the compiled reference's results (tests/golden/host_routes.npz, made by make_golden.py from
Stitcher::subalign) and live where oracle/_ref is present."""
import os

import numpy as np
import pytest

from centrolign_amd import capi
from oracle import pyoracle as po
from tests import helpers as H

ROUTE_NAMES = {1: "pd1", 2: "pd2", 3: "ad1", 4: "ad2", 5: "w", 6: "u"}


def _golden():
    z = np.load(os.path.join(H.GOLDEN, "host_routes.npz"))
    return z, H.host_route_batches.build(z["seq1"], z["seq2"])


def _check(batch, params, aln_off, pairs, want_routes):
    seen = {}
    for k in range(batch.n_problems):
        try:
            route, got = capi.host_route_align(batch, k, params)
        except capi.ClError as e:
            assert e.code == -6          # PO-POA: the device's route
            continue
        seen[route] = seen.get(route, 0) + 1
        assert np.array_equal(got, pairs[int(aln_off[k]):int(aln_off[k + 1])]), (k, ROUTE_NAMES[route])
    for r in want_routes:
        assert seen.get(r, 0) > 0, "route %s not exercised" % ROUTE_NAMES[r]
    return seen


@pytest.mark.parametrize("tag,routes", [("ad_linear", (3, 4)), ("w_linear", (5,)), ("w_dags", (5,)), ("mixed_dags", (1, 2, 3, 4, 6))])
def test_host_routes_match_reference_golden(tag, routes):
    z, batches = _golden()
    batch, params = batches[tag]
    seen = _check(batch, params, z[tag + ".aln_off"], z[tag + ".pairs"], routes)
    assert sum(seen.values()) >= 4


@pytest.mark.ref
@pytest.mark.skipif(not po.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_host_routes_vs_compiled_reference_live():
    from centrolign_amd import synth
    sp = capi.default_stitch_params()
    sp.min_wfa_size, sp.max_wfa_size, sp.max_wfa_ratio = 80, 10 ** 9, 3.0
    sp.max_trivial_size, sp.deletion_alignment_ratio = 40, 4
    sp.deletion_alignment_short_max_size, sp.deletion_alignment_long_min_size = 60, 50
    for seed, only_del in ((31, 0), (32, 1), (33, 0)):
        b = synth.random_dag_batch(120, seed=seed, max_n=130)
        b.only_deletion_alns[:] = only_del
        ref, _ = po.ref_stitch_batch(b, sp)
        _check(b, sp, ref.aln_off, ref.pairs, (5,) if not only_del else (3, 6))
