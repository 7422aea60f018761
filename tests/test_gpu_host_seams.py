"""The host-side seams of the product (functions that are host algorithms in the reference too: exhaustive_chain_dp, internal_fuse, the guide-tree
plan, despecify_indel_breakpoints, the text formats, Bonder / simplify_bubbles / InconsistencyIdentifier, the match finder's host half) checked
against their reference goldens ON THE GPU BOX as well: the same assertions as the CPU suite (golden-based ones only — oracle/_ref does not
travel), so that the driver's `-m gpu` record covers them with the library build that ran there."""
import pytest

from tests import test_cyclize, test_cyclize_flow, test_despecify, test_exhaustive, test_extraction, test_io, test_plan

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [0, 1, 2])
def test_exhaustive_chain_equals_reference(m):
    test_exhaustive.test_exhaustive_chain_equals_reference(m)


def test_internal_fuse_and_masks():
    test_cyclize.test_internal_fuse_matches_the_reference()
    test_cyclize.test_diagonal_mask_and_update_mask_match_the_reference()


def test_bonds_bubbles_and_inconsistencies():
    test_cyclize_flow.test_identify_bonds_matches_the_reference()
    test_cyclize_flow.test_apply_bonds_and_simplify_bubbles_match_the_reference()
    test_cyclize_flow.test_inconsistencies_match_the_reference()


def test_plan_and_formats():
    assert len(test_plan.CASES) > 0
    for case in test_plan.CASES:
        test_plan.test_plan_equals_reference(case)
    test_plan.test_parse_fasta()
    test_io.test_texts_match_reference_golden()
    test_io.test_induced_pairwise_cigar_matches_the_reference()
    test_despecify.test_golden()
    test_extraction.test_concatenated_batches_hold_the_same_problems()


@pytest.mark.parametrize("tag,routes", [("ad_linear", (3, 4)), ("w_linear", (5,)), ("w_dags", (5,)), ("mixed_dags", (1, 2, 3, 4, 6))])
def test_host_routes_of_do_alignment(tag, routes):
    """pure deletion, greedy_partial_alignment, deletion_wfa_po_poa and pwfa_po_poa (include/centrolign/alignment.hpp:1178-2338) against the compiled
    reference's committed results (tests/golden/host_routes.npz, made by tests/golden/make_golden.py from Stitcher::subalign): the independent check of the
    two WFA variants on the GPU box — they are host algorithms, the goldens are the reference's own output (missing #4)"""
    from tests import test_host_routes
    test_host_routes.test_host_routes_match_reference_golden(tag, routes)
