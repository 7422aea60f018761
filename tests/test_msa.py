"""End to end without the reference in the loop: sequences + guide tree -> leaf graphs, calibration, one cl_merge per tree node, output
text — byte-identical to what the compiled reference's whole pipeline (Core::execute + write_gfa / explicit_cigar) printed for the same
FASTA + Newick (tests/golden/msa_text.npz, made by ref_msa_dump)."""
import os

import numpy as np
import pytest

from centrolign_amd import msa, synth
from tests import helpers as H

Z = np.load(os.path.join(H.GOLDEN, "msa_text.npz"))


def test_tree_helpers():
    t = msa.balanced_tree(["a", "b", "c", "d", "e"])
    assert msa.newick(t) == "((a,b),(c,(d,e)))" and msa.leaves_of(t) == ["a", "b", "c", "d", "e"]


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 3])
@pytest.mark.parametrize("case", H.msa_cases(), ids=lambda c: c[0])
def test_gpu_native_run_prints_the_references_text(gpu_ctx, case, workers):
    """workers = 3: leaf calibrations and sibling merges side by side on one device, one cl_context per worker thread"""
    name, n, length, seed, budget = case
    seqs = synth.hor_sequences(seed, length, n, seq_div=0.01, hor_div=0.03, indel_hor=2)
    names = ["seq%d" % i for i in range(n)]
    r = msa.progressive_msa(gpu_ctx, dict(zip(names, seqs)), msa.balanced_tree(names), max_num_match_pairs=budget, workers=workers)
    want = bytes(Z[name])
    got = msa.output_text(r)
    assert got == (want.rstrip(b"\n") if n == 2 else want)   # the CIGAR line ends in a newline in the dump
    assert r["stats"]["merges"] == n - 1 and len(r["root"].path_off) - 1 == n
