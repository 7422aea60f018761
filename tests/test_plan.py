"""The front of the driver behind the C ABI: cl_msa_plan_create = Tree(newick) + Execution's normalisation and order
(src/tree.cpp:39-160, src/execution.cpp:12-92), cl_parse_fasta = parse_fasta (src/utility.cpp:19-65).  Goldens: the compiled
reference's own leaf and merge order on odd guide trees — polytomies, unary chains, quoted labels, leaves without a sequence, FASTA order
different from the tree's, malformed input (tests/golden/msa_plans.json, made by tests/golden/make_golden.py plans)."""
import json
import os

import pytest

from centrolign_amd import capi
from oracle import pyoracle as po
from tests import helpers as H

CASES = json.load(open(os.path.join(H.GOLDEN, "msa_plans.json")))


def plan_text(newick, names):
    try:
        leaves, merges = capi.msa_plan(newick, names)
    except capi.ClError as e:
        return "E " + str(e).split(": ", 1)[-1] + "\n"
    under = [[names[s]] for s in leaves]
    out = ["L %s" % names[s] for s in leaves]
    for a, b in merges:
        out.append("M %s;%s" % (",".join(sorted(under[a])), ",".join(sorted(under[b]))))
        under.append(under[a] + under[b])
    return "\n".join(out) + "\n"


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["newick"][:30] or "in-order")
def test_plan_equals_reference(case):
    got = plan_text(case["newick"], case["names"])
    if case["plan"].startswith("E "):
        assert got.startswith("E ") and case["plan"][2:].strip() in got
    else:
        assert got == case["plan"]


def test_parse_fasta():
    assert capi.parse_fasta(">a desc\nACGT\nAC\n>b\nGG\n") == [("a", "ACGTAC"), ("b", "GG")]
    assert capi.parse_fasta(">x\nAC\nGT") == [("x", "ACGT")]
    for bad in ("ACGT\n", ">\nAC\n", "", ">a\nAC\nACG\n", ">a\nACG\nAC\nAC\n"):
        with pytest.raises(capi.ClError):
            capi.parse_fasta(bad)


@pytest.mark.ref
@pytest.mark.skipif(not po.have_ref(), reason="needs the compiled reference (build container only)")
def test_plan_live_random_trees():
    """random guide trees (polytomies, unary nodes, missing leaves) against ref_msa_plan"""
    import ctypes as C
    import random
    lib = po.ref_lib()
    lib.ref_msa_plan.restype = C.c_int
    lib.ref_msa_plan.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint64, C.POINTER(C.c_char_p)]
    rng = random.Random(5)

    def tree(names):
        if len(names) == 1:
            t = names[0]
            while rng.random() < 0.2:
                t = "(" + t + ")"
            return t
        k = min(len(names), rng.choice([2, 2, 2, 3, 4]))
        cuts = sorted(rng.sample(range(1, len(names)), k - 1))
        parts = [names[i:j] for i, j in zip([0] + cuts, cuts + [len(names)])]
        return "(" + ",".join(tree(p) for p in parts) + ")" + (":%g" % rng.random() if rng.random() < 0.3 else "")
    for trial in range(200):
        n = rng.randint(2, 12)
        names = ["n%d" % i for i in range(n)]
        rng.shuffle(names)
        nwk = tree(names) + ";"
        have = [x for x in names if rng.random() < 0.8] or names[:1]
        rng.shuffle(have)
        arr = (C.c_char_p * len(have))(*[x.encode() for x in have])
        txt = C.c_char_p()
        lib.ref_msa_plan(nwk.encode(), arr, len(have), C.byref(txt))
        assert plan_text(nwk, have) == txt.value.decode(), (nwk, have)
