"""The data formats either side of a merge: cl_leaf_graph (make_base_graph + add_sentinels), cl_explicit_cigar, cl_write_gfa against
the compiled reference — golden digests (tests/golden/io.npz) and live where oracle/_ref is present.  Byte-identical text."""
import hashlib
import os

import numpy as np
import pytest

from centrolign_amd import capi, synth
from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

Z = np.load(os.path.join(H.GOLDEN, "io.npz"))
SEQS = ("A", "ACGTNacgtnXYZ-", "GATTACA" * 40)


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def io_texts():
    """(name, text) of every fixture case, produced by the functions under test"""
    out = []
    for m in range(3):
        z, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
        pairs = z["stitched"].reshape(-1, 2)
        fused = capi.fuse(graphs[0], graphs[1], pairs)
        names = ["seq%d" % i for i in range(len(fused.path_off) - 1)]
        out.append(("gfa.merge%d" % m, capi.write_gfa(fused, names)))
        out.append(("gfa_raw.merge%d" % m, capi.write_gfa(fused, names, decode=False)))
        out.append(("cigar.merge%d" % m, capi.explicit_cigar(graphs[0], graphs[1], pairs)))
    for name, g1, g2, pairs in H.fuse_cases()[:10]:
        fused = capi.fuse(g1, g2, pairs)
        out.append(("gfa." + name, capi.write_gfa(fused, ["p%d" % i for i in range(len(fused.path_off) - 1)])))
        out.append(("cigar." + name, capi.explicit_cigar(g1, g2, pairs)))
    return out


def test_texts_match_reference_golden():
    for name, text in io_texts():
        assert _sha(text) == str(Z[name][0]), name
    small = capi.write_gfa(capi.fuse(*H.fuse_cases()[2][1:]), ["a", "b", "c", "d", "e", "f"][:len(capi.fuse(*H.fuse_cases()[2][1:]).path_off) - 1])
    assert small == bytes(Z["gfa_text.fuse02"])           # one fixture in full
    assert small.startswith(b"H\tVN:Z:1.0\nS\t1\t") and small.endswith(b"\t*\n")


@pytest.mark.parametrize("seq", SEQS)
def test_leaf_graph_structure(seq):
    g = capi.leaf_graph(seq)
    n = len(seq)
    assert (g.src_id, g.snk_id) == (n, n + 1) and g.label[n] == 5 and g.label[n + 1] == 6
    code = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
    assert g.label[:n].tolist() == [code.get(c.upper(), 5) for c in seq]
    assert capi.graphs_equal(g, synth.base_graph_from_sequence(g.label[:n]))   # the builder the other tests use
    for k in capi.GRAPH_KEYS:
        assert np.array_equal(getattr(g, k), Z["leaf%d.%s" % (SEQS.index(seq), k)]), k


def test_errors():
    with pytest.raises(capi.ClError):
        capi.leaf_graph("")
    g1, g2, pairs = H.fuse_cases()[0][1:]
    bad = pairs.copy()
    bad[0, 1] = len(g2.label)
    with pytest.raises(capi.ClError):
        capi.explicit_cigar(g1, g2, bad)
    assert capi.explicit_cigar(g1, g2, np.zeros((0, 2), np.uint64)) == b""


@pytest.mark.skipif(not po.have_ref(), reason="needs oracle/_ref (build container only)")
def test_live_reference_agreement():
    for seq in SEQS + ("".join("ACGTN"[i] for i in np.random.default_rng(1).integers(0, 5, 3000)),):
        assert capi.graphs_equal(capi.leaf_graph(seq), po.ref_leaf_graph(seq))
    for name, g1, g2, pairs in H.fuse_cases():
        fused = capi.fuse(g1, g2, pairs)
        names = ["path_%d" % i for i in range(len(fused.path_off) - 1)]
        for dec in (True, False):
            assert capi.write_gfa(fused, names, dec) == po.ref_write_gfa(fused, names, dec), name
        assert capi.write_gfa(g2, names[:len(g2.path_off) - 1]) == po.ref_write_gfa(g2, names[:len(g2.path_off) - 1])
        assert capi.explicit_cigar(g1, g2, pairs) == po.ref_explicit_cigar(g1, g2, pairs)


def test_induced_pairwise_cigar_matches_the_reference():
    """the -A output of the CLI (src/core.cpp:546-550): explicit_cigar of the pairwise alignment an acyclic MSA graph induces on two of its
    paths — on the subproblem graphs of the restart fixture (committed expectations from the compiled reference) and, where the reference
    is built, live on random bubble graphs"""
    import os
    import numpy as np
    from centrolign_amd import synth
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "induced_cigars.npz"))
    for key in z["keys"]:
        f, a, b = str(key).split("|")
        g, names = capi.read_gfa(bytes(z["gfa." + f]))
        assert capi.induced_pairwise_cigar(g, int(a), int(b)) == bytes(z["cigar." + str(key)]), key
    g, names = capi.read_gfa(bytes(z["gfa." + str(z["keys"][0]).split("|")[0]]))
    assert capi.induced_pairwise_cigar(g, 0, 0) == b"%d=" % int(g.path_off[1] - g.path_off[0])
    with pytest.raises(capi.ClError):
        capi.induced_pairwise_cigar(g, 0, len(names))
    try:
        from oracle import pyoracle as po
        po.ref_lib()
    except Exception:
        return
    rng = np.random.default_rng(15)
    for k in range(4):
        g = synth.bubble_graph("".join("ACGT"[b] for b in rng.integers(0, 4, 250)), 3, seed=30 + k, alt_p=0.12, skip_p=0.06)
        for a in range(3):
            for b in range(3):
                assert capi.induced_pairwise_cigar(g, a, b) == po.ref_induced_pairwise_cigar(g, a, b), (k, a, b)
