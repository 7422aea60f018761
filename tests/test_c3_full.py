"""BASELINE configs[2] at full size on the device: 10 x 1 Mbp (seed 7), guide tree of SURVEY.md §8(d), the whole progressive MSA through
the C ABI.  The single-threaded reference finishes eight of the nine merges of this input in the build container before it runs out of
memory at the root (tests/golden/make_c3_digests.py, 37 minutes); the GFA text of each of those eight subproblems, as the reference's -S
option writes it (Core::emit_subproblem, src/core.cpp:392-422), is pinned here by sha256 and must be reproduced byte for byte.  The
root merge at the default budget has no reference output: there it is pinned by what can be checked without one — every path of the root
graph spells its input sequence, the graph is the fuse of the two pinned children along an alignment that covers both of them completely,
and the text is the same from one worker context and from four.  The reference DOES finish the root merge when it is restarted (-R) from
its own two subproblem files with a budget of 500 000 match pairs (25 minutes, 28 GB): that GFA is pinned by digest and reproduced here the
same way — the two pinned children written as GFA, read back as a restart reads them, merged with that budget (5 + 5 paths, 36 chain
combinations, full-size graphs: the walk kernel's exchange between workgroups and the far pass at scale, against the reference)."""
import hashlib
import json
import os

import numpy as np
import pytest

from centrolign_amd import capi, msa, synth
from tests import helpers as H

GOLD = json.load(open(os.path.join(H.GOLDEN, "c3_10x1M_subproblems.json")))
_DEC = np.zeros(256, np.uint8)
for _c, _v in zip(b"ACGTN", range(5)):
    _DEC[_v] = _c


def spelled(graph, path):
    lo, hi = int(graph.path_off[path]), int(graph.path_off[path + 1])
    return _DEC[graph.label[graph.path_nodes[lo:hi]]].tobytes().decode()


@pytest.fixture(scope="module")
def c3_run(gpu_ctx):
    names, seqs, tree = synth.c3_workload()
    assert hashlib.sha256("".join(seqs[n] for n in names).encode()).hexdigest() == GOLD["input_sha256"]
    r = msa.progressive_msa(gpu_ctx, seqs, tree, workers=4, keep_merges=True)
    return names, seqs, tree, r


@pytest.mark.gpu
def test_every_subproblem_the_reference_finished_is_reproduced(c3_run):
    names, seqs, tree, r = c3_run
    seen = set()
    for m in r["stats"]["kept"]:
        key = ",".join(sorted(m["paths"]))
        if key not in GOLD["subproblems"]:
            continue
        gfa = capi.write_gfa(m["fused"], m["paths"])
        assert len(gfa) == GOLD["subproblems"][key]["bytes"], key
        assert hashlib.sha256(gfa).hexdigest() == GOLD["subproblems"][key]["sha256"], key
        seen.add(key)
    assert seen == set(GOLD["subproblems"]) and len(seen) == 8


@pytest.mark.gpu
def test_root_merge_properties(c3_run):
    names, seqs, tree, r = c3_run
    root, paths = r["root"], r["paths"]
    assert sorted(paths) == sorted(names)
    for i, nm in enumerate(paths):       # every embedded path spells its input sequence
        assert spelled(root, i) == seqs[nm], nm
    last = r["stats"]["kept"][-1]
    assert sorted(last["paths"]) == sorted(names)
    g1, g2 = last["graphs"]
    aln = last["align"]["alignment"]
    gap = np.uint64(0xFFFFFFFFFFFFFFFF)
    # the alignment walks both children from source to sink: every non-sentinel node of each appears on one of its sides,
    # in an order the graphs' own topological orders agree with (a path through each child)
    a, b = aln[:, 0], aln[:, 1]
    assert len(set(a[a != gap].tolist())) == int((a != gap).sum()) and len(set(b[b != gap].tolist())) == int((b != gap).sum())
    # THE REFERENCE'S OWN OUTPUT for the whole configuration at the default budget (round 4: the unmodified reference finished the root merge on a GPU
    # box's host — 36 minutes, 70 GB — restarted from its eight subproblem files; tests/golden/make_c3_root_default.py): byte for byte
    full = capi.write_gfa(root, paths)
    assert len(full) == GOLD["root_default_budget_reference"]["bytes"]
    assert hashlib.sha256(full).hexdigest() == GOLD["root_default_budget_reference"]["sha256"]
    assert GOLD["root_default_budget_reference"]["sha256"] == GOLD["root_default_budget_self_digest"]["sha256"]   # (what rounds 2 and 3 printed)
    # and fusing the two pinned children along it gives the root graph (cl_fuse is pinned by tests/test_fuse.py)
    from bench import relabelled
    assert capi.graphs_equal(capi.fuse(relabelled(g1, 5, 6), relabelled(g2, 7, 8), aln), last["fused"])


@pytest.mark.gpu
def test_one_worker_prints_the_same_gfa(gpu_ctx, c3_run):
    names, seqs, tree, r = c3_run
    r1 = msa.progressive_msa(gpu_ctx, seqs, tree, workers=1)
    assert capi.write_gfa(r1["root"], r1["paths"]) == capi.write_gfa(r["root"], r["paths"])


@pytest.mark.gpu
def test_root_merge_restarted_with_a_smaller_budget_matches_the_reference(gpu_ctx, c3_run):
    names, seqs, tree, r = c3_run
    gold = GOLD["root_restart_500k"]
    kept = {",".join(sorted(m["paths"])): m for m in r["stats"]["kept"]}
    children = []
    for key in ("s0,s1,s2,s3,s4", "s5,s6,s7,s8,s9"):
        m = kept[key]
        text = capi.write_gfa(m["fused"], m["paths"])
        assert hashlib.sha256(text).hexdigest() == GOLD["subproblems"][key]["sha256"]      # the very file the reference restarted from
        children.append(capi.read_gfa(text))                                                 # read_gfa + add_sentinels, as Core::restart does
    (g1, p1), (g2, p2) = children
    got = gpu_ctx.merge(g1, g2, score_scale=r["scale"], max_num_match_pairs=gold["max_num_match_pairs"])
    gfa = capi.write_gfa(got["fused"], p1 + p2)
    assert len(gfa) == gold["bytes"]
    assert hashlib.sha256(gfa).hexdigest() == gold["sha256"]
