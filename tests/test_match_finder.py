"""Match finding: PathMatchFinder::find_matches (include/centrolign/match_finder.hpp:120-212).

CPU suite: the oracle (oracle/match_oracle.cpp) and the host half of the product (cl_matches_from_suffix_array, fed the
oracle's suffix array) against the golden outputs of the compiled reference (tests/golden/match_finder.npz,
align4_30k_merge2.npz "ms.*") and, where oracle/_ref is present, against the reference live.
GPU suite: the device suffix array / LCP against the oracle's, cl_find_matches end to end against the goldens, and
size-independent properties of the suffix array at the BASELINE size (2 x 1 Mbp)."""
import os

import numpy as np
import pytest

from centrolign_amd import capi, synth
from oracle import pyoracle as po
from tests import helpers as H
from tests.test_extraction import load_stitch_case

Z = np.load(os.path.join(H.GOLDEN, "match_finder.npz"))
CASES = {name: (g1, g2, mc) for name, g1, g2, mc in H.match_cases()}


def _same(a, b):
    return all(np.array_equal(np.asarray(getattr(a, k)), np.asarray(getattr(b, k))) for k in capi.MatchSets._DT)


def _golden(name):
    return capi.MatchSets(**{k: Z["%s.%s" % (name, k)] for k in capi.MatchSets._DT})


def _host_half(g1, g2, mc):
    text = capi.match_joined_text(g1, g2)
    sa, lcp = po.oracle_suffix_array_lcp(text)
    return capi.matches_from_suffix_array(g1, g2, sa, lcp, max_count=mc)


def _check_merge(m, got):
    pre = "merge%d." % m
    for k in ("count1", "count2", "full_length"):
        assert np.array_equal(getattr(got, k), Z[pre + k]), k
    assert H.match_sets_digest(got) == str(Z[pre + "digest"][0])


def test_fixture_is_not_trivial():
    sizes = [len(Z[n + ".count1"]) for n in Z["names"]]
    assert sum(s > 100 for s in sizes) >= 15 and min(sizes) == 0
    multi = [n for n in Z["names"] if len(Z[n + ".count1"]) and (Z[n + ".count1"].max() > 1 or Z[n + ".count2"].max() > 1)]
    assert len(multi) >= 10                                  # repeats: sets with several walks per graph
    n_multi_path = sum(len(g1.path_off) > 2 or len(g2.path_off) > 2 for g1, g2, _ in CASES.values())
    assert n_multi_path >= 8                                  # merged-subproblem-like inputs with several paths


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    g1, g2, mc = CASES[name]
    assert _same(po.oracle_find_matches(g1, g2, max_count=mc), _golden(name))


@pytest.mark.parametrize("name", list(CASES))
def test_host_half_matches_reference_golden(name):
    g1, g2, mc = CASES[name]
    assert _same(_host_half(g1, g2, mc), _golden(name))


@pytest.mark.parametrize("m", [0, 1, 2])
def test_msa_merges_match_reference_golden(m):
    """the three merges of the 4 x 30 kbp MSA: leaf x leaf twice, then the root merge of two 2-path graphs"""
    _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
    for got in (po.oracle_find_matches(graphs[0], graphs[1]), _host_half(graphs[0], graphs[1], 3000)):
        _check_merge(m, got)
    if m == 2:   # what the reference's own run handed to Core::align for this merge
        z = np.load(os.path.join(H.GOLDEN, "align4_30k_merge2.npz"))
        assert _same(got, capi.MatchSets(**{k: z["ms." + k] for k in capi.MatchSets._DT}))


def test_interval_tree_ranges_agree(monkeypatch):
    """the host half walks the LCP-interval tree in ranges of suffix-array positions side by side (cl_match_api.cpp); CL_MATCH_TREE_GRAIN
    sets their width: one position per range, a few, and one range for everything give the reference's sets — multi-path graphs with
    repeats, where the open intervals in front of a range, the previous occurrences and the duplicate counts all cross range borders"""
    names = [n for n in CASES if len(CASES[n][0].path_off) > 2 or len(CASES[n][1].path_off) > 2][:6] + list(CASES)[:4]
    for grain in ("1", "2", "5", "37", "1000000000"):
        monkeypatch.setenv("CL_MATCH_TREE_GRAIN", grain)
        for name in names:
            g1, g2, mc = CASES[name]
            assert _same(_host_half(g1, g2, mc), _golden(name)), (grain, name)
        _, graphs, _ = load_stitch_case("stitch4_30k_merge2.npz")
        _check_merge(2, _host_half(graphs[0], graphs[1], 3000))


def test_oracle_suffix_array_is_sorted_and_lcp_exact():
    rng = np.random.default_rng(3)
    for n, alphabet in ((1, 1), (2, 1), (50, 1), (300, 2), (2000, 4)):
        text = np.concatenate([rng.integers(1, alphabet + 1, n - 1), [0]]).astype(np.uint8)
        sa, lcp = po.oracle_suffix_array_lcp(text)
        assert sorted(sa.tolist()) == list(range(n))
        b = text.tobytes()
        for p in range(1, n):
            x, y = b[sa[p - 1]:], b[sa[p]:]
            assert x < y
            k = 0
            while k < min(len(x), len(y)) and x[k] == y[k]:
                k += 1
            assert lcp[p] == k


def test_params_and_errors():
    mp = capi.MatchParams()
    capi.load_library().cl_match_params_default(mp)
    assert mp.max_count == 3000 and mp.use_color_set_size == 1   # src/parameters.cpp:36-37
    g1, g2, _ = CASES["rand00"]
    text = capi.match_joined_text(g1, g2)
    sa, lcp = po.oracle_suffix_array_lcp(text)
    with pytest.raises(capi.ClError):
        capi.matches_from_suffix_array(g1, g2, sa[:-1], lcp[:-1])    # not the joined text's length
    bad = synth.base_graph_from_sequence(np.array([0, 1, 2], np.uint8))
    bad.path_nodes = np.array([0, 1, 9], np.uint32)                  # path node out of range
    with pytest.raises(capi.ClError):
        capi.match_joined_text(bad, g2)


@pytest.mark.skipif(not po.have_ref(), reason="needs oracle/_ref (build container only)")
def test_live_reference_agreement():
    """fresh seeds against the compiled reference itself: oracle and host half, every max_count"""
    for seed in range(12):
        rng = np.random.default_rng(5000 + seed)
        L = int(rng.integers(30, 1500))
        unit = rng.integers(0, 4, int(rng.integers(2, 40))).astype(np.uint8)
        anc = np.tile(unit, L // len(unit) + 1)[:L].copy()
        mut = rng.random(L) < 0.03
        anc[mut] = rng.integers(0, 4, int(mut.sum()))
        g1 = synth.bubble_graph(anc, int(rng.integers(1, 6)), seed=seed)
        g2 = synth.bubble_graph(anc, int(rng.integers(1, 6)), seed=seed + 77, sentinels=(7, 8))
        for mc in (1, 9, 3000):
            want = po.ref_find_matches(g1, g2, max_count=mc)
            assert _same(po.oracle_find_matches(g1, g2, max_count=mc), want), (seed, mc)
            assert _same(_host_half(g1, g2, mc), want), (seed, mc)
            if mc == 9:   # the reference's other counting structure (esa.hpp:233-277) gives the same sets
                assert _same(po.ref_find_matches(g1, g2, max_count=mc, use_color_set_size=False), want)


# ---------------------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_gpu_suffix_array_and_lcp_match_oracle(gpu_ctx):
    rng = np.random.default_rng(11)
    texts = [np.array([0], np.uint8), np.array([3, 0], np.uint8)]
    for n, alphabet in ((9, 1), (500, 1), (4097, 2), (30000, 4), (200000, 4)):
        texts.append(np.concatenate([rng.integers(1, alphabet + 1, n - 1), [0]]).astype(np.uint8))
    unit = rng.integers(1, 5, 171)
    rep = np.tile(unit, 600)
    mut = rng.random(len(rep)) < 0.01
    rep[mut] = rng.integers(1, 5, int(mut.sum()))
    texts.append(np.concatenate([rep, [0]]).astype(np.uint8))            # tandem repeat: long LCPs, many doubling rounds
    texts.append(np.concatenate([np.tile(unit, 300), [7], np.tile(unit, 300), [0]]).astype(np.uint8))   # exact copies
    for text in texts:
        sa, lcp, isa, rounds = gpu_ctx.suffix_array_lcp(text)
        want_sa, want_lcp = po.oracle_suffix_array_lcp(text)
        assert np.array_equal(sa, want_sa), len(text)
        assert np.array_equal(lcp, want_lcp), len(text)
        assert np.array_equal(isa[sa], np.arange(len(text)))
        assert rounds >= 1


@pytest.mark.gpu
def test_gpu_suffix_array_at_the_radix_sort_boundaries(gpu_ctx):
    """round 6: the suffix array's sorts and scans are the library's own (csrc/cl_radix.h: tiles of 2 048 keys, waves of 512, rounds of 64 lanes).  Text lengths on both sides of
    every boundary, alphabets that put keys in every byte value of a digit (labels up to 127) and in a single one (all keys equal in the first rounds), against the oracle"""
    rng = np.random.default_rng(29)
    texts = []
    for n in (2, 3, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6145, 10240, 65536, 65537):
        for alphabet in (1, 3, 127):
            texts.append(np.concatenate([rng.integers(1, alphabet + 1, n - 1), [0]]).astype(np.uint8))
    texts.append(np.concatenate([np.tile(np.array([5, 9, 5, 9, 5, 100], np.uint8), 3000), [0]]).astype(np.uint8))   # short period: every doubling round re-sorts large equal groups
    for text in texts:
        sa, lcp, isa, rounds = gpu_ctx.suffix_array_lcp(text)
        want_sa, want_lcp = po.oracle_suffix_array_lcp(text)
        assert np.array_equal(sa, want_sa), (len(text), int(text.max()))
        assert np.array_equal(lcp, want_lcp), (len(text), int(text.max()))
        assert np.array_equal(isa[sa], np.arange(len(text)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_find_matches_matches_reference_golden(gpu_ctx, name):
    g1, g2, mc = CASES[name]
    got, st = gpu_ctx.find_matches(g1, g2, max_count=mc, want_stats=True)
    assert _same(got, _golden(name))
    assert _same(got, po.oracle_find_matches(g1, g2, max_count=mc))
    assert st["doubling_rounds"] >= 1 and st["text_length"] == len(capi.match_joined_text(g1, g2))


@pytest.mark.gpu
@pytest.mark.parametrize("m", [0, 1, 2])
def test_gpu_find_matches_msa_merges(gpu_ctx, m):
    _, graphs, _ = load_stitch_case("stitch4_30k_merge%d.npz" % m)
    _check_merge(m, gpu_ctx.find_matches(graphs[0], graphs[1]))


@pytest.mark.gpu
def test_gpu_matches_feed_core_align(gpu_ctx):
    """find_matches -> core_align: the merge's alignment from nothing but the two graphs equals the reference's
    (align4_30k_merge2.npz), i.e. the path of Core's per-merge work (core.hpp:289-296) end to end"""
    z = np.load(os.path.join(H.GOLDEN, "align4_30k_merge2.npz"))
    _, graphs, _ = load_stitch_case("stitch4_30k_merge2.npz")
    ms = gpu_ctx.find_matches(graphs[0], graphs[1])
    got = gpu_ctx.core_align(graphs[0], graphs[1], ms, score_scale=float(z["score_scale"][0]), max_num_match_pairs=40000)
    assert np.array_equal(got["alignment"].reshape(-1), z["stitched"])


@pytest.mark.gpu
def test_gpu_full_size_suffix_array_properties(gpu_ctx):
    """BASELINE size (2 x 1 Mbp leaf pair): the suffix array is a permutation, consecutive suffixes are in order given the
    LCP (character after the common prefix strictly increases), the LCP values are exact on a sample, and the match sets
    are well-formed (every walk spells the same sequence, count products within max_count)"""
    seqs = synth.hor_sequences(7, 1000000, 2)
    g1 = synth.base_graph_from_sequence(seqs[0])
    g2 = synth.base_graph_from_sequence(seqs[1], sentinels=(7, 8))
    text = capi.match_joined_text(g1, g2)
    sa, lcp, isa, rounds = gpu_ctx.suffix_array_lcp(text)
    n = len(text)
    assert np.array_equal(np.sort(sa), np.arange(n, dtype=np.uint32))
    assert np.array_equal(isa[sa], np.arange(n))
    a, b, l = sa[:-1].astype(np.int64), sa[1:].astype(np.int64), lcp[1:].astype(np.int64)
    pad = np.concatenate([text, np.zeros(1, np.uint8)])
    assert np.all(pad[np.minimum(a + l, n)] < pad[np.minimum(b + l, n)])
    rng = np.random.default_rng(0)
    raw = text.tobytes()
    for p in rng.integers(1, n, 400):
        x, y, k = int(sa[p - 1]), int(sa[p]), int(lcp[p])
        assert raw[x:x + k] == raw[y:y + k]
    ms, st = gpu_ctx.find_matches(g1, g2, want_stats=True)
    assert ms.n_sets > 50000 and st["doubling_rounds"] == rounds
    # the compiled reference's output on this pair, by digest (tests/golden/make_golden.py step 10)
    assert ms.n_sets == int(Z["c2.n_sets"][0]) and H.match_sets_digest(ms) == str(Z["c2.digest"][0])
    assert np.all(ms.count1 * ms.count2 <= 3000) and np.all(ms.count1 * ms.count2 > 0)
    lab1, lab2 = g1.label, g2.label
    for s in rng.integers(0, ms.n_sets, 300):
        w1 = [ms.nodes1[ms.walk_off1[w]:ms.walk_off1[w + 1]] for w in range(int(ms.set_off1[s]), int(ms.set_off1[s + 1]))]
        w2 = [ms.nodes2[ms.walk_off2[w]:ms.walk_off2[w + 1]] for w in range(int(ms.set_off2[s]), int(ms.set_off2[s + 1]))]
        spell = lab1[w1[0]].tobytes()
        assert len(spell) == ms.full_length[s]
        assert all(lab1[w].tobytes() == spell for w in w1) and all(lab2[w].tobytes() == spell for w in w2)
