"""ctypes mirror of include/centrolign_amd.h (the C ABI of the MI355X stitch path).

Plumbing only: numpy arrays in, numpy arrays out.  There is NO CPU fallback here: if the HIP shared
library is missing or no gfx950 device is visible, loading / context creation raises.

The flat batch layout (`StitchBatch`) is the C struct `cl_stitch_batch`: per graph side the concatenated
SubGraphInfo fields of every between-anchor subproblem (reference:
include/centrolign/subgraph_extraction.hpp:14-33).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CL_LIBRARY: another build of the SAME library (the sanitizer build, `make SAN=1` -> lib/san/: scripts/san_suite.sh); never a fallback
LIB_PATH = os.environ.get("CL_LIBRARY") or os.path.join(_HERE, "lib", "libcentrolign_amd.so")

CL_GAP = np.uint64(0xFFFFFFFFFFFFFFFF)

ROUTE_NAMES = {0: "po", 1: "pd1", 2: "pd2", 3: "ad1", 4: "ad2", 5: "w", 6: "u"}

ERRORS = {
    0: "CL_OK", -1: "CL_ERR_INVALID_ARGUMENT", -2: "CL_ERR_BAD_GAP_PARAMS", -3: "CL_ERR_NO_DEVICE",
    -4: "CL_ERR_HIP", -5: "CL_ERR_OUT_OF_MEMORY", -6: "CL_ERR_UNSUPPORTED_ROUTE", -7: "CL_ERR_CYCLIC_GRAPH",
    -8: "CL_ERR_UNREACHABLE_SINK",
}


class ClError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d)%s" % (ERRORS.get(code, "CL_ERR_?"), code, (": " + msg) if msg else ""))


class AlignParams(C.Structure):
    """cl_align_params == AlignmentParameters<3> (include/centrolign/alignment.hpp:56-65)"""
    _fields_ = [("match", C.c_uint32), ("mismatch", C.c_uint32),
                ("gap_open", C.c_uint32 * 3), ("gap_extend", C.c_uint32 * 3)]


class StitchParams(C.Structure):
    """cl_stitch_params == Stitcher public tunables (include/centrolign/stitcher.hpp:48-64)"""
    _fields_ = [("alignment_params", AlignParams),
                ("max_trivial_size", C.c_uint64), ("min_wfa_size", C.c_uint64), ("max_wfa_size", C.c_uint64),
                ("max_wfa_ratio", C.c_double), ("wfa_pruning_dist", C.c_uint64),
                ("deletion_alignment_ratio", C.c_uint64),
                ("deletion_alignment_short_max_size", C.c_uint64),
                ("deletion_alignment_long_min_size", C.c_uint64)]


def default_stitch_params():
    """The values the centrolign CLI runs with (src/parameters.cpp:74-85)."""
    p = StitchParams()
    p.alignment_params.match = 20
    p.alignment_params.mismatch = 80
    p.alignment_params.gap_open[:] = [60, 800, 2500]
    p.alignment_params.gap_extend[:] = [30, 5, 1]
    p.max_trivial_size = 30000
    p.min_wfa_size = 40000000
    p.max_wfa_size = 75000000
    p.max_wfa_ratio = 1.05
    p.wfa_pruning_dist = 25
    p.deletion_alignment_ratio = 8
    p.deletion_alignment_short_max_size = 1500
    p.deletion_alignment_long_min_size = 2000
    return p


def make_align_params(match, mismatch, gap_open, gap_extend):
    p = AlignParams()
    p.match, p.mismatch = match, mismatch
    go = list(gap_open) + [gap_open[-1]] * (3 - len(gap_open))
    ge = list(gap_extend) + [gap_extend[-1]] * (3 - len(gap_extend))
    p.gap_open[:] = go
    p.gap_extend[:] = ge
    return p


class GraphSideC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("node_off", "label", "prev_off", "prev_idx", "next_off", "next_idx",
                 "src_off", "src_idx", "snk_off", "snk_idx", "back_translation")]


class StitchBatchC(C.Structure):
    _fields_ = [("n_problems", C.c_uint64), ("side", GraphSideC * 2), ("only_deletion_alns", C.c_void_p)]


class StitchResultC(C.Structure):
    _fields_ = [("n_problems", C.c_uint64), ("aln_off", C.POINTER(C.c_uint64)), ("pairs", C.POINTER(C.c_uint64)),
                ("score", C.POINTER(C.c_int64)), ("route", C.POINTER(C.c_uint8)), ("num_pw", C.POINTER(C.c_uint8))]


class PlanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("n_problems", "n_po_poa", "dp_cells", "dp_bytes", "n_linear", "max_cells", "workspace_bytes",
                 "n_launches", "n_strip_fallbacks")]


class BaseGraphC(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("label", C.c_void_p), ("next_off", C.c_void_p), ("next_idx", C.c_void_p),
                ("prev_off", C.c_void_p), ("prev_idx", C.c_void_p), ("n_paths", C.c_uint64), ("path_off", C.c_void_p),
                ("path_nodes", C.c_void_p), ("src_id", C.c_uint64), ("snk_id", C.c_uint64)]


class AnchorSegmentsC(C.Structure):
    _fields_ = [("n_segments", C.c_uint64), ("seg_off", C.c_void_p), ("walk_off", C.c_void_p), ("walk1", C.c_void_p),
                ("walk2", C.c_void_p)]


class AlignmentC(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("pairs", C.POINTER(C.c_uint64))]


class BaseGraph:
    """numpy holder for a cl_base_graph (BaseGraph + SentinelTableau of one side of a merge)"""

    def __init__(self, label, next_off, next_idx, prev_off, prev_idx, path_off, path_nodes, src_id, snk_id):
        self.label = np.ascontiguousarray(label, np.uint8)
        self.next_off = np.ascontiguousarray(next_off, np.uint64)
        self.next_idx = np.ascontiguousarray(next_idx, np.uint32)
        self.prev_off = np.ascontiguousarray(prev_off, np.uint64)
        self.prev_idx = np.ascontiguousarray(prev_idx, np.uint32)
        self.path_off = np.ascontiguousarray(path_off, np.uint64)
        self.path_nodes = np.ascontiguousarray(path_nodes, np.uint32)
        self.src_id, self.snk_id = int(src_id), int(snk_id)

    def as_c(self):
        g = BaseGraphC()
        g.n_nodes = len(self.label)
        g.n_paths = len(self.path_off) - 1
        for k in ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes"):
            setattr(g, k, getattr(self, k).ctypes.data)
        g.src_id, g.snk_id = self.src_id, self.snk_id
        return g


class AnchorSegments:
    def __init__(self, seg_off, walk_off, walk1, walk2):
        self.seg_off = np.ascontiguousarray(seg_off, np.uint64)
        self.walk_off = np.ascontiguousarray(walk_off, np.uint64)
        self.walk1 = np.ascontiguousarray(walk1, np.uint32)
        self.walk2 = np.ascontiguousarray(walk2, np.uint32)

    def as_c(self):
        s = AnchorSegmentsC()
        s.n_segments = len(self.seg_off) - 1
        for k in ("seg_off", "walk_off", "walk1", "walk2"):
            setattr(s, k, getattr(self, k).ctypes.data)
        return s


def _side_from_c(sc, n):
    """copy a cl_graph_side (C pointers) into a GraphSide"""
    def arr(ptr, dt, cnt):
        if not ptr or cnt == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()
    node_off = arr(sc.node_off, np.uint64, n + 1)
    nn = int(node_off[-1])
    prev_off = arr(sc.prev_off, np.uint64, nn + 1)
    next_off = arr(sc.next_off, np.uint64, nn + 1)
    src_off = arr(sc.src_off, np.uint64, n + 1)
    snk_off = arr(sc.snk_off, np.uint64, n + 1)
    return GraphSide(node_off=node_off, label=arr(sc.label, np.uint8, nn), prev_off=prev_off,
                     prev_idx=arr(sc.prev_idx, np.uint32, int(prev_off[-1])), next_off=next_off,
                     next_idx=arr(sc.next_idx, np.uint32, int(next_off[-1])), src_off=src_off,
                     src_idx=arr(sc.src_idx, np.uint32, int(src_off[-1])), snk_off=snk_off,
                     snk_idx=arr(sc.snk_idx, np.uint32, int(snk_off[-1])),
                     back_translation=arr(sc.back_translation, np.uint64, nn))


def despecify_indel_breakpoints(score, gap_before, gap_score_before, gap_after, gap_score_after,
                                min_indel_fuzz_length=50, indel_fuzz_score_proportion=0.001):
    """Stitcher::despecify_indel_breakpoints on parallel arrays; returns (keep mask, kept gap_before, gap_score_before,
    gap_after, gap_score_after)"""
    lib = load_library()
    n = len(score)
    sc = np.ascontiguousarray(score, np.float64)
    gb = np.array(gap_before, np.int64); ga = np.array(gap_after, np.int64)
    gsb = np.array(gap_score_before, np.float64); gsa = np.array(gap_score_after, np.float64)
    keep = np.zeros(max(n, 1), np.uint8)
    kept = C.c_uint64(0)
    lib.cl_despecify_indel_breakpoints.restype = C.c_int
    rc = lib.cl_despecify_indel_breakpoints(C.c_uint64(n), sc.ctypes.data_as(C.c_void_p), gb.ctypes.data_as(C.c_void_p),
                                            gsb.ctypes.data_as(C.c_void_p), ga.ctypes.data_as(C.c_void_p),
                                            gsa.ctypes.data_as(C.c_void_p), C.c_int64(min_indel_fuzz_length),
                                            C.c_double(indel_fuzz_score_proportion), keep.ctypes.data_as(C.c_void_p), C.byref(kept))
    if rc != 0:
        raise ClError(rc)
    k = int(kept.value)
    return keep[:n].astype(bool), gb[:k], gsb[:k], ga[:k], gsa[:k]


def host_route_align(batch, k, params=None):
    """cl_host_route_align: one subproblem by the host route Stitcher::do_alignment takes for it (pure deletion, greedy,
    deletion-WFA, pruned WFA); no device needed.  Returns (route, (n, 2) uint64 pairs)"""
    lib = load_library()
    params = params or default_stitch_params()
    bc = batch.as_c()
    route, out, n = C.c_int(-100), C.c_void_p(), C.c_uint64(0)
    lib.cl_host_route_align.restype = C.c_int
    lib.cl_host_route_align.argtypes = [C.POINTER(StitchBatchC), C.c_uint64, C.POINTER(StitchParams), C.POINTER(C.c_int),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    rc = lib.cl_host_route_align(C.byref(bc), int(k), C.byref(params), C.byref(route), C.byref(out), C.byref(n))
    if rc != 0:
        raise ClError(rc, "route %d" % route.value)
    cnt = int(n.value)
    pairs = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(max(cnt, 1) * 2,))[:2 * cnt].copy().reshape(cnt, 2)
    C.CDLL(None).free(out)
    return int(route.value), pairs


def stitch_rank_order(batch, k, side, mode="auto"):
    """cl_stitch_rank_order: the topological order the device ranks subgraph `side` of subproblem k by ("auto": the packer's choice, "lifo": the reference's order,
    "level": by longest path from a source); no device needed.  Returns (order: uint32 local node ids by rank, reads more than four ranks back, longest read)"""
    lib = load_library()
    bc = batch.as_c()
    n = int(batch.side[side].node_off[k + 1] - batch.side[side].node_off[k])
    order = np.zeros(max(n, 1), np.uint32)
    far, longest = C.c_uint32(0), C.c_uint32(0)
    lib.cl_stitch_rank_order.restype = C.c_int
    lib.cl_stitch_rank_order.argtypes = [C.POINTER(StitchBatchC), C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rc = lib.cl_stitch_rank_order(C.byref(bc), int(k), int(side), {"auto": 0, "lifo": 1, "level": 2}[mode], order.ctypes.data, C.byref(far), C.byref(longest))
    if rc != 0:
        raise ClError(rc, "cl_stitch_rank_order")
    return order[:n], int(far.value), int(longest.value)


def partition_anchors(graph1, graph2, chain, score_scale=1.0, score_boundaries=False, use_annotated_score=False, **overrides):
    """Partitioner::partition_anchors (include/centrolign/partitioner.hpp:72-213), host only.  `chain` is the dict that
    Context.anchor_chain returns (walk_off, walk1, walk2, count1, count2, full_length, chain[:,0] = match set, score).
    Returns an (n_segments, 2) array of (first, past-the-last) anchor indices."""
    lib = load_library()
    pp = PartitionParams()
    lib.cl_partition_params_default(C.byref(pp))
    pp.score_scale = float(score_scale)
    pp.score_boundaries = int(score_boundaries)
    pp.use_annotated_score = int(use_annotated_score)
    for k, v in overrides.items():
        setattr(pp, k, v)
    keep = dict(walk_off=np.ascontiguousarray(chain["walk_off"], np.uint64), walk1=np.ascontiguousarray(chain["walk1"], np.uint32),
                walk2=np.ascontiguousarray(chain["walk2"], np.uint32), count1=np.ascontiguousarray(chain["count1"], np.uint64),
                count2=np.ascontiguousarray(chain["count2"], np.uint64), full_length=np.ascontiguousarray(chain["full_length"], np.uint64),
                match_set=np.ascontiguousarray(np.asarray(chain["chain"])[:, 0] if len(chain["chain"]) else np.zeros(0), np.uint64),
                score=np.ascontiguousarray(chain["score"], np.float64))
    n = len(keep["count1"])
    af = AnchorFieldsC(n, *[keep[k].ctypes.data for k in ("walk_off", "walk1", "walk2", "count1", "count2", "full_length", "match_set", "score")])
    g1, g2 = graph1.as_c(), graph2.as_c()
    out, ns = C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_partition_anchors(C.byref(g1), C.byref(g2), C.byref(af), C.byref(pp), C.byref(out), C.byref(ns))
    if rc != 0:
        raise ClError(rc)
    k = int(ns.value)
    seg = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(max(k, 1) * 2,))[:2 * k].copy().reshape(k, 2)
    C.CDLL(None).free(out)
    return seg


def split_branching_matches(graph1, graph2, matches, anchor_split_limit=5, min_split_length=128, min_path_length_spread=50,
                            max_split_match_set_size=16):
    """Anchorer::split_branching_matches (include/centrolign/anchorer.hpp:800-956); host only.  Returns the new MatchSets."""
    lib = load_library()
    g1, g2, mc = graph1.as_c(), graph2.as_c(), matches.as_c()
    sp = SplitParams(anchor_split_limit, min_split_length, min_path_length_spread, max_split_match_set_size)
    h = C.c_void_p()
    rc = lib.cl_split_branching_matches(C.byref(g1), C.byref(g2), C.byref(mc), C.byref(sp), C.byref(h))
    if rc != 0:
        raise ClError(rc)
    return _take_owned_match_sets(lib, h)


def _take_owned_match_sets(lib, h):
    """copies a cl_owned_match_sets into a MatchSets and frees it"""
    try:
        v = MatchSetsC()
        lib.cl_owned_match_sets_view(h, C.byref(v))
        n = int(v.n_sets)

        def arr(ptr, dt, k):
            if k == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(k,)).copy()
        so1, so2 = arr(v.set_off1, np.uint64, n + 1), arr(v.set_off2, np.uint64, n + 1)
        wo1, wo2 = arr(v.walk_off1, np.uint64, int(so1[-1]) + 1), arr(v.walk_off2, np.uint64, int(so2[-1]) + 1)
        return MatchSets(set_off1=so1, walk_off1=wo1, nodes1=arr(v.nodes1, np.uint32, int(wo1[-1])), set_off2=so2, walk_off2=wo2,
                         nodes2=arr(v.nodes2, np.uint32, int(wo2[-1])), count1=arr(v.count1, np.uint64, n),
                         count2=arr(v.count2, np.uint64, n), full_length=arr(v.full_length, np.uint64, n))
    finally:
        lib.cl_owned_match_sets_free(h)


GRAPH_KEYS = ("label", "next_off", "next_idx", "prev_off", "prev_idx", "path_off", "path_nodes")


def _take_owned_base_graph(lib, h):
    """copies a cl_owned_base_graph into a BaseGraph and frees it"""
    try:
        v = BaseGraphC()
        lib.cl_owned_base_graph_view(h, C.byref(v))
        n, p = int(v.n_nodes), int(v.n_paths)

        def arr(ptr, dt, k):
            if k == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(k,)).copy()
        no, po, pao = arr(v.next_off, np.uint64, n + 1), arr(v.prev_off, np.uint64, n + 1), arr(v.path_off, np.uint64, p + 1)
        return BaseGraph(arr(v.label, np.uint8, n), no, arr(v.next_idx, np.uint32, int(no[-1])), po, arr(v.prev_idx, np.uint32, int(po[-1])),
                         pao, arr(v.path_nodes, np.uint32, int(pao[-1])), int(v.src_id), int(v.snk_id))
    finally:
        lib.cl_owned_base_graph_free(h)


def fuse(dest, source, pairs):
    """fuse (include/centrolign/fuse.hpp:46-152): `source` merged into `dest` along the alignment ((n, 2) uint64 AlignedPairs,
    gap = 2^64 - 1); host only.  Returns the fused BaseGraph."""
    lib = load_library()
    g1, g2 = dest.as_c(), source.as_c()
    pairs = np.ascontiguousarray(pairs, np.uint64).reshape(-1, 2)
    h = C.c_void_p()
    rc = lib.cl_fuse(C.byref(g1), C.byref(g2), pairs.ctypes.data, len(pairs), C.byref(h))
    if rc != 0:
        raise ClError(rc)
    return _take_owned_base_graph(lib, h)


def leaf_graph(sequence):
    """make_base_graph + add_sentinels (src/modify_graph.cpp:30-77) of one sequence (str of ACGTN); host only"""
    lib = load_library()
    raw = sequence.encode() if isinstance(sequence, str) else bytes(sequence)
    h = C.c_void_p()
    rc = lib.cl_leaf_graph(raw, len(raw), C.byref(h))
    if rc != 0:
        raise ClError(rc)
    return _take_owned_base_graph(lib, h)


def explicit_cigar(graph1, graph2, pairs):
    """explicit_cigar(alignment, graph1, graph2) (include/centrolign/alignment.hpp:2804-2843); returns bytes"""
    lib = load_library()
    g1, g2 = graph1.as_c(), graph2.as_c()
    pairs = np.ascontiguousarray(pairs, np.uint64).reshape(-1, 2)
    p, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_explicit_cigar(C.byref(g1), C.byref(g2), pairs.ctypes.data, len(pairs), C.byref(p), C.byref(n))
    if rc != 0:
        raise ClError(rc)
    try:
        return C.string_at(p, int(n.value))
    finally:
        _libc_free(p)


def write_gfa(graph, path_names, decode=True):
    """write_gfa (include/centrolign/gfa.hpp:46-157); returns the GFA text as bytes"""
    lib = load_library()
    g = graph.as_c()
    names = (C.c_char_p * max(len(path_names), 1))(*[s.encode() for s in path_names])
    p, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_write_gfa(C.byref(g), names, int(decode), C.byref(p), C.byref(n))
    if rc != 0:
        raise ClError(rc)
    try:
        return C.string_at(p, int(n.value))
    finally:
        _libc_free(p)


def read_gfa(text, add_sentinels=True):
    """read_gfa + add_sentinels(graph, 5, 6) (src/gfa.cpp:9-96, src/modify_graph.cpp:47-77), as a -R restart loads a subproblem file:
    returns (BaseGraph, path names)"""
    lib = load_library()
    raw = text.encode() if isinstance(text, str) else bytes(text)
    h, names, n = C.c_void_p(), C.POINTER(C.c_char_p)(), C.c_uint64(0)
    lib.cl_read_gfa.restype = C.c_int
    lib.cl_read_gfa.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_uint64)]
    rc = lib.cl_read_gfa(raw, len(raw), int(add_sentinels), C.byref(h), C.byref(names), C.byref(n))
    if rc != 0:
        raise ClError(rc, "cl_read_gfa")
    out = [names[i].decode() for i in range(int(n.value))]
    for i in range(int(n.value)):
        _libc_free(C.c_void_p(C.cast(names, C.POINTER(C.c_void_p))[i]))
    _libc_free(C.cast(names, C.c_void_p))
    return _take_owned_base_graph(lib, h), out


def subproblem_hash_hex(sequence_names):
    """Execution::subproblem_hash through to_hex (src/execution.cpp:190-203): the <hash> of PREFIX_<hash>.gfa under -S / -R"""
    lib = load_library()
    arr = (C.c_char_p * max(len(sequence_names), 1))(*[s.encode() for s in sequence_names])
    buf = C.create_string_buffer(17)
    lib.cl_subproblem_hash_hex.restype = C.c_int
    rc = lib.cl_subproblem_hash_hex(arr, C.c_uint64(len(sequence_names)), buf)
    if rc != 0:
        raise ClError(rc, "cl_subproblem_hash_hex")
    return buf.value.decode()


def internal_fuse(graph, pairs):
    """internal_fuse (include/centrolign/fuse.hpp:144-247): the graph merged with itself along the alignment(s) (n, 2); host only.
    Returns (fused BaseGraph — possibly cyclic, trans [old node -> new node])"""
    lib = load_library()
    g = graph.as_c()
    pairs = np.ascontiguousarray(pairs, np.uint64).reshape(-1, 2)
    h = C.c_void_p()
    trans = np.zeros(len(graph.label), np.uint64)
    lib.cl_internal_fuse.restype = C.c_int
    lib.cl_internal_fuse.argtypes = [C.POINTER(BaseGraphC), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p]
    rc = lib.cl_internal_fuse(C.byref(g), pairs.ctypes.data, len(pairs), C.byref(h), trans.ctypes.data)
    if rc != 0:
        raise ClError(rc, "cl_internal_fuse")
    return _take_owned_base_graph(lib, h), trans


def induced_pairwise_cigar(graph, path1, path2):
    """explicit_cigar(induced_pairwise_alignment(graph, path1, path2), …) (src/alignment.cpp:84-229): the CLI's -A output for two of the
    sequences of an acyclic MSA graph; bytes"""
    lib = load_library()
    g, p, n = graph.as_c(), C.c_void_p(), C.c_uint64(0)
    lib.cl_induced_pairwise_cigar.restype = C.c_int
    lib.cl_induced_pairwise_cigar.argtypes = [C.POINTER(BaseGraphC), C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    rc = lib.cl_induced_pairwise_cigar(C.byref(g), int(path1), int(path2), C.byref(p), C.byref(n))
    if rc != 0:
        raise ClError(rc, "cl_induced_pairwise_cigar")
    try:
        return C.string_at(p, int(n.value))
    finally:
        _libc_free(p)


def graphs_equal(a, b):
    return a.src_id == b.src_id and a.snk_id == b.snk_id and all(np.array_equal(getattr(a, k), getattr(b, k)) for k in GRAPH_KEYS)


def match_params(max_count=3000, use_color_set_size=True, params=None):
    mp = MatchParams()
    mp.max_count = int(max_count)
    mp.use_color_set_size = int(use_color_set_size)
    mp.score = params or default_chain_params()
    return mp


def match_joined_text(graph1, graph2):
    """PathESA's joined path text (include/centrolign/path_esa.hpp:92-118) as a uint8 array; host only"""
    lib = load_library()
    g1, g2 = graph1.as_c(), graph2.as_c()
    p, n = C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_match_joined_text(C.byref(g1), C.byref(g2), C.byref(p), C.byref(n))
    if rc != 0:
        raise ClError(rc)
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(n.value),)).copy()
    finally:
        _libc_free(p)


def _libc_free(p):
    C.CDLL(None).free.argtypes = [C.c_void_p]
    C.CDLL(None).free(p)


def matches_from_suffix_array(graph1, graph2, sa, lcp, max_count=3000, params=None, want_stats=False):
    """the host half of cl_find_matches (LCP-interval tree, counts, query, walk-out) on a caller-supplied suffix array + LCP
    array of match_joined_text(graph1, graph2); host only"""
    lib = load_library()
    g1, g2, mp, st = graph1.as_c(), graph2.as_c(), match_params(max_count, True, params), MatchStats()
    sa, lcp = np.ascontiguousarray(sa, np.uint32), np.ascontiguousarray(lcp, np.uint32)
    h = C.c_void_p()
    rc = lib.cl_matches_from_suffix_array(C.byref(g1), C.byref(g2), C.byref(mp), sa.ctypes.data, lcp.ctypes.data, len(sa), C.byref(h), C.byref(st))
    if rc != 0:
        raise ClError(rc)
    ms = _take_owned_match_sets(lib, h)
    return (ms, st.as_dict()) if want_stats else ms


def extract_stitch_batch(graph1, graph2, segments):
    """Extractor::extract_graphs_between in Stitcher::stitch's consumption order (host only, no GPU needed)"""
    lib = load_library()
    g1, g2, sg = graph1.as_c(), graph2.as_c(), segments.as_c()
    h = C.c_void_p()
    rc = lib.cl_extract_stitch_batch(C.byref(g1), C.byref(g2), C.byref(sg), C.byref(h))
    if rc != 0:
        msg = lib.cl_last_error(None)
        raise ClError(rc, msg.decode() if msg else "")
    try:
        v = lib.cl_owned_batch_view(h).contents
        n = int(v.n_problems)
        od = np.ctypeslib.as_array(C.cast(v.only_deletion_alns, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
        return StitchBatch(_side_from_c(v.side[0], n), _side_from_c(v.side[1], n), od)
    finally:
        lib.cl_owned_batch_free(h)


def concat_stitch_batches(batches):
    """the subproblems of several batches as ONE batch, in order (node ids inside a problem are local to it, so the flat arrays simply
    follow one another with their offsets moved on): subproblems of different merges then share the plan's launches"""
    sides = []
    for si in (0, 1):
        arrs = {k: [] for k in _SIDE_DTYPES}
        base = {"node_off": 0, "prev_off": 0, "next_off": 0, "src_off": 0, "snk_off": 0}
        have_next = all(b.side[si].next_off is not None for b in batches)
        have_back = all(b.side[si].back_translation is not None for b in batches)
        for i, b in enumerate(batches):
            sd = b.side[si]
            last = i + 1 == len(batches)
            for off, idx in (("node_off", None), ("prev_off", "prev_idx"), ("next_off", "next_idx"), ("src_off", "src_idx"), ("snk_off", "snk_idx")):
                if off == "next_off" and not have_next:
                    continue
                a = getattr(sd, off).astype(np.uint64)
                arrs[off].append((a if last else a[:-1]) + np.uint64(base[off]))
                base[off] += int(a[-1])
                if idx:
                    arrs[idx].append(getattr(sd, idx))
            arrs["label"].append(sd.label)
            if have_back:
                arrs["back_translation"].append(sd.back_translation)
        sides.append(GraphSide(**{k: (np.concatenate(v) if v else None) for k, v in arrs.items()}))
    od = None
    if all(b.only_deletion_alns is not None for b in batches):
        od = np.concatenate([b.only_deletion_alns for b in batches])
    return StitchBatch(sides[0], sides[1], od)


class MatchSetsC(C.Structure):
    _fields_ = [("n_sets", C.c_uint64)] + [(n, C.c_void_p) for n in
                ("set_off1", "walk_off1", "nodes1", "set_off2", "walk_off2", "nodes2", "count1", "count2", "full_length")]


class ChainParams(C.Structure):
    """cl_chain_params: Anchorer gap parameters + ScoreFunction fields"""
    _fields_ = [("gap_open", C.c_double * 3), ("gap_extend", C.c_double * 3), ("anchor_score_function", C.c_int),
                ("pair_count_power", C.c_double), ("length_intercept", C.c_double), ("length_decay_power", C.c_double),
                ("global_anchoring", C.c_int)]


class ChainResultC(C.Structure):
    _fields_ = [("n_anchors", C.c_uint64), ("anchors", C.POINTER(C.c_uint32)), ("n_pairs", C.c_uint64),
                ("dp", C.POINTER(C.c_float)), ("n_ties", C.c_uint64), ("device_ms", C.c_float),
                ("prep_ms", C.c_float), ("index_ms", C.c_float), ("traceback_ms", C.c_float),
                ("gap_before_first", C.c_int64), ("gap_after_last", C.c_int64),
                ("gap_score_before_first", C.c_double), ("gap_score_after_last", C.c_double)]


class PartitionParams(C.Structure):
    """cl_partition_params"""
    _fields_ = [("constraint_method", C.c_int), ("minimum_segment_score", C.c_double), ("minimum_segment_average", C.c_double),
                ("window_length", C.c_double), ("generalized_length_mean", C.c_double), ("boundary_score_factor", C.c_double),
                ("score_scale", C.c_double), ("score_boundaries", C.c_int), ("use_annotated_score", C.c_int),
                ("score_function", ChainParams)]


class AnchorFieldsC(C.Structure):
    _fields_ = [("n_anchors", C.c_uint64)] + [(n, C.c_void_p) for n in
                ("walk_off", "walk1", "walk2", "count1", "count2", "full_length", "match_set", "score")]


class SplitParams(C.Structure):
    """cl_split_params: Anchorer::anchor_split_limit, min_split_length, min_path_length_spread, max_split_match_set_size"""
    _fields_ = [("anchor_split_limit", C.c_uint64), ("min_split_length", C.c_uint64), ("min_path_length_spread", C.c_uint64),
                ("max_split_match_set_size", C.c_uint64)]


class MatchParams(C.Structure):
    """cl_match_params: BaseMatchFinder::max_count, use_color_set_size + the ScoreFunction of the positive-weight filter"""
    _fields_ = [("max_count", C.c_uint64), ("use_color_set_size", C.c_int), ("score", ChainParams)]


class MatchStats(C.Structure):
    """cl_match_stats"""
    _fields_ = [("text_length", C.c_uint64), ("doubling_rounds", C.c_uint32), ("n_internal_nodes", C.c_uint64), ("n_candidates", C.c_uint64),
                ("sa_ms", C.c_float), ("lcp_ms", C.c_float), ("tree_ms", C.c_double), ("query_ms", C.c_double), ("walk_ms", C.c_double),
                ("text_ms", C.c_double), ("suffix_wall_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AnchorParams(C.Structure):
    """cl_anchor_params"""
    _fields_ = [("chain", ChainParams), ("max_num_match_pairs", C.c_uint64), ("score_scale", C.c_double),
                ("autocalibrate_gap_penalties", C.c_int), ("do_fill_in_anchoring", C.c_int), ("chaining_algorithm_plus_one", C.c_int)]


class AnchorChainResultC(C.Structure):
    _fields_ = [("n_anchors", C.c_uint64), ("anchors", C.POINTER(C.c_uint64)), ("gap_before", C.POINTER(C.c_int64)),
                ("gap_after", C.POINTER(C.c_int64)), ("gap_score_before", C.POINTER(C.c_double)),
                ("gap_score_after", C.POINTER(C.c_double)), ("score", C.POINTER(C.c_double)),
                ("count1", C.POINTER(C.c_uint64)), ("count2", C.POINTER(C.c_uint64)), ("full_length", C.POINTER(C.c_uint64)),
                ("walk_off", C.POINTER(C.c_uint64)), ("walk1", C.POINTER(C.c_uint32)), ("walk2", C.POINTER(C.c_uint32)),
                ("n_sets", C.c_uint64), ("set_order", C.POINTER(C.c_uint64)), ("scale", C.c_double), ("n_ties", C.c_uint64),
                ("fill_in_pairs", C.c_uint64), ("fill_in_device_ms", C.c_float), ("dp_device_ms", C.c_float), ("dp_pair_evals", C.c_double),
                ("dp_match_pairs", C.c_uint64), ("dp_combinations", C.c_uint32)]


class CoreAlignParams(C.Structure):
    """cl_core_align_params"""
    _fields_ = [("split_matches_at_branchpoints", C.c_int), ("split", SplitParams), ("anchor", AnchorParams),
                ("partition", PartitionParams), ("min_indel_fuzz_length", C.c_int64), ("indel_fuzz_score_proportion", C.c_double),
                ("stitch", StitchParams)]


class MergeParams(C.Structure):
    """cl_merge_params"""
    _fields_ = [("match", MatchParams), ("align", CoreAlignParams)]


class CoreAlignResultC(C.Structure):
    _fields_ = [("alignment", AlignmentC), ("n_segments", C.c_uint64), ("seg_off", C.POINTER(C.c_uint64)),
                ("walk_off", C.POINTER(C.c_uint64)), ("walk1", C.POINTER(C.c_uint32)), ("walk2", C.POINTER(C.c_uint32)),
                ("scale", C.c_double), ("n_chain_anchors", C.c_uint64), ("chain_ms", C.c_float), ("partition_ms", C.c_float),
                ("stitch_ms", C.c_float), ("chain_device_ms", C.c_float), ("chain_pair_evals", C.c_double), ("chain_match_pairs", C.c_uint64),
                ("chain_combinations", C.c_uint32), ("ref_restrain_memory", C.c_uint32), ("ref_packed_path_merge", C.c_uint32),
                ("ref_path_merge_widths", C.c_uint32)]


class MergeResultC(C.Structure):
    """cl_merge_result"""
    _fields_ = [("align", CoreAlignResultC), ("fused", C.c_void_p), ("n_match_sets", C.c_uint64), ("match_ms", C.c_float), ("align_ms", C.c_float),
                ("fuse_ms", C.c_float)]


def _core_align_dict(out):
    """CoreAlignResultC -> dict of numpy copies"""
    n, ns = int(out.alignment.n_pairs), int(out.n_segments)
    seg_off = np.ctypeslib.as_array(out.seg_off, shape=(ns + 1,)).copy()
    na = int(seg_off[-1])
    walk_off = np.ctypeslib.as_array(out.walk_off, shape=(na + 1,)).copy()
    nw = int(walk_off[-1])
    return dict(alignment=np.ctypeslib.as_array(out.alignment.pairs, shape=(max(n, 1) * 2,))[:2 * n].copy().reshape(n, 2),
                seg_off=seg_off, walk_off=walk_off,
                walk1=np.ctypeslib.as_array(out.walk1, shape=(max(nw, 1),))[:nw].copy(),
                walk2=np.ctypeslib.as_array(out.walk2, shape=(max(nw, 1),))[:nw].copy(),
                scale=float(out.scale), n_chain_anchors=int(out.n_chain_anchors), chain_ms=float(out.chain_ms),
                partition_ms=float(out.partition_ms), stitch_ms=float(out.stitch_ms), chain_device_ms=float(out.chain_device_ms),
                chain_pair_evals=float(out.chain_pair_evals), chain_match_pairs=int(out.chain_match_pairs),
                chain_combinations=int(out.chain_combinations), ref_restrain_memory=bool(out.ref_restrain_memory),
                ref_packed_path_merge=bool(out.ref_packed_path_merge), ref_path_merge_widths=int(out.ref_path_merge_widths))


def chain_exhaustive(graph1, graph2, matches, params=None, num_match_sets=None):
    """exhaustive_chain_dp + heaviest_weight_path (include/centrolign/anchorer.hpp:1342-1509, src/anchorer.cpp:68-133), the O(M^2)
    "-g 0" chaining; host only.  Returns the chain (n, 3) uint32 [match_set, idx1, idx2]"""
    lib = load_library()
    params = params or default_chain_params()
    g1, g2, mc, out = graph1.as_c(), graph2.as_c(), matches.as_c(), ChainResultC()
    n = matches.n_sets if num_match_sets is None else num_match_sets
    rc = lib.cl_chain_exhaustive(None, C.byref(g1), C.byref(g2), C.byref(mc), n, C.byref(params), C.byref(out))
    if rc != 0:
        raise ClError(rc)
    try:
        na = int(out.n_anchors)
        return np.ctypeslib.as_array(out.anchors, shape=(max(na, 1) * 3,))[:3 * na].copy().reshape(na, 3)
    finally:
        lib.cl_chain_result_free(C.byref(out))


class FastaC(C.Structure):
    """cl_fasta"""
    _fields_ = [("n_sequences", C.c_uint64), ("names", C.POINTER(C.c_char_p)), ("sequences", C.POINTER(C.c_char_p)),
                ("lengths", C.POINTER(C.c_uint64)), ("owner", C.c_void_p)]


class MsaPlanC(C.Structure):
    """cl_msa_plan"""
    _fields_ = [("n_leaves", C.c_uint64), ("leaf_sequence", C.POINTER(C.c_uint64)), ("n_merges", C.c_uint64),
                ("merge_children", C.POINTER(C.c_uint64))]


class BondParams(C.Structure):
    """cl_bond_params: Bonder's tunables as the CLI sets them (src/parameters.cpp:91-97)"""
    _fields_ = [("min_opt_proportion", C.c_double), ("include_gap_scores", C.c_int), ("min_length", C.c_double),
                ("deviation_drift_factor", C.c_double), ("separation_drift_factor", C.c_double),
                ("deduplication_slosh_proportion", C.c_double), ("trim_window_proportion", C.c_double)]


class PolishParams(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("max_tight_cycle_size", "max_bond_inconsistency_window", "min_inconsistency_disjoint_length",
                                          "min_inconsistency_total_length", "padding_target_min_length", "padding_max_length_limit")]


class MsaParams(C.Structure):
    """cl_msa_params"""
    _fields_ = [("merge", MergeParams), ("skip_calibration", C.c_int), ("n_workers", C.c_int), ("devices", C.POINTER(C.c_int)), ("n_devices", C.c_int),
                ("subproblems_prefix", C.c_char_p), ("restart", C.c_int),
                ("induced_pairwise_prefix", C.c_char_p), ("cyclize", C.c_int), ("max_tandem_duplication_search_rounds", C.c_uint64),
                ("bonds", BondParams), ("polish", PolishParams)]


class MsaStats(C.Structure):
    """cl_msa_stats"""
    _fields_ = [("n_merges", C.c_uint64), ("root_nodes", C.c_uint64), ("score_scale", C.c_double), ("calibration_s", C.c_double),
                ("match_s", C.c_double), ("align_s", C.c_double), ("fuse_s", C.c_double), ("total_s", C.c_double), ("n_restarted", C.c_uint64),
                ("n_bonds", C.c_uint64), ("n_polished_regions", C.c_uint64), ("bonds_s", C.c_double), ("cyclize_s", C.c_double)]


def parse_fasta(text):
    """parse_fasta (src/utility.cpp:19-65); host only.  Returns [(name, sequence)]"""
    lib = load_library()
    raw = text.encode() if isinstance(text, str) else bytes(text)
    f = FastaC()
    rc = lib.cl_parse_fasta(None, raw, len(raw), C.byref(f))
    if rc != 0:
        msg = lib.cl_last_error(None)
        raise ClError(rc, msg.decode() if msg else "")
    try:
        return [(f.names[i].decode(), C.string_at(f.sequences[i], int(f.lengths[i])).decode()) for i in range(int(f.n_sequences))]
    finally:
        lib.cl_fasta_free(C.byref(f))


def msa_plan(newick, names):
    """Tree(newick) + Execution's normalisation and order (src/tree.cpp, src/execution.cpp:12-92); host only.  Returns
    (leaf_sequence: index into names per leaf in calibration order, merges: [(slot of graph 1, slot of graph 2)] in execution order);
    slot i < len(leaf_sequence) is leaf i, slot len(leaf_sequence) + k the result of merge k"""
    lib = load_library()
    arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
    p = MsaPlanC()
    rc = lib.cl_msa_plan_create(None, (newick or "").encode(), arr, len(names), C.byref(p))
    if rc != 0:
        msg = lib.cl_last_error(None)
        raise ClError(rc, msg.decode() if msg else "")
    try:
        leaves = [int(p.leaf_sequence[i]) for i in range(int(p.n_leaves))]
        merges = [(int(p.merge_children[2 * k]), int(p.merge_children[2 * k + 1])) for k in range(int(p.n_merges))]
        return leaves, merges
    finally:
        lib.cl_msa_plan_free(C.byref(p))


def default_chain_params(global_anchoring=True):
    """the CLI's anchoring parameters (src/parameters.cpp:39-60)"""
    p = ChainParams()
    p.global_anchoring = int(global_anchoring)
    p.gap_open[:] = [1.25, 50.0, 5000.0]
    p.gap_extend[:] = [2.5, 0.1, 0.0015]
    p.anchor_score_function = 2  # ConcaveLengthScaleInverseCount
    p.pair_count_power = 0.5
    p.length_intercept = 2250.0
    p.length_decay_power = 2.0
    return p


class MatchSets:
    """numpy holder for cl_match_sets (std::vector<match_set_t>)"""
    _DT = dict(set_off1=np.uint64, walk_off1=np.uint64, nodes1=np.uint32, set_off2=np.uint64, walk_off2=np.uint64,
               nodes2=np.uint32, count1=np.uint64, count2=np.uint64, full_length=np.uint64)

    def __init__(self, **arrays):
        for k, dt in self._DT.items():
            setattr(self, k, np.ascontiguousarray(arrays[k], dtype=dt))

    @property
    def n_sets(self):
        return len(self.count1)

    def n_pairs(self):
        return int((np.diff(self.set_off1.astype(np.int64)) * np.diff(self.set_off2.astype(np.int64))).sum())

    def as_c(self):
        c = MatchSetsC()
        c.n_sets = self.n_sets
        for k in self._DT:
            setattr(c, k, getattr(self, k).ctypes.data)
        return c

    @staticmethod
    def from_dump(d, prefix):
        return MatchSets(**{k: d[prefix + "ms." + k] for k in MatchSets._DT})

    def reordered(self, set_order):
        """the sets in the order anchor_chain left the caller's vector in (set_order of its result): position k holds original set
        set_order[k].  The reference reorders in place (anchorer.hpp:1108-1173); here the caller does it when it needs the next call to
        start from that order (the tandem-duplication rounds of cyclisation)."""
        order = np.asarray(set_order, np.int64)
        out = {}
        for side in ("1", "2"):
            so, wo, nd = getattr(self, "set_off" + side).astype(np.int64), getattr(self, "walk_off" + side).astype(np.int64), getattr(self, "nodes" + side)
            n_walks = so[order + 1] - so[order]
            new_so = np.concatenate([[0], np.cumsum(n_walks)])
            walk_ids = np.concatenate([np.arange(so[s], so[s + 1]) for s in order]) if len(order) else np.zeros(0, np.int64)
            lens = wo[walk_ids + 1] - wo[walk_ids]
            new_wo = np.concatenate([[0], np.cumsum(lens)])
            node_ids = np.concatenate([np.arange(wo[w], wo[w + 1]) for w in walk_ids]) if len(walk_ids) else np.zeros(0, np.int64)
            out["set_off" + side], out["walk_off" + side], out["nodes" + side] = new_so, new_wo, nd[node_ids]
        for k in ("count1", "count2", "full_length"):
            out[k] = getattr(self, k)[order]
        return MatchSets(**out)


class LaunchInfo(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("n_problems", C.c_uint64), ("dp_cells", C.c_uint64),
                ("dp_bytes", C.c_uint64), ("last_ms", C.c_float), ("in_pass_ms", C.c_float), ("lds_bytes", C.c_uint32), ("max_sweep", C.c_uint32),
                ("max_n1", C.c_uint32), ("max_n2", C.c_uint32), ("event_ms", C.c_float)]


_SIDE_DTYPES = dict(node_off=np.uint64, label=np.uint8, prev_off=np.uint64, prev_idx=np.uint32,
                    next_off=np.uint64, next_idx=np.uint32, src_off=np.uint64, src_idx=np.uint32,
                    snk_off=np.uint64, snk_idx=np.uint32, back_translation=np.uint64)


class GraphSide:
    """numpy holder for one cl_graph_side"""

    def __init__(self, **arrays):
        for name, dt in _SIDE_DTYPES.items():
            a = arrays.get(name)
            setattr(self, name, None if a is None else np.ascontiguousarray(a, dtype=dt))

    @property
    def n_problems(self):
        return len(self.node_off) - 1

    def as_c(self):
        c = GraphSideC()
        for name in _SIDE_DTYPES:
            a = getattr(self, name)
            setattr(c, name, None if a is None else a.ctypes.data)
        return c

    def problem(self, k):
        """(n, label, prev lists, sources, sinks) of problem k as python objects (debug / tiny tests)"""
        b, e = int(self.node_off[k]), int(self.node_off[k + 1])
        prevs = [list(map(int, self.prev_idx[int(self.prev_off[v]):int(self.prev_off[v + 1])])) for v in range(b, e)]
        src = list(map(int, self.src_idx[int(self.src_off[k]):int(self.src_off[k + 1])]))
        snk = list(map(int, self.snk_idx[int(self.snk_off[k]):int(self.snk_off[k + 1])]))
        return e - b, bytes(self.label[b:e]), prevs, src, snk


class StitchBatch:
    """numpy holder for a cl_stitch_batch"""

    def __init__(self, side1, side2, only_deletion_alns=None):
        assert side1.n_problems == side2.n_problems
        self.side = (side1, side2)
        self.only_deletion_alns = (None if only_deletion_alns is None
                                   else np.ascontiguousarray(only_deletion_alns, dtype=np.uint8))

    @property
    def n_problems(self):
        return self.side[0].n_problems

    def sizes(self):
        n1 = np.diff(self.side[0].node_off).astype(np.int64)
        n2 = np.diff(self.side[1].node_off).astype(np.int64)
        return n1, n2

    def dp_cells(self):
        n1, n2 = self.sizes()
        m = (n1 > 0) & (n2 > 0)
        return int(((n1[m] + 1) * (n2[m] + 1)).sum())

    def as_c(self):
        c = StitchBatchC()
        c.n_problems = self.n_problems
        c.side[0] = self.side[0].as_c()
        c.side[1] = self.side[1].as_c()
        c.only_deletion_alns = None if self.only_deletion_alns is None else self.only_deletion_alns.ctypes.data
        return c

    def subset(self, idx):
        """a new batch holding the problems idx (in that order)"""
        idx = np.asarray(idx, dtype=np.int64)
        sides = []
        for s in self.side:
            sides.append(_subset_side(s, idx))
        od = None if self.only_deletion_alns is None else self.only_deletion_alns[idx]
        return StitchBatch(sides[0], sides[1], od)

    @staticmethod
    def from_c(bc):
        """a StitchBatch over COPIES of the arrays of a cl_stitch_batch the library handed to a callback (StitchBatchC)"""
        n = int(bc.n_problems)

        def arr(ptr, dt, count):
            if not ptr:
                return None
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(max(int(count), 1),))[:int(count)].copy()
        sides = []
        for si in range(2):
            s = bc.side[si]
            node_off = arr(s.node_off, np.uint64, n + 1)
            nodes = int(node_off[-1])
            prev_off = arr(s.prev_off, np.uint64, nodes + 1)
            next_off = arr(s.next_off, np.uint64, nodes + 1)
            src_off, snk_off = arr(s.src_off, np.uint64, n + 1), arr(s.snk_off, np.uint64, n + 1)
            sides.append(GraphSide(node_off=node_off, label=arr(s.label, np.uint8, nodes), prev_off=prev_off, prev_idx=arr(s.prev_idx, np.uint32, prev_off[-1]),
                                   next_off=next_off, next_idx=None if next_off is None else arr(s.next_idx, np.uint32, next_off[-1]),
                                   src_off=src_off, src_idx=arr(s.src_idx, np.uint32, src_off[-1]), snk_off=snk_off, snk_idx=arr(s.snk_idx, np.uint32, snk_off[-1]),
                                   back_translation=arr(s.back_translation, np.uint64, nodes)))
        return StitchBatch(sides[0], sides[1], arr(bc.only_deletion_alns, np.uint8, n))

    @staticmethod
    def concat(batches):
        sides = []
        for si in range(2):
            sides.append(_concat_sides([b.side[si] for b in batches]))
        if any(b.only_deletion_alns is not None for b in batches):
            od = np.concatenate([b.only_deletion_alns if b.only_deletion_alns is not None
                                 else np.zeros(b.n_problems, np.uint8) for b in batches])
        else:
            od = None
        return StitchBatch(sides[0], sides[1], od)


def _ranges(starts, lens):
    """concatenated aranges: for each i, starts[i] .. starts[i]+lens[i]-1"""
    lens = np.asarray(lens, dtype=np.int64)
    starts = np.asarray(starts, dtype=np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    out_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    rep = np.repeat(starts - out_off, lens)
    return rep + np.arange(total, dtype=np.int64)


def _subset_side(s, idx):
    nb = s.node_off[idx].astype(np.int64)
    nl = (s.node_off[idx + 1] - s.node_off[idx]).astype(np.int64)
    nodes = _ranges(nb, nl)
    out = {}
    out["node_off"] = np.concatenate([[0], np.cumsum(nl)]).astype(np.uint64)
    out["label"] = s.label[nodes]
    for pre in ("prev", "next"):
        off = getattr(s, pre + "_off")
        if off is None:
            continue
        ib = off[nodes].astype(np.int64)
        il = (off[nodes + 1] - off[nodes]).astype(np.int64) if len(nodes) else np.zeros(0, np.int64)
        out[pre + "_off"] = np.concatenate([[0], np.cumsum(il)]).astype(np.uint64)
        out[pre + "_idx"] = getattr(s, pre + "_idx")[_ranges(ib, il)]
    for pre in ("src", "snk"):
        off = getattr(s, pre + "_off")
        ib = off[idx].astype(np.int64)
        il = (off[idx + 1] - off[idx]).astype(np.int64)
        out[pre + "_off"] = np.concatenate([[0], np.cumsum(il)]).astype(np.uint64)
        out[pre + "_idx"] = getattr(s, pre + "_idx")[_ranges(ib, il)]
    if s.back_translation is not None:
        out["back_translation"] = s.back_translation[nodes]
    return GraphSide(**out)


def _concat_sides(sides):
    out = {}
    out["label"] = np.concatenate([s.label for s in sides])

    def cat_off(name):
        parts, base = [np.zeros(1, np.uint64)], np.uint64(0)
        for s in sides:
            o = getattr(s, name)
            parts.append(o[1:] + base)
            base = base + o[-1]
        return np.concatenate(parts)

    out["node_off"] = cat_off("node_off")
    for pre in ("prev", "next"):
        if any(getattr(s, pre + "_off") is None for s in sides):
            continue
        out[pre + "_off"] = cat_off(pre + "_off")
        out[pre + "_idx"] = np.concatenate([getattr(s, pre + "_idx") for s in sides])
    for pre in ("src", "snk"):
        out[pre + "_off"] = cat_off(pre + "_off")
        out[pre + "_idx"] = np.concatenate([getattr(s, pre + "_idx") for s in sides])
    if all(s.back_translation is not None for s in sides):
        out["back_translation"] = np.concatenate([s.back_translation for s in sides])
    return GraphSide(**out)


class StitchResult:
    """numpy copy of a cl_stitch_result"""

    def __init__(self, aln_off, pairs, score, route, num_pw):
        self.aln_off, self.pairs, self.score, self.route, self.num_pw = aln_off, pairs, score, route, num_pw

    @property
    def n_problems(self):
        return len(self.aln_off) - 1

    def alignment(self, k):
        b, e = int(self.aln_off[k]), int(self.aln_off[k + 1])
        return self.pairs[b:e]

    @staticmethod
    def from_c(rc):
        n = int(rc.n_problems)
        aln_off = np.ctypeslib.as_array(rc.aln_off, shape=(n + 1,)).copy()
        tot = int(aln_off[-1])
        pairs = (np.ctypeslib.as_array(rc.pairs, shape=(max(tot, 1) * 2,))[:tot * 2].copy().reshape(tot, 2))
        score = np.ctypeslib.as_array(rc.score, shape=(max(n, 1),))[:n].copy()
        route = np.ctypeslib.as_array(rc.route, shape=(max(n, 1),))[:n].copy()
        num_pw = np.ctypeslib.as_array(rc.num_pw, shape=(max(n, 1),))[:n].copy()
        return StitchResult(aln_off, pairs, score, route, num_pw)

    def same_as(self, other, check_score=True, check_route=True):
        """None if identical, else a short description of the first difference"""
        if self.n_problems != other.n_problems:
            return "n_problems %d vs %d" % (self.n_problems, other.n_problems)
        if not np.array_equal(self.aln_off, other.aln_off):
            k = int(np.argmax(np.diff(self.aln_off.astype(np.int64)) != np.diff(other.aln_off.astype(np.int64))))
            return "alignment length differs first at problem %d" % k
        if not np.array_equal(self.pairs, other.pairs):
            row = int(np.argmax((self.pairs != other.pairs).any(axis=1)))
            k = int(np.searchsorted(self.aln_off, row, side="right") - 1)
            return "aligned pair differs first at problem %d (pair %d): %s vs %s" % (
                k, row - int(self.aln_off[k]), self.pairs[row], other.pairs[row])
        if check_route and not np.array_equal(self.route, other.route):
            return "route differs at problem %d" % int(np.argmax(self.route != other.route))
        if check_route and not np.array_equal(self.num_pw, other.num_pw):
            return "num_pw differs at problem %d" % int(np.argmax(self.num_pw != other.num_pw))
        if check_score:
            m = self.route == 0
            if not np.array_equal(self.score[m], other.score[m]):
                k = int(np.flatnonzero(m)[np.argmax(self.score[m] != other.score[m])])
                return "score differs at problem %d: %d vs %d" % (k, self.score[k], other.score[k])
        return None


_lib = None
ABI_VERSION = 12    # CL_ABI_VERSION of include/centrolign_amd.h these ctypes structures mirror


def load_library(path=None):
    """dlopen the HIP product library; raises if it is not built (no fallback by design)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("centrolign_amd HIP library not built: %s is missing "
                          "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C centrolign_amd/csrc`)" % p)
    lib = C.CDLL(p)
    lib.cl_abi_version.restype = C.c_int
    if lib.cl_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, these bindings are written for %d (include/centrolign_amd.h: CL_ABI_VERSION): rebuild the library"
                          % (p, lib.cl_abi_version(), ABI_VERSION))
    lib.cl_device_count.restype = C.c_int
    lib.cl_context_create.restype = C.c_void_p
    lib.cl_context_create.argtypes = [C.c_int]
    lib.cl_context_destroy.argtypes = [C.c_void_p]
    lib.cl_last_error.restype = C.c_char_p
    lib.cl_last_error.argtypes = [C.c_void_p]
    lib.cl_device_name.restype = C.c_char_p
    lib.cl_device_name.argtypes = [C.c_void_p]
    lib.cl_stitch_params_default.argtypes = [C.POINTER(StitchParams)]
    lib.cl_po_poa_batch.restype = C.c_int
    lib.cl_po_poa_batch.argtypes = [C.c_void_p, C.POINTER(StitchBatchC), C.c_void_p, C.POINTER(AlignParams),
                                    C.POINTER(StitchResultC)]
    lib.cl_stitch_batch_align.restype = C.c_int
    lib.cl_stitch_batch_align.argtypes = [C.c_void_p, C.POINTER(StitchBatchC), C.POINTER(StitchParams),
                                          C.POINTER(StitchResultC)]
    lib.cl_stitch_result_free.argtypes = [C.POINTER(StitchResultC)]
    lib.cl_stitch_plan_create.restype = C.c_int
    lib.cl_stitch_plan_create.argtypes = [C.c_void_p, C.POINTER(StitchBatchC), C.POINTER(StitchParams), C.c_void_p,
                                          C.POINTER(C.c_void_p)]
    lib.cl_stitch_plan_execute.restype = C.c_int
    lib.cl_stitch_plan_execute.argtypes = [C.c_void_p, C.c_void_p]
    lib.cl_stitch_plan_execute_profiled.restype = C.c_int
    lib.cl_stitch_plan_execute_profiled.argtypes = [C.c_void_p, C.c_void_p]
    lib.cl_stitch_plan_execute_evented.restype = C.c_int
    lib.cl_stitch_plan_execute_evented.argtypes = [C.c_void_p, C.c_void_p]
    lib.cl_stitch_plan_sync.restype = C.c_int
    lib.cl_stitch_plan_sync.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.cl_stitch_plan_collect.restype = C.c_int
    lib.cl_stitch_plan_collect.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(StitchResultC)]
    lib.cl_stitch_plan_destroy.argtypes = [C.c_void_p, C.c_void_p]
    lib.cl_stitch_plan_stats.restype = C.c_int
    lib.cl_stitch_plan_stats.argtypes = [C.c_void_p, C.POINTER(PlanStats)]
    lib.cl_stitch_plan_launch_count.restype = C.c_int
    lib.cl_stitch_plan_launch_count.argtypes = [C.c_void_p]
    lib.cl_stitch_plan_launch_info.restype = C.c_int
    lib.cl_stitch_plan_launch_info.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(LaunchInfo)]
    lib.cl_extract_stitch_batch.restype = C.c_int
    lib.cl_extract_stitch_batch.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(AnchorSegmentsC),
                                            C.POINTER(C.c_void_p)]
    lib.cl_owned_batch_view.restype = C.POINTER(StitchBatchC)
    lib.cl_owned_batch_view.argtypes = [C.c_void_p]
    lib.cl_owned_batch_free.argtypes = [C.c_void_p]
    lib.cl_stitch.restype = C.c_int
    lib.cl_stitch.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(AnchorSegmentsC),
                              C.POINTER(StitchParams), C.POINTER(AlignmentC)]
    lib.cl_alignment_free.argtypes = [C.POINTER(AlignmentC)]
    lib.cl_chain_params_default.argtypes = [C.POINTER(ChainParams)]
    lib.cl_chain_sparse_affine.restype = C.c_int
    lib.cl_chain_sparse_affine.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.c_uint64,
                                           C.POINTER(ChainParams), C.c_double, C.c_int, C.POINTER(ChainResultC)]
    lib.cl_chain_exhaustive.restype = C.c_int
    lib.cl_chain_exhaustive.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.c_uint64,
                                        C.POINTER(ChainParams), C.POINTER(ChainResultC)]
    lib.cl_chain_sparse.restype = C.c_int
    lib.cl_chain_sparse.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.c_uint64,
                                    C.POINTER(ChainParams), C.c_int, C.POINTER(ChainResultC)]
    lib.cl_chain_result_free.argtypes = [C.POINTER(ChainResultC)]
    lib.cl_core_align_params_default.restype = None
    lib.cl_core_align_params_default.argtypes = [C.POINTER(CoreAlignParams)]
    lib.cl_core_align.restype = C.c_int
    lib.cl_core_align.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.POINTER(CoreAlignParams),
                                  C.POINTER(CoreAlignResultC)]
    lib.cl_core_align_result_free.restype = None
    lib.cl_core_align_result_free.argtypes = [C.POINTER(CoreAlignResultC)]
    lib.cl_partition_params_default.restype = None
    lib.cl_partition_params_default.argtypes = [C.POINTER(PartitionParams)]
    lib.cl_partition_anchors.restype = C.c_int
    lib.cl_partition_anchors.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(AnchorFieldsC), C.POINTER(PartitionParams),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.cl_split_params_default.restype = None
    lib.cl_split_params_default.argtypes = [C.POINTER(SplitParams)]
    lib.cl_split_branching_matches.restype = C.c_int
    lib.cl_split_branching_matches.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.POINTER(SplitParams),
                                               C.POINTER(C.c_void_p)]
    lib.cl_owned_match_sets_view.restype = None
    lib.cl_owned_match_sets_view.argtypes = [C.c_void_p, C.POINTER(MatchSetsC)]
    lib.cl_owned_match_sets_free.restype = None
    lib.cl_owned_match_sets_free.argtypes = [C.c_void_p]
    lib.cl_fuse.restype = C.c_int
    lib.cl_fuse.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.cl_owned_base_graph_view.restype = None
    lib.cl_owned_base_graph_view.argtypes = [C.c_void_p, C.POINTER(BaseGraphC)]
    lib.cl_owned_base_graph_free.restype = None
    lib.cl_owned_base_graph_free.argtypes = [C.c_void_p]
    lib.cl_leaf_graph.restype = C.c_int
    lib.cl_leaf_graph.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.cl_explicit_cigar.restype = C.c_int
    lib.cl_explicit_cigar.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.cl_write_gfa.restype = C.c_int
    lib.cl_write_gfa.argtypes = [C.POINTER(BaseGraphC), C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.cl_estimate_score_scale.restype = C.c_int
    lib.cl_estimate_score_scale.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.POINTER(AnchorParams),
                                            C.POINTER(C.c_double)]
    lib.cl_leaf_intrinsic_scale.restype = C.c_int
    lib.cl_leaf_intrinsic_scale.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(MatchParams), C.POINTER(AnchorParams), C.POINTER(C.c_double)]
    lib.cl_merge_params_default.restype = None
    lib.cl_merge_params_default.argtypes = [C.POINTER(MergeParams)]
    lib.cl_merge.restype = C.c_int
    lib.cl_merge.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MergeParams), C.POINTER(MergeResultC)]
    lib.cl_merge_result_free.restype = None
    lib.cl_merge_result_free.argtypes = [C.POINTER(MergeResultC)]
    lib.cl_parse_fasta.restype = C.c_int
    lib.cl_parse_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(FastaC)]
    lib.cl_fasta_free.restype = None
    lib.cl_fasta_free.argtypes = [C.POINTER(FastaC)]
    lib.cl_msa_plan_create.restype = C.c_int
    lib.cl_msa_plan_create.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint64, C.POINTER(MsaPlanC)]
    lib.cl_msa_plan_free.restype = None
    lib.cl_msa_plan_free.argtypes = [C.POINTER(MsaPlanC)]
    lib.cl_msa_params_default.restype = None
    lib.cl_msa_params_default.argtypes = [C.POINTER(MsaParams)]
    lib.cl_msa.restype = C.c_int
    lib.cl_msa.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(MsaParams), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                           C.POINTER(MsaStats)]
    lib.cl_match_params_default.restype = None
    lib.cl_match_params_default.argtypes = [C.POINTER(MatchParams)]
    lib.cl_find_matches.restype = C.c_int
    lib.cl_find_matches.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchParams), C.POINTER(C.c_void_p),
                                    C.POINTER(MatchStats)]
    lib.cl_match_joined_text.restype = C.c_int
    lib.cl_match_joined_text.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.cl_suffix_array_lcp.restype = C.c_int
    lib.cl_suffix_array_lcp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    lib.cl_matches_from_suffix_array.restype = C.c_int
    lib.cl_matches_from_suffix_array.argtypes = [C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchParams), C.c_void_p, C.c_void_p,
                                                 C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(MatchStats)]
    lib.cl_anchor_chain.restype = C.c_int
    lib.cl_anchor_chain.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC),
                                    C.POINTER(AnchorParams), C.POINTER(AnchorChainResultC)]
    lib.cl_anchor_chain_result_free.restype = None
    lib.cl_anchor_chain_result_free.argtypes = [C.POINTER(AnchorChainResultC)]
    lib.cl_anchor_chain_masked.restype = C.c_int
    lib.cl_anchor_chain_masked.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(BaseGraphC), C.POINTER(MatchSetsC), C.POINTER(AnchorParams),
                                           C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(AnchorChainResultC)]
    lib.cl_generate_diagonal_mask.restype = C.c_int
    lib.cl_generate_diagonal_mask.argtypes = [C.POINTER(MatchSetsC), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.cl_update_mask.restype = C.c_int
    lib.cl_update_mask.argtypes = [C.POINTER(MatchSetsC), C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.cl_internal_stitch.restype = C.c_int
    lib.cl_internal_stitch.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(StitchParams),
                                       C.POINTER(AlignmentC)]
    if path is None:
        _lib = lib
    return lib


def fallback_counters(reset=False):
    """cl_fallback_counters: dict(strip_fallbacks, walk_stalls, chain_dps, stitch_plans, strip_pairs, bond_trims_past_the_end) of this process"""
    lib = load_library()
    st = (C.c_uint64 * 6)()
    lib.cl_fallback_counters.argtypes = [C.c_void_p, C.c_int]
    lib.cl_fallback_counters.restype = None
    lib.cl_fallback_counters(st, int(bool(reset)))
    return dict(zip(("strip_fallbacks", "walk_stalls", "chain_dps", "stitch_plans", "strip_pairs", "bond_trims_past_the_end"), [int(x) for x in st]))


EXPORTED_SYMBOLS = [
    "cl_abi_version", "cl_device_count", "cl_context_create", "cl_context_destroy", "cl_last_error",
    "cl_context_peer_export", "cl_context_peer_group", "cl_context_peer_stats", "cl_context_peer_selftest", "cl_context_peer_steal", "cl_context_memory", "cl_fallback_counters",
    "cl_device_name", "cl_stitch_params_default", "cl_po_poa_batch", "cl_stitch_batch_align",
    "cl_stitch_result_free", "cl_stitch_result_alloc", "cl_context_set_stitch_hook", "cl_stitch_plan_create", "cl_stitch_plan_execute", "cl_stitch_plan_execute_profiled", "cl_stitch_plan_execute_evented", "cl_stitch_plan_sync",
    "cl_stitch_plan_collect", "cl_stitch_plan_destroy", "cl_stitch_plan_stats",
    "cl_stitch_plan_launch_count", "cl_stitch_plan_launch_info",
    "cl_extract_stitch_batch", "cl_owned_batch_view", "cl_owned_batch_free", "cl_stitch", "cl_alignment_free",
    "cl_despecify_indel_breakpoints", "cl_chain_params_default", "cl_chain_sparse_affine", "cl_chain_sparse", "cl_chain_exhaustive", "cl_chain_result_free",
    "cl_parse_fasta", "cl_fasta_free", "cl_msa_plan_create", "cl_msa_plan_free", "cl_msa_params_default", "cl_msa",
    "cl_read_gfa", "cl_subproblem_hash_hex", "cl_internal_fuse", "cl_induced_pairwise_cigar",
    "cl_anchor_chain", "cl_anchor_chain_result_free", "cl_anchor_chain_masked", "cl_generate_diagonal_mask", "cl_update_mask", "cl_internal_stitch",
    "cl_partition_params_default", "cl_partition_anchors", "cl_host_route_align", "cl_stitch_rank_order",
    "cl_core_align_params_default", "cl_core_align", "cl_core_align_result_free",
    "cl_split_params_default", "cl_split_branching_matches", "cl_owned_match_sets_view", "cl_owned_match_sets_free",
    "cl_estimate_score_scale", "cl_leaf_intrinsic_scale", "cl_leaf_graph", "cl_explicit_cigar", "cl_write_gfa",
    "cl_fuse", "cl_owned_base_graph_view", "cl_owned_base_graph_free", "cl_merge_params_default", "cl_merge", "cl_merge_result_free",
    "cl_match_params_default", "cl_find_matches", "cl_match_joined_text", "cl_suffix_array_lcp", "cl_matches_from_suffix_array",
    "cl_bond_params_default", "cl_identify_bonds", "cl_bonds_free", "cl_leaf_calibrate", "cl_leaf_calibration_free", "cl_alignment_list_free",
    "cl_leaf_bond_alignments", "cl_simplify_bubbles", "cl_apply_bonds", "cl_polish_params_default", "cl_identify_inconsistencies",
    "cl_polish_cyclized_graph",
]


class ChainAnchorsC(C.Structure):
    _fields_ = [("n", C.c_uint64), ("walk_off", C.c_void_p), ("walk1", C.c_void_p), ("walk2", C.c_void_p), ("score", C.c_void_p),
                ("gap_after", C.c_void_p), ("gap_score_after", C.c_void_p)]


class BondsC(C.Structure):
    _fields_ = [("n_intervals", C.c_uint64), ("interval_off", C.POINTER(C.c_uint64)), ("offset1", C.POINTER(C.c_uint64)),
                ("offset2", C.POINTER(C.c_uint64)), ("length", C.POINTER(C.c_uint64)), ("score", C.POINTER(C.c_double))]


class AlignmentListC(C.Structure):
    _fields_ = [("n", C.c_uint64), ("alignments", C.POINTER(AlignmentC))]


def bond_params(min_length=None, **kw):
    lib = load_library()
    bp = BondParams()
    lib.cl_bond_params_default(C.byref(bp))
    if min_length is not None:
        bp.min_length = float(min_length)
    for k, v in kw.items():
        setattr(bp, k, v)
    return bp


def _chain_anchors(chain):
    """dict(walk_off, walk1, walk2, score, gap_after, gap_score_after) -> (ChainAnchorsC, keep-alive arrays)"""
    keep = [np.ascontiguousarray(chain["walk_off"], np.uint64), np.ascontiguousarray(chain["walk1"], np.uint32),
            np.ascontiguousarray(chain["walk2"], np.uint32), np.ascontiguousarray(chain["score"], np.float64),
            np.ascontiguousarray(chain["gap_after"], np.int64), np.ascontiguousarray(chain["gap_score_after"], np.float64)]
    c = ChainAnchorsC(max(len(keep[0]) - 1, 0), *[a.ctypes.data for a in keep])
    return c, keep


def identify_bonds(leaf, opt_chain, secondary_chain, params=None, deduplicate=True):
    """Bonder::identify_bonds (+ deduplicate_self_bonds) on a leaf against itself (include/centrolign/bonder.hpp:116-452, src/bonder.cpp:473-551);
    host only.  Chains: dicts of walk_off, walk1, walk2, score, gap_after, gap_score_after.  Returns dict(interval_off, offset1, offset2, length, score)"""
    lib = load_library()
    lib.cl_identify_bonds.restype = C.c_int
    lib.cl_identify_bonds.argtypes = [C.POINTER(BaseGraphC), C.POINTER(ChainAnchorsC), C.POINTER(ChainAnchorsC), C.POINTER(BondParams), C.c_int, C.POINTER(BondsC)]
    params = params or bond_params()
    g = leaf.as_c()
    oc, k1 = _chain_anchors(opt_chain)
    sc, k2 = _chain_anchors(secondary_chain)
    out = BondsC()
    rc = lib.cl_identify_bonds(C.byref(g), C.byref(oc), C.byref(sc), C.byref(params), int(deduplicate), C.byref(out))
    if rc:
        raise ClError(rc, "cl_identify_bonds")
    try:
        n = int(out.n_intervals)
        off = np.ctypeslib.as_array(out.interval_off, shape=(n + 1,)).copy()
        t = int(off[-1])

        def arr(ptr):
            return np.ctypeslib.as_array(ptr, shape=(max(t, 1),))[:t].copy()
        return dict(interval_off=off, offset1=arr(out.offset1), offset2=arr(out.offset2), length=arr(out.length), score=arr(out.score))
    finally:
        lib.cl_bonds_free(C.byref(out))


def simplify_bubbles(graph):
    """simplify_bubbles + purge_uncovered_nodes (src/modify_graph.cpp:89-382); host only"""
    lib = load_library()
    lib.cl_simplify_bubbles.restype = C.c_int
    lib.cl_simplify_bubbles.argtypes = [C.POINTER(BaseGraphC), C.POINTER(C.c_void_p)]
    g, h = graph.as_c(), C.c_void_p()
    rc = lib.cl_simplify_bubbles(C.byref(g), C.byref(h))
    if rc:
        raise ClError(rc, "cl_simplify_bubbles")
    return _take_owned_base_graph(lib, h)


def apply_bonds(root, path_of_alignment, alignments):
    """Core::apply_bonds up to the polishing step (src/core.cpp:613-645): bond alignments in path positions -> fused and simplified graph; host only"""
    lib = load_library()
    lib.cl_apply_bonds.restype = C.c_int
    lib.cl_apply_bonds.argtypes = [C.POINTER(BaseGraphC), C.c_uint64, C.c_void_p, C.POINTER(AlignmentC), C.POINTER(C.c_void_p)]
    g, h = root.as_c(), C.c_void_p()
    keep = [np.ascontiguousarray(a, np.uint64).reshape(-1, 2) for a in alignments]
    arr = (AlignmentC * max(len(keep), 1))()
    for i, a in enumerate(keep):
        arr[i].n_pairs = len(a)
        arr[i].pairs = a.ctypes.data_as(C.POINTER(C.c_uint64))
    pa = np.ascontiguousarray(path_of_alignment, np.uint64)
    rc = lib.cl_apply_bonds(C.byref(g), len(keep), pa.ctypes.data, arr, C.byref(h))
    if rc:
        raise ClError(rc, "cl_apply_bonds")
    return _take_owned_base_graph(lib, h)


def polish_params(**kw):
    lib = load_library()
    pp = PolishParams()
    lib.cl_polish_params_default(C.byref(pp))
    for k, v in kw.items():
        setattr(pp, k, int(v))
    return pp


def identify_inconsistencies(graph, params=None):
    """InconsistencyIdentifier::identify_inconsistencies (include/centrolign/inconsistency_identifier.hpp:66-187); host only; (n, 2) node pairs"""
    lib = load_library()
    lib.cl_identify_inconsistencies.restype = C.c_int
    lib.cl_identify_inconsistencies.argtypes = [C.POINTER(BaseGraphC), C.POINTER(PolishParams), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    params = params or polish_params()
    g, ptr, n = graph.as_c(), C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_identify_inconsistencies(C.byref(g), C.byref(params), C.byref(ptr), C.byref(n))
    if rc:
        raise ClError(rc, "cl_identify_inconsistencies")
    k = int(n.value)
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(max(k, 1) * 2,))[:2 * k].copy().reshape(k, 2)
    _libc_free(ptr)
    return a


def _take_mask(lib, ptr, n):
    k = int(n.value)
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(max(k, 1) * 3,))[:3 * k].copy().reshape(k, 3)
    _libc_free(ptr)
    return a


def generate_diagonal_mask(matches):
    """Core::generate_diagonal_mask (src/core.cpp:301-321): (n, 3) [set, idx1, idx2], sorted"""
    lib = load_library()
    mc, ptr, n = matches.as_c(), C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_generate_diagonal_mask(C.byref(mc), C.byref(ptr), C.byref(n))
    if rc:
        raise ClError(rc, "cl_generate_diagonal_mask")
    return _take_mask(lib, ptr, n)


def update_mask(matches, chain_walk1, chain_walk2, mask, mask_reciprocal=False):
    """Core::update_mask (src/core.cpp:323-372): the mask grown by every pair that shares an aligned node pair with the chain"""
    lib = load_library()
    w1, w2 = np.ascontiguousarray(chain_walk1, np.uint32), np.ascontiguousarray(chain_walk2, np.uint32)
    m = np.ascontiguousarray(mask, np.uint64).reshape(-1, 3)
    mc, ptr, n = matches.as_c(), C.c_void_p(), C.c_uint64(0)
    rc = lib.cl_update_mask(C.byref(mc), len(w1), w1.ctypes.data, w2.ctypes.data, int(mask_reciprocal), m.ctypes.data, len(m), C.byref(ptr), C.byref(n))
    if rc:
        raise ClError(rc, "cl_update_mask")
    return _take_mask(lib, ptr, n)


class Plan:
    def __init__(self, ctx, handle, batch):
        self.ctx, self.handle, self._batch = ctx, handle, batch

    def execute(self):
        self.ctx._check(self.ctx.lib.cl_stitch_plan_execute(self.ctx.handle, self.handle))

    def execute_profiled(self):
        self.ctx._check(self.ctx.lib.cl_stitch_plan_execute_profiled(self.ctx.handle, self.handle))

    def execute_evented(self):
        """one concurrent pass with HIP events round every launch on its stream: launches()[i]["event_ms"] after sync()"""
        self.ctx._check(self.ctx.lib.cl_stitch_plan_execute_evented(self.ctx.handle, self.handle))

    def sync(self):
        ms = C.c_float(0)
        self.ctx._check(self.ctx.lib.cl_stitch_plan_sync(self.ctx.handle, self.handle, C.byref(ms)))
        return float(ms.value)

    def collect(self):
        rc = StitchResultC()
        self.ctx._check(self.ctx.lib.cl_stitch_plan_collect(self.ctx.handle, self.handle, C.byref(rc)))
        try:
            return StitchResult.from_c(rc)
        finally:
            self.ctx.lib.cl_stitch_result_free(C.byref(rc))

    def stats(self):
        st = PlanStats()
        self.ctx._check(self.ctx.lib.cl_stitch_plan_stats(self.handle, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in PlanStats._fields_}

    def launches(self):
        """per-kernel-launch info of the last execute (call after sync)"""
        out = []
        for i in range(self.ctx.lib.cl_stitch_plan_launch_count(self.handle)):
            li = LaunchInfo()
            self.ctx._check(self.ctx.lib.cl_stitch_plan_launch_info(self.ctx.handle, self.handle, i, C.byref(li)))
            out.append(dict(kernel=li.kernel.decode(), n_problems=int(li.n_problems), dp_cells=int(li.dp_cells),
                            dp_bytes=int(li.dp_bytes), ms=float(li.last_ms), in_pass_ms=float(li.in_pass_ms), lds_bytes=int(li.lds_bytes), max_sweep=int(li.max_sweep),
                            longest=(int(li.max_n1), int(li.max_n2)), event_ms=float(li.event_ms)))
        return out

    def destroy(self):
        if self.handle:
            self.ctx.lib.cl_stitch_plan_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Context:
    """cl_context: one HIP device + stream.  Raises ClError(CL_ERR_NO_DEVICE) without a GPU."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.device = int(device)
        self.handle = self.lib.cl_context_create(device)
        if not self.handle:
            msg = self.lib.cl_last_error(None)
            raise ClError(-3, msg.decode() if msg else "")

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.cl_last_error(self.handle)
            raise ClError(rc, msg.decode() if msg else "")

    def device_name(self):
        return self.lib.cl_device_name(self.handle).decode()

    def po_poa_batch(self, batch, num_pw, params):
        num_pw = np.ascontiguousarray(num_pw, dtype=np.uint8)
        assert len(num_pw) == batch.n_problems
        bc, rc = batch.as_c(), StitchResultC()
        self._check(self.lib.cl_po_poa_batch(self.handle, C.byref(bc), num_pw.ctypes.data, C.byref(params), C.byref(rc)))
        try:
            return StitchResult.from_c(rc)
        finally:
            self.lib.cl_stitch_result_free(C.byref(rc))

    def stitch_batch_align(self, batch, params=None):
        params = params or default_stitch_params()
        bc, rc = batch.as_c(), StitchResultC()
        self._check(self.lib.cl_stitch_batch_align(self.handle, C.byref(bc), C.byref(params), C.byref(rc)))
        try:
            return StitchResult.from_c(rc)
        finally:
            self.lib.cl_stitch_result_free(C.byref(rc))

    def set_stitch_hook(self, fn, min_cells=0):
        """cl_context_set_stitch_hook: from now on cl_stitch (hence merge / core_align) on this context hands every extracted batch of at least min_cells DP cells to
        fn(batch: StitchBatch, params: StitchParams) -> StitchResult for the WHOLE batch in its order, instead of aligning it alone (several devices share a merge's
        subproblems: centrolign_amd.msa).  fn = None removes the hook."""
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(StitchBatchC), C.POINTER(StitchParams), C.POINTER(StitchResultC))
        self.lib.cl_context_set_stitch_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        self.lib.cl_stitch_result_alloc.argtypes = [C.POINTER(StitchResultC), C.c_uint64, C.c_uint64]
        if fn is None:
            self._check(self.lib.cl_context_set_stitch_hook(self.handle, None, None, 0))
            self._stitch_hook = None
            return

        def tramp(_user, _ctx, bc, pc, out):
            try:
                res = fn(StitchBatch.from_c(bc.contents), pc.contents)
                n, tot = res.n_problems, int(res.aln_off[-1])
                if self.lib.cl_stitch_result_alloc(out, n, tot) != 0:
                    return -5
                o = out.contents
                C.memmove(o.aln_off, np.ascontiguousarray(res.aln_off, np.uint64).ctypes.data, 8 * (n + 1))
                if tot:
                    C.memmove(o.pairs, np.ascontiguousarray(res.pairs, np.uint64).ctypes.data, 16 * tot)
                if n:
                    C.memmove(o.score, np.ascontiguousarray(res.score, np.int64).ctypes.data, 8 * n)
                    C.memmove(o.route, np.ascontiguousarray(res.route, np.uint8).ctypes.data, n)
                    C.memmove(o.num_pw, np.ascontiguousarray(res.num_pw, np.uint8).ctypes.data, n)
                return 0
            except Exception as e:   # noqa: BLE001 (an exception must not cross the C frames)
                self._stitch_hook_error = e
                return -1
        cb = HOOK(tramp)
        self._stitch_hook = cb          # (kept alive as long as the library may call it)
        self._check(self.lib.cl_context_set_stitch_hook(self.handle, C.cast(cb, C.c_void_p), None, int(min_cells)))

    def stitch(self, graph1, graph2, segments, params=None):
        """Stitcher::stitch (include/centrolign/stitcher.hpp:104-206): stitched Alignment as an (n, 2) uint64 array"""
        params = params or default_stitch_params()
        g1, g2, sg, out = graph1.as_c(), graph2.as_c(), segments.as_c(), AlignmentC()
        self._check(self.lib.cl_stitch(self.handle, C.byref(g1), C.byref(g2), C.byref(sg), C.byref(params), C.byref(out)))
        try:
            n = int(out.n_pairs)
            return np.ctypeslib.as_array(out.pairs, shape=(max(n, 1) * 2,))[:2 * n].copy().reshape(n, 2)
        finally:
            self.lib.cl_alignment_free(C.byref(out))

    def chain_sparse_affine(self, graph1, graph2, matches, scale=1.0, params=None, num_match_sets=None, want_dp=False,
                            sparse=False):
        """sparse_affine_chain_dp (include/centrolign/anchorer.hpp:1812-2471), or with sparse=True sparse_chain_dp
        (:1511-1750), on the GPU.
        Returns dict(chain=(n,3) uint32 [match_set, idx1, idx2], dp=float32[n_pairs] or None, n_ties, device_ms)"""
        params = params or default_chain_params()
        g1, g2, mc, out = graph1.as_c(), graph2.as_c(), matches.as_c(), ChainResultC()
        n = matches.n_sets if num_match_sets is None else num_match_sets
        if sparse:
            self._check(self.lib.cl_chain_sparse(self.handle, C.byref(g1), C.byref(g2), C.byref(mc), n, C.byref(params),
                                                 int(want_dp), C.byref(out)))
        else:
            self._check(self.lib.cl_chain_sparse_affine(self.handle, C.byref(g1), C.byref(g2), C.byref(mc), n, C.byref(params),
                                                        float(scale), int(want_dp), C.byref(out)))
        try:
            na, npairs = int(out.n_anchors), int(out.n_pairs)
            chain = np.ctypeslib.as_array(out.anchors, shape=(max(na, 1) * 3,))[:3 * na].copy().reshape(na, 3)
            dp = np.ctypeslib.as_array(out.dp, shape=(max(npairs, 1),))[:npairs].copy() if want_dp and npairs else None
            return dict(chain=chain, dp=dp, n_ties=int(out.n_ties), device_ms=float(out.device_ms), n_pairs=npairs,
                        prep_ms=float(out.prep_ms), index_ms=float(out.index_ms), traceback_ms=float(out.traceback_ms),
                        end_gaps=(int(out.gap_before_first), int(out.gap_after_last)),
                        end_gap_scores=(float(out.gap_score_before_first), float(out.gap_score_after_last)))
        finally:
            self.lib.cl_chain_result_free(C.byref(out))

    def find_matches(self, graph1, graph2, max_count=3000, use_color_set_size=True, params=None, want_stats=False):
        """PathMatchFinder::find_matches (include/centrolign/match_finder.hpp:120-212): suffix array + LCP on the device, the
        minimal-rare-match query on the host.  Returns MatchSets (and the cl_match_stats dict)."""
        g1, g2, mp, st = graph1.as_c(), graph2.as_c(), match_params(max_count, use_color_set_size, params), MatchStats()
        h = C.c_void_p()
        self._check(self.lib.cl_find_matches(self.handle, C.byref(g1), C.byref(g2), C.byref(mp), C.byref(h), C.byref(st)))
        ms = _take_owned_match_sets(self.lib, h)
        return (ms, st.as_dict()) if want_stats else ms

    def suffix_array_lcp(self, text):
        """device suffix array / LCP array / inverse suffix array of a uint8 text ending in a unique smallest character;
        returns (sa, lcp, isa, rounds)"""
        text = np.ascontiguousarray(text, np.uint8)
        n = len(text)
        sa, lcp, isa = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        rounds = C.c_uint32(0)
        self._check(self.lib.cl_suffix_array_lcp(self.handle, text.ctypes.data, n, sa.ctypes.data, lcp.ctypes.data, isa.ctypes.data, C.byref(rounds)))
        return sa, lcp, isa, int(rounds.value)

    def anchor_chain(self, graph1, graph2, matches, max_num_match_pairs=1250000, score_scale=1.0, autocalibrate=True,
                     params=None, fill_in=True, chaining_algorithm=None):
        """Anchorer::anchor_chain (include/centrolign/anchorer.hpp:958-996) without branch splitting.
        Returns dict(chain (n,3) [position in the reordered sets, idx1, idx2], gap_before/after,
        gap_score_before/after, score, count1, count2, full_length, walk_off, walk1, walk2, set_order, scale, n_ties)"""
        ap = AnchorParams()
        ap.chain = params or default_chain_params()
        ap.max_num_match_pairs = int(max_num_match_pairs)
        ap.score_scale = float(score_scale)
        ap.autocalibrate_gap_penalties = int(autocalibrate)
        ap.do_fill_in_anchoring = int(fill_in)
        ap.chaining_algorithm_plus_one = 0 if chaining_algorithm is None else int(chaining_algorithm) + 1   # the CLI's -g: 1 Sparse (ChainMerge tables), 2 SparseAffine
        g1, g2, mc, out = graph1.as_c(), graph2.as_c(), matches.as_c(), AnchorChainResultC()
        self._check(self.lib.cl_anchor_chain(self.handle, C.byref(g1), C.byref(g2), C.byref(mc), C.byref(ap), C.byref(out)))
        return self._anchor_chain_dict(out)

    def _anchor_chain_dict(self, out):
        try:
            na, ns = int(out.n_anchors), int(out.n_sets)

            def arr(ptr, n, width=1):
                return np.ctypeslib.as_array(ptr, shape=(max(n, 1) * width,))[:n * width].copy()
            walk_off = np.ctypeslib.as_array(out.walk_off, shape=(na + 1,)).copy()
            nw = int(walk_off[-1])
            return dict(chain=arr(out.anchors, na, 3).reshape(na, 3), gap_before=arr(out.gap_before, na),
                        gap_after=arr(out.gap_after, na), gap_score_before=arr(out.gap_score_before, na),
                        gap_score_after=arr(out.gap_score_after, na), score=arr(out.score, na),
                        count1=arr(out.count1, na), count2=arr(out.count2, na), full_length=arr(out.full_length, na),
                        walk_off=walk_off, walk1=arr(out.walk1, nw), walk2=arr(out.walk2, nw),
                        set_order=arr(out.set_order, ns), scale=float(out.scale), n_ties=int(out.n_ties),
                        fill_in_pairs=int(out.fill_in_pairs), fill_in_device_ms=float(out.fill_in_device_ms))
        finally:
            self.lib.cl_anchor_chain_result_free(C.byref(out))

    def anchor_chain_masked(self, graph1, graph2, matches, mask, override_scale=None, max_num_match_pairs=1250000, score_scale=1.0,
                            autocalibrate=True, params=None, fill_in=True):
        """Anchorer::anchor_chain with masked matches and an overriding scale (include/centrolign/anchorer.hpp:135-145; the cyclisation
        rounds, src/core.cpp:221-227).  mask: (n, 3) [set, idx1, idx2] in the indexing of `matches`.  Same dict as anchor_chain."""
        ap = AnchorParams()
        ap.chain = params or default_chain_params()
        ap.max_num_match_pairs = int(max_num_match_pairs)
        ap.score_scale = float(score_scale)
        ap.autocalibrate_gap_penalties = int(autocalibrate)
        ap.do_fill_in_anchoring = int(fill_in)
        m = np.ascontiguousarray(mask, np.uint64).reshape(-1, 3)
        sc = C.c_double(float(override_scale)) if override_scale is not None else None
        g1, g2, mc, out = graph1.as_c(), graph2.as_c(), matches.as_c(), AnchorChainResultC()
        self._check(self.lib.cl_anchor_chain_masked(self.handle, C.byref(g1), C.byref(g2), C.byref(mc), C.byref(ap), m.ctypes.data, len(m),
                                                    C.byref(sc) if sc is not None else None, C.byref(out)))
        return self._anchor_chain_dict(out)

    def internal_stitch(self, graph, walk_off, walk1, walk2, params=None):
        """Stitcher::internal_stitch (include/centrolign/stitcher.hpp:209-234): a chain of anchors inside ONE graph -> (n, 2) uint64"""
        params = params or default_stitch_params()
        wo, w1, w2 = np.ascontiguousarray(walk_off, np.uint64), np.ascontiguousarray(walk1, np.uint32), np.ascontiguousarray(walk2, np.uint32)
        g, out = graph.as_c(), AlignmentC()
        self._check(self.lib.cl_internal_stitch(self.handle, C.byref(g), len(wo) - 1 if len(wo) else 0, wo.ctypes.data, w1.ctypes.data, w2.ctypes.data,
                                                C.byref(params), C.byref(out)))
        try:
            n = int(out.n_pairs)
            return np.ctypeslib.as_array(out.pairs, shape=(max(n, 1) * 2,))[:2 * n].copy().reshape(n, 2)
        finally:
            self.lib.cl_alignment_free(C.byref(out))

    def core_align(self, graph1, graph2, matches, score_scale=1.0, max_num_match_pairs=1250000, score_boundaries=False, tweak=None):
        """Core::align (include/centrolign/core.hpp:181-252) with the CLI's default configuration: anchor chain (branch
        splitting, chaining, fill-in), partition, despecify, stitch.  `tweak(params)` may edit the CoreAlignParams.
        Returns dict(alignment (n,2) uint64, seg_off, walk_off, walk1, walk2, scale, n_chain_anchors, *_ms)"""
        ap = CoreAlignParams()
        self.lib.cl_core_align_params_default(C.byref(ap))
        ap.anchor.score_scale = float(score_scale)
        ap.anchor.max_num_match_pairs = int(max_num_match_pairs)
        ap.partition.score_boundaries = int(score_boundaries)
        if tweak:
            tweak(ap)
        g1, g2, mc, out = graph1.as_c(), graph2.as_c(), matches.as_c(), CoreAlignResultC()
        self._check(self.lib.cl_core_align(self.handle, C.byref(g1), C.byref(g2), C.byref(mc), C.byref(ap), C.byref(out)))
        try:
            return _core_align_dict(out)
        finally:
            self.lib.cl_core_align_result_free(C.byref(out))

    def leaf_intrinsic_scale(self, leaf, max_count=3000, max_num_match_pairs=1250000, params=None, fill_in=True):
        """the per-leaf step of Core's calibration (src/core.cpp:122-166): self matches, main-diagonal subset,
        Anchorer::estimate_score_scale.  ScoreFunction::score_scale = the mean over the leaves."""
        ap = AnchorParams()
        ap.chain = params or default_chain_params()
        ap.max_num_match_pairs = int(max_num_match_pairs)
        ap.score_scale = 1.0
        ap.autocalibrate_gap_penalties = 1
        ap.do_fill_in_anchoring = int(fill_in)
        g, mp, scale = leaf.as_c(), match_params(max_count, True, params), C.c_double(0)
        self._check(self.lib.cl_leaf_intrinsic_scale(self.handle, C.byref(g), C.byref(mp), C.byref(ap), C.byref(scale)))
        return float(scale.value)

    def leaf_calibrate(self, leaf, max_count=3000, max_num_match_pairs=1250000, params=None, fill_in=True):
        """cl_leaf_calibrate: the leaf's intrinsic scale plus an opaque handle on what the tandem-duplication rounds read (the self matches,
        the main-diagonal chain); returns (scale, handle) — release with free_leaf_calibration"""
        ap = AnchorParams()
        ap.chain = params or default_chain_params()
        ap.max_num_match_pairs = int(max_num_match_pairs)
        ap.score_scale = 1.0
        ap.autocalibrate_gap_penalties = 1
        ap.do_fill_in_anchoring = int(fill_in)
        g, mp, scale, h = leaf.as_c(), match_params(max_count, True, params), C.c_double(0), C.c_void_p()
        self.lib.cl_leaf_calibrate.restype = C.c_int
        self.lib.cl_leaf_calibrate.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.c_void_p, C.POINTER(AnchorParams), C.POINTER(C.c_double), C.POINTER(C.c_void_p)]
        self._check(self.lib.cl_leaf_calibrate(self.handle, C.byref(g), C.byref(mp), C.byref(ap), C.byref(scale), C.byref(h)))
        return float(scale.value), h

    def free_leaf_calibration(self, h):
        self.lib.cl_leaf_calibration_free.argtypes = [C.c_void_p]
        self.lib.cl_leaf_calibration_free(h)

    def leaf_bond_alignments(self, leaf, memo, score_scale, max_num_match_pairs=1250000, params=None, fill_in=True, stitch_params=None,
                             bonds=None, max_rounds=3):
        """cl_leaf_bond_alignments: the tandem-duplication rounds of one leaf (src/core.cpp:199-296); list of (n, 2) uint64 alignments in
        path positions"""
        ap = AnchorParams()
        ap.chain = params or default_chain_params()
        ap.max_num_match_pairs = int(max_num_match_pairs)
        ap.score_scale = float(score_scale)
        ap.autocalibrate_gap_penalties = 1
        ap.do_fill_in_anchoring = int(fill_in)
        sp = stitch_params or default_stitch_params()
        bp = bonds or bond_params()
        g, out = leaf.as_c(), AlignmentListC()
        self.lib.cl_leaf_bond_alignments.restype = C.c_int
        self.lib.cl_leaf_bond_alignments.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.c_void_p, C.POINTER(AnchorParams), C.POINTER(StitchParams),
                                                     C.POINTER(BondParams), C.c_uint64, C.POINTER(AlignmentListC)]
        self._check(self.lib.cl_leaf_bond_alignments(self.handle, C.byref(g), memo, C.byref(ap), C.byref(sp), C.byref(bp), int(max_rounds), C.byref(out)))
        try:
            res = []
            for i in range(int(out.n)):
                n = int(out.alignments[i].n_pairs)
                res.append(np.ctypeslib.as_array(out.alignments[i].pairs, shape=(max(n, 1) * 2,))[:2 * n].copy().reshape(n, 2))
            return res
        finally:
            self.lib.cl_alignment_list_free(C.byref(out))

    def polish_cyclized_graph(self, graph, path_names, sequence_names, score_scale, newick=None, max_num_match_pairs=1250000, max_count=3000,
                              polish=None, tweak=None):
        """Core::polish_cyclized_graph (src/core.cpp:650-767): realigns the regions cl_identify_inconsistencies reports; returns
        (polished BaseGraph, number of regions)"""
        mp = MergeParams()
        self.lib.cl_merge_params_default(C.byref(mp))
        mp.match.max_count = int(max_count)
        mp.align.anchor.score_scale = float(score_scale)
        mp.align.anchor.max_num_match_pairs = int(max_num_match_pairs)
        if tweak:
            tweak(mp)
        pp = polish or polish_params()
        g, h, n = graph.as_c(), C.c_void_p(), C.c_uint64(0)
        pn = (C.c_char_p * max(len(path_names), 1))(*[x.encode() for x in path_names])
        sn = (C.c_char_p * max(len(sequence_names), 1))(*[x.encode() for x in sequence_names])
        self.lib.cl_polish_cyclized_graph.restype = C.c_int
        self.lib.cl_polish_cyclized_graph.argtypes = [C.c_void_p, C.POINTER(BaseGraphC), C.POINTER(C.c_char_p), C.c_char_p, C.POINTER(C.c_char_p), C.c_uint64,
                                                      C.POINTER(MergeParams), C.POINTER(PolishParams), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        self._check(self.lib.cl_polish_cyclized_graph(self.handle, C.byref(g), pn, newick.encode() if newick else None, sn, len(sequence_names),
                                                      C.byref(mp), C.byref(pp), C.byref(h), C.byref(n)))
        return _take_owned_base_graph(self.lib, h), int(n.value)

    def merge(self, graph1, graph2, score_scale=1.0, max_num_match_pairs=1250000, max_count=3000, tweak=None):
        """one merge of the progressive MSA (the loop body of Core::do_execution, include/centrolign/core.hpp:268-392):
        reassign_sentinels, find_matches, Core::align, fuse.  Returns dict(alignment (n,2), fused BaseGraph, n_match_sets, *_ms)"""
        mp = MergeParams()
        self.lib.cl_merge_params_default(C.byref(mp))
        mp.match.max_count = int(max_count)
        mp.align.anchor.score_scale = float(score_scale)
        mp.align.anchor.max_num_match_pairs = int(max_num_match_pairs)
        if tweak:
            tweak(mp)
        g1, g2, out = graph1.as_c(), graph2.as_c(), MergeResultC()
        self._check(self.lib.cl_merge(self.handle, C.byref(g1), C.byref(g2), C.byref(mp), C.byref(out)))
        try:
            al = _core_align_dict(out.align)
            fused, out.fused = _take_owned_base_graph(self.lib, out.fused), None
            return dict(alignment=al["alignment"], fused=fused, n_match_sets=int(out.n_match_sets), match_ms=float(out.match_ms),
                        align_ms=float(out.align_ms), fuse_ms=float(out.fuse_ms), align=al)
        finally:
            self.lib.cl_merge_result_free(C.byref(out))

    def msa(self, fasta_text, newick=None, max_num_match_pairs=1250000, max_count=3000, skip_calibration=False, subproblems_prefix=None,
            restart=False, induced_pairwise_prefix=None, workers=1, cyclize=False, min_cyclizing_length=None, devices=None, chaining_algorithm=None):
        """the whole CLI flow in the library (cl_msa): FASTA text (+ Newick text) -> explicit CIGAR (two sequences) or GFA; returns
        (text bytes, stats dict).  cyclize = the CLI's -c, min_cyclizing_length its -y; devices = device ordinals the worker contexts are
        spread over (worker w on devices[w % len(devices)]; one process, several GPUs)"""
        raw = fasta_text.encode() if isinstance(fasta_text, str) else bytes(fasta_text)
        mp = MsaParams()
        self.lib.cl_msa_params_default(C.byref(mp))
        mp.merge.match.max_count = int(max_count)
        mp.merge.align.anchor.max_num_match_pairs = int(max_num_match_pairs)
        mp.skip_calibration = int(skip_calibration)
        if chaining_algorithm is not None:
            mp.merge.align.anchor.chaining_algorithm_plus_one = int(chaining_algorithm) + 1   # the CLI's -g: 1 Sparse over ChainMerge, 2 SparseAffine
        mp.n_workers = int(workers)
        dev_arr = None
        if devices:
            dev_arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            mp.devices, mp.n_devices = C.cast(dev_arr, C.POINTER(C.c_int)), len(devices)
        mp.subproblems_prefix = subproblems_prefix.encode() if subproblems_prefix else None   # -S
        mp.restart = int(restart)                                                             # -R
        mp.induced_pairwise_prefix = induced_pairwise_prefix.encode() if induced_pairwise_prefix else None   # -A
        mp.cyclize = int(cyclize)                                                             # -c
        if min_cyclizing_length is not None:
            mp.bonds.min_length = float(min_cyclizing_length)                                 # -y
        p, n, st = C.c_void_p(), C.c_uint64(0), MsaStats()
        self._check(self.lib.cl_msa(self.handle, raw, len(raw), None if not newick else newick.encode(), C.byref(mp), C.byref(p), C.byref(n), C.byref(st)))
        try:
            return C.string_at(p, int(n.value)), {k: getattr(st, k) for k, _ in MsaStats._fields_}
        finally:
            _libc_free(p)

    def plan(self, batch, params=None, force_num_pw=None):
        params = params or default_stitch_params()
        bc = batch.as_c()
        h = C.c_void_p()
        f = None
        if force_num_pw is not None:
            f = np.ascontiguousarray(force_num_pw, dtype=np.uint8)
        self._check(self.lib.cl_stitch_plan_create(self.handle, C.byref(bc), C.byref(params),
                                                   None if f is None else f.ctypes.data, C.byref(h)))
        return Plan(self, h, batch)

    # ---- merge groups: one merge over several GPUs (cl_peer_api.cpp) ----
    def peer_export(self):
        """cl_context_peer_export: this context's handle (bytes) for the other members of a merge group"""
        h = (C.c_ubyte * 128)()
        self.lib.cl_context_peer_export.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.lib.cl_context_peer_export(self.handle, h))
        return bytes(h)

    def peer_group(self, members, my_index, epoch_base):
        """cl_context_peer_group: members = every member's peer_export() in the group's order (own entry included); [] leaves the group"""
        self.lib.cl_context_peer_group.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        if not members or len(members) <= 1:
            self._check(self.lib.cl_context_peer_group(self.handle, 0, 0, None, 0))
            return
        buf = (C.c_ubyte * (128 * len(members))).from_buffer_copy(b"".join(members))
        self._check(self.lib.cl_context_peer_group(self.handle, len(members), int(my_index), buf, int(epoch_base)))

    def peer_selftest(self, token, timeout_ms=5000):
        """cl_context_peer_selftest: True when stores, arrival words and waits made it once round the current group"""
        self.lib.cl_context_peer_selftest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        return self.lib.cl_context_peer_selftest(self.handle, int(token), int(timeout_ms)) == 0

    def peer_stats(self):
        st = (C.c_uint64 * 6)()
        self.lib.cl_context_peer_stats.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.lib.cl_context_peer_stats(self.handle, st))
        return dict(shared_dps=int(st[0]), shared_far_launches=int(st[1]), merged_blocks=int(st[2]), epoch_mark=int(st[3]), selftest_mark=int(st[4]), steals=int(st[5]))

    def peer_steal(self, job):
        """cl_context_peer_steal: the next chunk number of job `job` from the group's one atomic counter (member 0's exported memory); local count without a group"""
        out = C.c_uint32(0)
        self.lib.cl_context_peer_steal.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        self._check(self.lib.cl_context_peer_steal(self.handle, int(job), C.byref(out)))
        return int(out.value)

    def memory_stats(self, reset_peak=False):
        """cl_context_memory: device bytes this context holds / held at most / keeps cached, hipMemGetInfo's free and total, page-locked host bytes"""
        st = (C.c_uint64 * 6)()
        self.lib.cl_context_memory.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self._check(self.lib.cl_context_memory(self.handle, st, int(bool(reset_peak))))
        return dict(zip(("live_bytes", "peak_bytes", "cached_bytes", "device_free_bytes", "device_total_bytes", "pinned_host_bytes"), [int(x) for x in st]))

    def fallback_counters(self, reset=False):
        """cl_fallback_counters (process-wide): second attempts the device path made by itself — strips re-run anti-diagonal-wise, walks re-run per block"""
        return fallback_counters(reset)

    def close(self):
        if self.handle:
            self.lib.cl_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
