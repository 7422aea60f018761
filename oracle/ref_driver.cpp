/*
 * ref_driver.cpp — TEST INFRASTRUCTURE ONLY.  Our own glue code that links against the UNMODIFIED
 * reference (compiled from /root/reference by oracle/Makefile into oracle/_ref/) and exposes it through a
 * small C ABI so that tests, the golden-vector generator and bench.py's cpu_baseline leg can call the real
 * thing.  It contains no reference source; it reaches the reference's protected members the same way the
 * reference's own tests do (derive + using-declarations, src/test/test_stitcher.cpp:23-28).
 *
 *   ref_po_poa        -> centrolign::po_poa<NumPW>              (include/centrolign/alignment.hpp:78-85)
 *   ref_stitch_batch  -> centrolign::Stitcher::subalign per problem (src/stitcher.cpp:24-78)
 *   ref_msa_dump      -> the CLI pipeline (src/main.cpp:239-301, include/centrolign/core.hpp:182-403) with the
 *                        stitch subproblems of every merge written out as flat arrays
 */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <fstream>
#include <sstream>
#include <string>
#include <map>
#include <vector>

#include "centrolign/alignment.hpp"
#include "centrolign/core.hpp"
#include "centrolign/execution.hpp"
#include "centrolign/tree.hpp"
#include "centrolign/fuse.hpp"
#include "centrolign/gfa.hpp"
#include "centrolign/induced_match_finder.hpp"
#include "centrolign/inconsistency_identifier.hpp"
#include "centrolign/modify_graph.hpp"
#include "centrolign/parameters.hpp"
#include "centrolign/stitcher.hpp"
#include "centrolign/utility.hpp"

#include "cl_oracle.h"

using namespace centrolign;

namespace {

/* Rebuild a BaseGraph whose previous() AND next() lists have exactly the given orders.  Both orders come
 * from one edge-insertion sequence (BaseGraph::add_edge appends to both lists, src/graph.cpp), so a greedy
 * merge that emits an edge when it heads both of its lists reproduces a valid sequence. */
BaseGraph build_graph(const clo_graph* g) {
    BaseGraph bg;
    for (uint64_t v = 0; v < g->n; ++v) bg.add_node((char)g->label[v]);
    if (!g->next_off) {
        for (uint64_t v = 0; v < g->n; ++v)
            for (uint64_t e = g->prev_off[v]; e < g->prev_off[v + 1]; ++e) bg.add_edge(g->prev_idx[e], v);
        return bg;
    }
    std::vector<uint64_t> np(g->n), pp(g->n);
    for (uint64_t v = 0; v < g->n; ++v) { np[v] = g->next_off[v]; pp[v] = g->prev_off[v]; }
    std::vector<uint64_t> work;
    for (uint64_t v = g->n; v-- > 0;) work.push_back(v);
    auto ready = [&](uint64_t u) -> bool {
        if (np[u] >= g->next_off[u + 1]) return false;
        uint64_t v = g->next_idx[np[u]];
        return pp[v] < g->prev_off[v + 1] && g->prev_idx[pp[v]] == u;
    };
    uint64_t emitted = 0, total = g->n ? g->prev_off[g->n] - g->prev_off[0] : 0;
    while (!work.empty()) {
        uint64_t u = work.back();
        work.pop_back();
        while (ready(u)) {
            uint64_t v = g->next_idx[np[u]];
            bg.add_edge(u, v);
            ++emitted; ++np[u]; ++pp[v];
            if (pp[v] < g->prev_off[v + 1]) work.push_back(g->prev_idx[pp[v]]);
        }
    }
    if (emitted != total) {
        /* inconsistent orders: honour previous() (the only order results depend on) */
        BaseGraph bg2;
        for (uint64_t v = 0; v < g->n; ++v) bg2.add_node((char)g->label[v]);
        for (uint64_t v = 0; v < g->n; ++v)
            for (uint64_t e = g->prev_off[v]; e < g->prev_off[v + 1]; ++e) bg2.add_edge(g->prev_idx[e], v);
        return bg2;
    }
    return bg;
}

std::vector<uint64_t> to_vec(const uint32_t* p, uint64_t n) { return std::vector<uint64_t>(p, p + n); }

template <int NumPW>
AlignmentParameters<NumPW> make_params(const cl_align_params* p) {
    AlignmentParameters<NumPW> ap;
    ap.match = p->match;
    ap.mismatch = p->mismatch;
    for (int i = 0; i < NumPW; ++i) { ap.gap_open[i] = p->gap_open[i]; ap.gap_extend[i] = p->gap_extend[i]; }
    return ap;
}

struct DumpStitcher : public Stitcher {
    using Stitcher::subalign;
    using Stitcher::do_alignment;
    using Extractor::extract_graphs_between;
    using Extractor::source_sink_minmax;
};

void set_stitcher(DumpStitcher& st, const cl_stitch_params* sp) {
    st.alignment_params = make_params<3>(&sp->alignment_params);
    st.max_trivial_size = sp->max_trivial_size;
    st.min_wfa_size = sp->min_wfa_size;
    st.max_wfa_size = sp->max_wfa_size;
    st.max_wfa_ratio = sp->max_wfa_ratio;
    st.wfa_pruning_dist = sp->wfa_pruning_dist;
    st.deletion_alignment_ratio = sp->deletion_alignment_ratio;
    st.deletion_alignment_short_max_size = sp->deletion_alignment_short_max_size;
    st.deletion_alignment_long_min_size = sp->deletion_alignment_long_min_size;
}

void side_graph(const cl_graph_side* s, uint64_t k, clo_graph* g) {
    uint64_t b = s->node_off[k];
    g->n = s->node_off[k + 1] - b;
    g->label = s->label + b;
    g->prev_off = s->prev_off + b;
    g->prev_idx = s->prev_idx;
    g->next_off = s->next_off ? s->next_off + b : nullptr;
    g->next_idx = s->next_idx;
    g->n_src = s->src_off[k + 1] - s->src_off[k];
    g->src = s->src_idx + s->src_off[k];
    g->n_snk = s->snk_off[k + 1] - s->snk_off[k];
    g->snk = s->snk_idx + s->snk_off[k];
}

SubGraphInfo make_info(const cl_graph_side* s, uint64_t k) {
    clo_graph g;
    side_graph(s, k, &g);
    SubGraphInfo info;
    info.subgraph = build_graph(&g);
    info.sources = to_vec(g.src, g.n_src);
    info.sinks = to_vec(g.snk, g.n_snk);
    info.back_translation.resize(g.n);
    for (uint64_t v = 0; v < g.n; ++v)
        info.back_translation[v] = s->back_translation ? s->back_translation[s->node_off[k] + v] : v;
    return info;
}

/* ---- flat dump writer ------------------------------------------------------------------------------ */
struct Dump {
    FILE* f = nullptr;
    bool open(const std::string& path) {
        f = fopen(path.c_str(), "wb");
        if (!f) return false;
        fwrite("CLDUMP1\n", 1, 8, f);
        return true;
    }
    void put(const std::string& name, uint8_t dtype, const void* data, uint64_t count, size_t esz) {
        uint32_t nl = (uint32_t)name.size();
        fwrite(&nl, 4, 1, f);
        fwrite(name.data(), 1, nl, f);
        fwrite(&dtype, 1, 1, f);
        fwrite(&count, 8, 1, f);
        if (count) fwrite(data, esz, count, f);
    }
    void u8(const std::string& n, const std::vector<uint8_t>& v) { put(n, 0, v.data(), v.size(), 1); }
    void u32(const std::string& n, const std::vector<uint32_t>& v) { put(n, 1, v.data(), v.size(), 4); }
    void u64(const std::string& n, const std::vector<uint64_t>& v) { put(n, 2, v.data(), v.size(), 8); }
    void i64(const std::string& n, const std::vector<int64_t>& v) { put(n, 3, v.data(), v.size(), 8); }
    void f64(const std::string& n, const std::vector<double>& v) { put(n, 4, v.data(), v.size(), 8); }
    void str(const std::string& n, const std::string& s) { put(n, 0, s.data(), s.size(), 1); }
    void close() { if (f) fclose(f); f = nullptr; }
};

struct FlatSide {
    std::vector<uint64_t> node_off{0}, prev_off{0}, next_off{0}, src_off{0}, snk_off{0}, back;
    std::vector<uint8_t> label;
    std::vector<uint32_t> prev_idx, next_idx, src_idx, snk_idx;
    void add(const SubGraphInfo& info) {
        const auto& g = info.subgraph;
        for (uint64_t v = 0; v < g.node_size(); ++v) {
            label.push_back((uint8_t)g.label(v));
            for (auto p : g.previous(v)) prev_idx.push_back((uint32_t)p);
            prev_off.push_back(prev_idx.size());
            for (auto q : g.next(v)) next_idx.push_back((uint32_t)q);
            next_off.push_back(next_idx.size());
            back.push_back(info.back_translation[v]);
        }
        node_off.push_back(label.size());
        for (auto s : info.sources) src_idx.push_back((uint32_t)s);
        src_off.push_back(src_idx.size());
        for (auto s : info.sinks) snk_idx.push_back((uint32_t)s);
        snk_off.push_back(snk_idx.size());
    }
    void write(Dump& d, const std::string& pre) {
        d.u64(pre + "node_off", node_off); d.u8(pre + "label", label);
        d.u64(pre + "prev_off", prev_off); d.u32(pre + "prev_idx", prev_idx);
        d.u64(pre + "next_off", next_off); d.u32(pre + "next_idx", next_idx);
        d.u64(pre + "src_off", src_off); d.u32(pre + "src_idx", src_idx);
        d.u64(pre + "snk_off", snk_off); d.u32(pre + "snk_idx", snk_idx);
        d.u64(pre + "back_translation", back);
    }
};

/* parent graph of a merge: labels, adjacency in BaseGraph order, embedded paths, sentinels */
void dump_base_graph(Dump& d, const std::string& pre, const BaseGraph& g, const SentinelTableau& t) {
    std::vector<uint8_t> label;
    std::vector<uint64_t> next_off{0}, prev_off{0}, path_off{0};
    std::vector<uint32_t> next_idx, prev_idx, path_nodes;
    for (uint64_t v = 0; v < g.node_size(); ++v) {
        label.push_back((uint8_t)g.label(v));
        for (auto q : g.next(v)) next_idx.push_back((uint32_t)q);
        next_off.push_back(next_idx.size());
        for (auto p : g.previous(v)) prev_idx.push_back((uint32_t)p);
        prev_off.push_back(prev_idx.size());
    }
    for (uint64_t p = 0; p < g.path_size(); ++p) {
        for (auto v : g.path(p)) path_nodes.push_back((uint32_t)v);
        path_off.push_back(path_nodes.size());
    }
    d.u8(pre + "label", label);
    d.u64(pre + "next_off", next_off); d.u32(pre + "next_idx", next_idx);
    d.u64(pre + "prev_off", prev_off); d.u32(pre + "prev_idx", prev_idx);
    d.u64(pre + "path_off", path_off); d.u32(pre + "path_nodes", path_nodes);
    d.u64(pre + "tableau", std::vector<uint64_t>{t.src_id, t.snk_id});
}

/* The CLI pipeline with the stitch loop opened up.  Follows Core::do_execution (core.hpp:256-403) and
 * Core::align (core.hpp:182-254) call for call; only Stitcher::stitch (stitcher.hpp:104-206) is unrolled here
 * so that each subalign input/output can be recorded. */
struct DumpCore : public Core {
    DumpCore(std::vector<std::pair<std::string, std::string>>&& seqs, Tree&& tree) : Core(std::move(seqs), std::move(tree)) {}

    double t_match = 0, t_chain = 0, t_partition = 0, t_extract = 0, t_subalign = 0, t_fuse = 0, t_calib = 0;

    template <class XMerge>
    Alignment align_dump(std::vector<match_set_t>& matches, const Subproblem& sp1, const Subproblem& sp2,
                         XMerge& x1, XMerge& x2, Dump* dump, const std::string& pre) {
        using clk = std::chrono::steady_clock;
        auto t0 = clk::now();
        bool restrain_memory = (sp1.graph.path_size() * sp2.graph.path_size() * anchorer.max_num_match_pairs *
                                    log2(anchorer.max_num_match_pairs) > memory_restraint_size);
        auto anchors = anchorer.anchor_chain(matches, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, restrain_memory);
        auto t1 = clk::now();
        auto segments = partitioner.partition_anchors(anchors, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, false);
        for (auto& seg : segments) stitcher.despecify_indel_breakpoints(seg);
        auto t2 = clk::now();

        DumpStitcher st;
        static_cast<Stitcher&>(st) = stitcher;
        std::vector<std::vector<std::pair<SubGraphInfo, SubGraphInfo>>> within;
        std::vector<std::pair<SubGraphInfo, SubGraphInfo>> between;
        std::tie(within, between) = st.extract_graphs_between(segments, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2);
        auto t3 = clk::now();

        FlatSide f1, f2;
        std::vector<uint8_t> only_del;
        std::vector<uint64_t> aln_off{0}, pairs;
        /* Stitcher::stitch emits  P0 A0 P1 A1 ... A(K-2) P(K-1): every subproblem but the first is preceded by
         * exactly one copied anchor (stitcher.hpp:157-203), so the anchors are dumped as a flat list. */
        std::vector<uint64_t> anchor_off{0}, anchor_pairs;
        Alignment stitched;
        double sub_s = 0;
        auto run = [&](const std::pair<SubGraphInfo, SubGraphInfo>& pr, bool od) {
            f1.add(pr.first); f2.add(pr.second);
            only_del.push_back(od ? 1 : 0);
            size_t before = stitched.size();
            auto a = clk::now();
            st.subalign(pr.first, pr.second, stitched, od);
            sub_s += std::chrono::duration<double>(clk::now() - a).count();
            for (size_t i = before; i < stitched.size(); ++i) { pairs.push_back(stitched[i].node_id1); pairs.push_back(stitched[i].node_id2); }
            aln_off.push_back(pairs.size() / 2);
        };
        for (size_t i = 0; i < between.size(); ++i) {
            if (i != 0) {
                const auto& seg_graphs = within[i - 1];
                const auto& seg = segments[i - 1];
                for (size_t j = 0; j < seg.size(); ++j) {
                    if (j != 0) run(seg_graphs[j - 1], false);
                    const auto& anchor = seg[j];
                    for (size_t k = 0; k < anchor.walk1.size(); ++k) {
                        stitched.emplace_back(anchor.walk1[k], anchor.walk2[k]);
                        anchor_pairs.push_back(anchor.walk1[k]); anchor_pairs.push_back(anchor.walk2[k]);
                    }
                    anchor_off.push_back(anchor_pairs.size() / 2);
                }
            }
            run(between[i], true);
        }
        auto t4 = clk::now();
        t_chain += std::chrono::duration<double>(t1 - t0).count();
        t_partition += std::chrono::duration<double>(t2 - t1).count();
        t_extract += std::chrono::duration<double>(t3 - t2).count();
        t_subalign += sub_s;
        (void)t4;
        if (dump) {
            dump_base_graph(*dump, pre + "parent1.", sp1.graph, sp1.tableau);
            dump_base_graph(*dump, pre + "parent2.", sp2.graph, sp2.tableau);
            {
                std::vector<uint64_t> seg_off{0}, walk_off{0};
                std::vector<uint32_t> w1, w2;
                for (const auto& seg : segments) {
                    for (const auto& a : seg) {
                        for (auto v : a.walk1) w1.push_back((uint32_t)v);
                        for (auto v : a.walk2) w2.push_back((uint32_t)v);
                        walk_off.push_back(w1.size());
                    }
                    seg_off.push_back(walk_off.size() - 1);
                }
                dump->u64(pre + "seg_off", seg_off); dump->u64(pre + "walk_off", walk_off);
                dump->u32(pre + "walk1", w1); dump->u32(pre + "walk2", w2);
            }
            dump->put(pre + "n_problems", 2, std::vector<uint64_t>{(uint64_t)only_del.size()}.data(), 1, 8);
            f1.write(*dump, pre + "g1.");
            f2.write(*dump, pre + "g2.");
            dump->u8(pre + "only_deletion_alns", only_del);
            dump->u64(pre + "aln_off", aln_off);
            dump->u64(pre + "pairs", pairs);
            dump->u64(pre + "anchor_off", anchor_off);
            dump->u64(pre + "anchor_pairs", anchor_pairs);
            std::vector<uint64_t> full;
            full.reserve(stitched.size() * 2);
            for (const auto& ap : stitched) { full.push_back(ap.node_id1); full.push_back(ap.node_id2); }
            dump->u64(pre + "stitched", full);
            dump->f64(pre + "subalign_seconds", std::vector<double>{sub_s});
        }
        return stitched;
    }

    void run(Dump* dump) {
        using clk = std::chrono::steady_clock;
        auto c0 = clk::now();
        if (!skip_calibration) calibrate_anchor_scores_and_identify_bonds(); /* core.cpp:63-68 */
        t_calib = std::chrono::duration<double>(clk::now() - c0).count();
        size_t merge = 0;
        while (!main_execution.finished()) {
            auto ptrs = main_execution.next();
            auto& next_problem = *std::get<0>(ptrs);
            auto& sp1 = *std::get<1>(ptrs);
            auto& sp2 = *std::get<2>(ptrs);
            reassign_sentinels(sp1.graph, sp1.tableau, 5, 6);
            reassign_sentinels(sp2.graph, sp2.tableau, 7, 8);
            auto m0 = clk::now();
            auto matches = path_match_finder.find_matches(sp1.graph, sp2.graph, sp1.tableau, sp2.tableau);
            t_match += std::chrono::duration<double>(clk::now() - m0).count();
            std::string pre = "m" + std::to_string(merge) + ".";
            if (dump) {
                /* the match sets exactly as PathMatchFinder returns them (anchor_chain reorders them later) */
                std::vector<uint64_t> so1{0}, so2{0}, wo1{0}, wo2{0}, c1, c2, fl;
                std::vector<uint32_t> n1, n2;
                for (const auto& ms : matches) {
                    for (const auto& w : ms.walks1) { for (auto v : w) n1.push_back((uint32_t)v); wo1.push_back(n1.size()); }
                    for (const auto& w : ms.walks2) { for (auto v : w) n2.push_back((uint32_t)v); wo2.push_back(n2.size()); }
                    so1.push_back(wo1.size() - 1); so2.push_back(wo2.size() - 1);
                    c1.push_back(ms.count1); c2.push_back(ms.count2); fl.push_back(ms.full_length);
                }
                dump->u64(pre + "ms.set_off1", so1); dump->u64(pre + "ms.walk_off1", wo1); dump->u32(pre + "ms.nodes1", n1);
                dump->u64(pre + "ms.set_off2", so2); dump->u64(pre + "ms.walk_off2", wo2); dump->u32(pre + "ms.nodes2", n2);
                dump->u64(pre + "ms.count1", c1); dump->u64(pre + "ms.count2", c2); dump->u64(pre + "ms.full_length", fl);
                dump->f64(pre + "score_scale", std::vector<double>{score_function.score_scale});
            }
            /* core.hpp:296-357: this driver covers the default SparseAffine / PathMerge<uint32,uint8> branch */
            PathMerge<uint32_t, uint8_t> pm1(sp1.graph, sp1.tableau);
            PathMerge<uint32_t, uint8_t> pm2(sp2.graph, sp2.tableau);
            next_problem.alignment = align_dump(matches, sp1, sp2, pm1, pm2, dump, pre);
            auto f0 = clk::now();
            BaseGraph fused = sp1.graph;
            fuse(fused, sp2.graph, sp1.tableau, sp2.tableau, next_problem.alignment);
            next_problem.graph = std::move(fused);
            next_problem.tableau = sp1.tableau;
            next_problem.complete = true;
            t_fuse += std::chrono::duration<double>(clk::now() - f0).count();
            ++merge;
        }
        if (dump) dump->put("n_merges", 2, std::vector<uint64_t>{(uint64_t)merge}.data(), 1, 8);
    }
};

}  // namespace

/* ---- chaining: the reference's DP on flat inputs ------------------------------------------------------------------ */
BaseGraph build_base_graph(const cl_base_graph* g, SentinelTableau& tableau) {
    BaseGraph bg;
    for (uint64_t v = 0; v < g->n_nodes; ++v) bg.add_node((char)g->label[v]);
    /* edges: honour both list orders with the same greedy merge as build_graph */
    clo_graph cg;
    cg.n = g->n_nodes; cg.label = g->label; cg.prev_off = g->prev_off; cg.prev_idx = g->prev_idx;
    cg.next_off = g->next_off; cg.next_idx = g->next_idx; cg.n_src = cg.n_snk = 0; cg.src = cg.snk = nullptr;
    BaseGraph with_edges = build_graph(&cg);
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        auto id = with_edges.add_path("p" + std::to_string(p));
        for (uint64_t i = g->path_off[p]; i < g->path_off[p + 1]; ++i) with_edges.extend_path(id, g->path_nodes[i]);
    }
    tableau.src_id = g->src_id;
    tableau.snk_id = g->snk_id;
    return with_edges;
}

struct OpenAnchorer : public Anchorer {
    explicit OpenAnchorer(const ScoreFunction& sf) : Anchorer(sf) {}
    using Anchorer::sparse_affine_chain_dp;
    using Anchorer::sparse_chain_dp;
    using Anchorer::estimate_score_scale;
    using Anchorer::split_branching_matches;
    using Anchorer::exhaustive_chain_dp;
};

template <class XMerge>
static int anchor_chain_with(int algo, const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, const clo_chain_params* cp,
                     int global_anchoring, uint64_t max_num_match_pairs, double score_scale, int autocalibrate, int fill_in,
                     int split_branching, uint64_t* anchors_out, int64_t* gap_before, int64_t* gap_after, double* gap_score_before,
                     double* gap_score_after, double* score, uint64_t* n_anchors, uint64_t* set_order_out, double* scale_out,
                     uint64_t* counts_out /* [3*n]: count1, count2, full_length */, uint64_t* walk_off_out /* [n+1] */,
                     uint32_t** walk1_out, uint32_t** walk2_out /* malloc'ed, release with ref_free */) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    std::vector<match_set_t> sets(ms->n_sets);
    std::map<std::vector<std::vector<uint64_t>>, uint64_t> identity;
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        for (uint64_t w = ms->set_off1[s]; w < ms->set_off1[s + 1]; ++w)
            sets[s].walks1.emplace_back(ms->nodes1 + ms->walk_off1[w], ms->nodes1 + ms->walk_off1[w + 1]);
        for (uint64_t w = ms->set_off2[s]; w < ms->set_off2[s + 1]; ++w)
            sets[s].walks2.emplace_back(ms->nodes2 + ms->walk_off2[w], ms->nodes2 + ms->walk_off2[w + 1]);
        sets[s].count1 = ms->count1[s];
        sets[s].count2 = ms->count2[s];
        sets[s].full_length = ms->full_length[s];
        if (!identity.emplace(sets[s].walks1, s).second) return -2;   // sets must be told apart by their walks
    }
    ScoreFunction sf;
    sf.anchor_score_function = (ScoreFunction::AnchorScore)cp->anchor_score_function;
    sf.pair_count_power = cp->pair_count_power;
    sf.length_intercept = cp->length_intercept;
    sf.length_decay_power = cp->length_decay_power;
    sf.score_scale = score_scale;
    OpenAnchorer an(sf);
    for (int i = 0; i < 3; ++i) { an.gap_open[i] = cp->gap_open[i]; an.gap_extend[i] = cp->gap_extend[i]; }
    an.global_anchoring = global_anchoring != 0;
    an.max_num_match_pairs = max_num_match_pairs;
    an.autocalibrate_gap_penalties = autocalibrate != 0;
    an.do_fill_in_anchoring = fill_in != 0;
    an.split_matches_at_branchpoints = split_branching != 0;
    an.chaining_algorithm = (Anchorer::ChainAlgorithm)algo;
    XMerge pm1(b1, t1), pm2(b2, t2);
    // the two steps of the public entry, run one after the other so that the estimated scale can be reported
    double scale = 1.0;
    if (split_branching) return -3;   // (would have to run before the estimate; not exposed here)
    std::vector<anchor_t> chain;
    if (algo == (int)Anchorer::SparseAffine) {
        if (autocalibrate) scale = an.estimate_score_scale(sets, b1, b2, t1, t2, pm1, pm2, false, nullptr, nullptr);
        chain = an.anchor_chain(sets, b1, b2, t1, t2, pm1, pm2, false, nullptr, &scale);
    } else {
        chain = an.anchor_chain(sets, b1, b2, t1, t2, pm1, pm2, false, nullptr, nullptr);   // (no estimate, no override: anchorer.hpp:973-984)
    }
    if (scale_out) *scale_out = scale;
    for (uint64_t k = 0; k < sets.size(); ++k) {
        auto it = identity.find(sets[k].walks1);
        if (it == identity.end()) return -4;
        set_order_out[k] = it->second;
    }
    *n_anchors = chain.size();
    {
        uint64_t total = 0;
        for (const auto& a : chain) total += a.walk1.size();
        *walk1_out = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
        *walk2_out = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
        uint64_t pos = 0;
        walk_off_out[0] = 0;
        for (size_t i = 0; i < chain.size(); ++i) {
            if (chain[i].walk1.size() != chain[i].walk2.size()) return -5;
            for (size_t j = 0; j < chain[i].walk1.size(); ++j) { (*walk1_out)[pos + j] = (uint32_t)chain[i].walk1[j]; (*walk2_out)[pos + j] = (uint32_t)chain[i].walk2[j]; }
            pos += chain[i].walk1.size();
            walk_off_out[i + 1] = pos;
        }
    }
    for (size_t i = 0; i < chain.size(); ++i) {
        anchors_out[3 * i] = chain[i].match_set;
        anchors_out[3 * i + 1] = chain[i].idx1;
        anchors_out[3 * i + 2] = chain[i].idx2;
        counts_out[3 * i] = chain[i].count1;
        counts_out[3 * i + 1] = chain[i].count2;
        counts_out[3 * i + 2] = chain[i].full_length;
        gap_before[i] = chain[i].gap_before;
        gap_after[i] = chain[i].gap_after;
        gap_score_before[i] = chain[i].gap_score_before;
        gap_score_after[i] = chain[i].gap_score_after;
        score[i] = chain[i].score;
    }
    return 0;
}


extern "C" {

/* algo 0: sparse_affine_chain_dp (anchorer.hpp:1812-2471), algo 1: sparse_chain_dp (:1511-1750); local anchoring,
 * the default (non memory-restrained) integer widths of anchor_chain (:1258-1290) */
int ref_chain_dp_ex(int algo, const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                    const clo_chain_params* cp, double local_scale, int global_anchoring, uint32_t* chain_out, uint64_t* chain_len,
                    double* seconds_out);
int ref_chain_dp(int algo, const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                 const clo_chain_params* cp, double local_scale, uint32_t* chain_out, uint64_t* chain_len, double* seconds_out) {
    return ref_chain_dp_ex(algo, g1, g2, ms, num_match_sets, cp, local_scale, 0, chain_out, chain_len, seconds_out);
}

/* global_anchoring != 0: sources = next(src sentinel), sinks = previous(snk sentinel), as anchorer.hpp:1069-1076 */
int ref_chain_dp_ex(int algo, const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                    const clo_chain_params* cp, double local_scale, int global_anchoring, uint32_t* chain_out, uint64_t* chain_len,
                    double* seconds_out) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    std::vector<match_set_t> sets(ms->n_sets);
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        for (uint64_t w = ms->set_off1[s]; w < ms->set_off1[s + 1]; ++w)
            sets[s].walks1.emplace_back(ms->nodes1 + ms->walk_off1[w], ms->nodes1 + ms->walk_off1[w + 1]);
        for (uint64_t w = ms->set_off2[s]; w < ms->set_off2[s + 1]; ++w)
            sets[s].walks2.emplace_back(ms->nodes2 + ms->walk_off2[w], ms->nodes2 + ms->walk_off2[w + 1]);
        sets[s].count1 = ms->count1[s];
        sets[s].count2 = ms->count2[s];
        sets[s].full_length = ms->full_length[s];
    }
    ScoreFunction sf;
    sf.anchor_score_function = (ScoreFunction::AnchorScore)cp->anchor_score_function;
    sf.pair_count_power = cp->pair_count_power;
    sf.length_intercept = cp->length_intercept;
    sf.length_decay_power = cp->length_decay_power;
    OpenAnchorer an(sf);
    std::array<double, 3> go{{cp->gap_open[0], cp->gap_open[1], cp->gap_open[2]}}, ge{{cp->gap_extend[0], cp->gap_extend[1], cp->gap_extend[2]}};
    PathMerge<uint32_t, uint8_t> pm1(b1, t1), pm2(b2, t2);
    using SmallMatchBank = MatchBank<uint32_t, uint16_t, float>;
    using SmallShiftMatchVector = std::vector<std::pair<int32_t, SmallMatchBank::match_id_t>>;
    using SmallDistMatchVector = std::vector<std::pair<uint32_t, SmallMatchBank::match_id_t>>;
    using FwdEdges = ForwardEdges<uint32_t, uint8_t>;
    std::vector<anchor_t> chain;
    const std::vector<uint64_t>* s1 = global_anchoring ? &b1.next(t1.src_id) : nullptr;
    const std::vector<uint64_t>* s2 = global_anchoring ? &b2.next(t2.src_id) : nullptr;
    const std::vector<uint64_t>* k1 = global_anchoring ? &b1.previous(t1.snk_id) : nullptr;
    const std::vector<uint64_t>* k2 = global_anchoring ? &b2.previous(t2.snk_id) : nullptr;
    auto a = std::chrono::steady_clock::now();
    if (algo == 2)   // exhaustive_chain_dp (anchorer.hpp:1342-1509), the O(M^2) algorithm behind "-g 0": no gap costs here
        chain = an.exhaustive_chain_dp<uint64_t, uint32_t>(sets, b1, b2, pm1, pm2, false, local_scale, num_match_sets, s1, s2, k1, k2, nullptr);
    else if (algo == 0)
        chain = an.sparse_affine_chain_dp<uint32_t, uint16_t, uint32_t, int32_t, uint32_t, float, SmallShiftMatchVector, SmallDistMatchVector,
                                          std::vector<uint32_t>, std::vector<uint32_t>, SmallMatchBank, FwdEdges>(
            sets, b1, b2, pm1, pm2, go, ge, local_scale, num_match_sets, true, s1, s2, k1, k2, nullptr);
    else
        chain = an.sparse_chain_dp<uint32_t, uint32_t, uint16_t, uint32_t, float, SmallDistMatchVector, std::vector<uint32_t>, SmallMatchBank, FwdEdges>(
            sets, b1, pm1, pm2, num_match_sets, true, s1, s2, k1, k2, nullptr);
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    *chain_len = chain.size();
    for (size_t i = 0; i < chain.size(); ++i) {
        chain_out[3 * i] = (uint32_t)chain[i].match_set;
        chain_out[3 * i + 1] = (uint32_t)chain[i].idx1;
        chain_out[3 * i + 2] = (uint32_t)chain[i].idx2;
    }
    return 0;
}

/* Anchorer::anchor_chain (the public entry, anchorer.hpp:958-996) on flat inputs.  set_order_out[k] = index (in the
 * caller's arrays) of the match set that sits at position k of the vector after the call (the call reorders it,
 * :1108-1173); anchors_out holds (match_set position, idx1, idx2).  Buffers sized for n_pairs anchors / n_sets sets. */
int ref_anchor_chain(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, const clo_chain_params* cp,
                     int global_anchoring, uint64_t max_num_match_pairs, double score_scale, int autocalibrate, int fill_in,
                     int split_branching, uint64_t* anchors_out, int64_t* gap_before, int64_t* gap_after, double* gap_score_before,
                     double* gap_score_after, double* score, uint64_t* n_anchors, uint64_t* set_order_out, double* scale_out,
                     uint64_t* counts_out, uint64_t* walk_off_out, uint32_t** walk1_out, uint32_t** walk2_out) {
    return anchor_chain_with<PathMerge<uint32_t, uint8_t>>((int)Anchorer::SparseAffine, g1, g2, ms, cp, global_anchoring, max_num_match_pairs, score_scale, autocalibrate, fill_in,
                                                            split_branching, anchors_out, gap_before, gap_after, gap_score_before, gap_score_after, score, n_anchors, set_order_out,
                                                            scale_out, counts_out, walk_off_out, walk1_out, walk2_out);
}

/* the same with Anchorer::chaining_algorithm = algo (the CLI's hidden -g: 0 Exhaustive, 1 Sparse) over ChainMerge structures, as Core::execute
 * builds them for those algorithms (include/centrolign/core.hpp:350-357) */
int ref_anchor_chain_algo(int algo, const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, const clo_chain_params* cp,
                          int global_anchoring, uint64_t max_num_match_pairs, double score_scale, int autocalibrate, int fill_in,
                          int split_branching, uint64_t* anchors_out, int64_t* gap_before, int64_t* gap_after, double* gap_score_before,
                          double* gap_score_after, double* score, uint64_t* n_anchors, uint64_t* set_order_out, double* scale_out,
                          uint64_t* counts_out, uint64_t* walk_off_out, uint32_t** walk1_out, uint32_t** walk2_out) {
    if (algo == (int)Anchorer::SparseAffine)
        return ref_anchor_chain(g1, g2, ms, cp, global_anchoring, max_num_match_pairs, score_scale, autocalibrate, fill_in, split_branching, anchors_out, gap_before, gap_after,
                                gap_score_before, gap_score_after, score, n_anchors, set_order_out, scale_out, counts_out, walk_off_out, walk1_out, walk2_out);
    return anchor_chain_with<ChainMerge>(algo, g1, g2, ms, cp, global_anchoring, max_num_match_pairs, score_scale, autocalibrate, fill_in,
                                         split_branching, anchors_out, gap_before, gap_after, gap_score_before, gap_score_after, score, n_anchors, set_order_out,
                                         scale_out, counts_out, walk_off_out, walk1_out, walk2_out);
}

void ref_free(void* p) { free(p); }

/* Anchorer::anchor_chain WITH masked matches and an overriding scale (anchorer.hpp:135-145), as the tandem-duplication rounds of
 * Core::calibrate_anchor_scores_and_identify_bonds call it (src/core.cpp:221-227).  mask_in: [3*n_mask] (set, idx1, idx2) in the indexing of
 * `ms`; the call reorders the sets and re-indexes the mask in place: set_order_out as in ref_anchor_chain, *mask_out (malloc'ed, [3 * *n_mask_out])
 * is the mask AFTER the call in the reordered indexing.  Only the chain's identities, walks and scores are returned. */
int ref_anchor_chain_masked(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, const clo_chain_params* cp,
                            int global_anchoring, uint64_t max_num_match_pairs, double score_scale, int fill_in,
                            const uint64_t* mask_in, uint64_t n_mask, const double* override_scale,
                            uint64_t* anchors_out, double* score, uint64_t* n_anchors, uint64_t* set_order_out,
                            uint64_t* walk_off_out, uint32_t** walk1_out, uint32_t** walk2_out, uint64_t** mask_out, uint64_t* n_mask_out) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    std::vector<match_set_t> sets(ms->n_sets);
    std::map<std::pair<std::vector<std::vector<uint64_t>>, std::vector<std::vector<uint64_t>>>, uint64_t> identity;
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        for (uint64_t w = ms->set_off1[s]; w < ms->set_off1[s + 1]; ++w)
            sets[s].walks1.emplace_back(ms->nodes1 + ms->walk_off1[w], ms->nodes1 + ms->walk_off1[w + 1]);
        for (uint64_t w = ms->set_off2[s]; w < ms->set_off2[s + 1]; ++w)
            sets[s].walks2.emplace_back(ms->nodes2 + ms->walk_off2[w], ms->nodes2 + ms->walk_off2[w + 1]);
        sets[s].count1 = ms->count1[s];
        sets[s].count2 = ms->count2[s];
        sets[s].full_length = ms->full_length[s];
        if (!identity.emplace(std::make_pair(sets[s].walks1, sets[s].walks2), s).second) return -2;
    }
    ScoreFunction sf;
    sf.anchor_score_function = (ScoreFunction::AnchorScore)cp->anchor_score_function;
    sf.pair_count_power = cp->pair_count_power;
    sf.length_intercept = cp->length_intercept;
    sf.length_decay_power = cp->length_decay_power;
    sf.score_scale = score_scale;
    OpenAnchorer an(sf);
    for (int i = 0; i < 3; ++i) { an.gap_open[i] = cp->gap_open[i]; an.gap_extend[i] = cp->gap_extend[i]; }
    an.global_anchoring = global_anchoring != 0;
    an.max_num_match_pairs = max_num_match_pairs;
    an.autocalibrate_gap_penalties = true;
    an.do_fill_in_anchoring = fill_in != 0;
    an.split_matches_at_branchpoints = false;
    an.chaining_algorithm = Anchorer::SparseAffine;
    PathMerge<> pm1(b1, t1), pm2(b2, t2);
    std::unordered_set<std::tuple<size_t, size_t, size_t>> mask;
    for (uint64_t i = 0; i < n_mask; ++i) mask.emplace(mask_in[3 * i], mask_in[3 * i + 1], mask_in[3 * i + 2]);
    double scale = override_scale ? *override_scale : 1.0;
    std::vector<anchor_t> chain = an.anchor_chain(sets, b1, b2, t1, t2, pm1, pm2, false, &mask, override_scale ? &scale : nullptr);
    for (uint64_t k = 0; k < sets.size(); ++k) {
        auto it = identity.find(std::make_pair(sets[k].walks1, sets[k].walks2));
        if (it == identity.end()) return -4;
        set_order_out[k] = it->second;
    }
    *n_anchors = chain.size();
    uint64_t total = 0;
    for (const auto& a : chain) total += a.walk1.size();
    *walk1_out = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    *walk2_out = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    uint64_t pos = 0;
    walk_off_out[0] = 0;
    for (size_t i = 0; i < chain.size(); ++i) {
        for (size_t j = 0; j < chain[i].walk1.size(); ++j) { (*walk1_out)[pos + j] = (uint32_t)chain[i].walk1[j]; (*walk2_out)[pos + j] = (uint32_t)chain[i].walk2[j]; }
        pos += chain[i].walk1.size();
        walk_off_out[i + 1] = pos;
        anchors_out[3 * i] = chain[i].match_set;
        anchors_out[3 * i + 1] = chain[i].idx1;
        anchors_out[3 * i + 2] = chain[i].idx2;
        score[i] = chain[i].score;
    }
    std::vector<std::tuple<size_t, size_t, size_t>> sorted(mask.begin(), mask.end());
    std::sort(sorted.begin(), sorted.end());
    *n_mask_out = sorted.size();
    *mask_out = (uint64_t*)malloc((sorted.size() ? sorted.size() : 1) * 3 * sizeof(uint64_t));
    for (size_t i = 0; i < sorted.size(); ++i) {
        (*mask_out)[3 * i] = std::get<0>(sorted[i]); (*mask_out)[3 * i + 1] = std::get<1>(sorted[i]); (*mask_out)[3 * i + 2] = std::get<2>(sorted[i]);
    }
    return 0;
}

/* typed overrides of the reference's Parameters, "b:name=1;i:name=5000;d:name=0.2;s:name=text" (what the CLI's options and config file set) */
static void apply_overrides(Parameters& params, const char* overrides) {
    if (!overrides) return;
    std::stringstream ss(overrides);
    std::string item;
    while (std::getline(ss, item, ';')) {
        if (item.size() < 4 || item[1] != ':') continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos) continue;
        const std::string name = item.substr(2, eq - 2), val = item.substr(eq + 1);
        switch (item[0]) {
            case 'b': params.set<bool>(name, val != "0"); break;
            case 'i': params.set<int64_t>(name, (int64_t)atoll(val.c_str())); break;
            case 'd': params.set<double>(name, atof(val.c_str())); break;
            default: params.set<std::string>(name, val); break;
        }
    }
}

/* The CLI's -c flow (src/core.cpp:63-94) with the cyclisation steps opened up so that every intermediate result can be recorded:
 * per leaf the calibration chain, per tandem-duplication round the secondary chain, the bonds and the bond alignments (:196-297);
 * after the MSA the graph Core::apply_bonds fuses (:613-637), what simplify_bubbles makes of it, the inconsistencies, and the polished graph.
 * The steps are the reference's own functions called in the reference's order; tests/golden/make_golden.py checks that the text that comes
 * out equals the unmodified flow's (oracle/_ref/ref_cli with the same parameters). */
static void dump_chain(Dump& d, const std::string& pre, const std::vector<anchor_t>& chain) {
    std::vector<uint64_t> walk_off{0};
    std::vector<uint32_t> w1, w2;
    std::vector<double> score, gsa;
    std::vector<int64_t> ga;
    for (const auto& a : chain) {
        for (auto v : a.walk1) w1.push_back((uint32_t)v);
        for (auto v : a.walk2) w2.push_back((uint32_t)v);
        walk_off.push_back(w1.size());
        score.push_back(a.score); ga.push_back(a.gap_after); gsa.push_back(a.gap_score_after);
    }
    d.u64(pre + "walk_off", walk_off); d.u32(pre + "walk1", w1); d.u32(pre + "walk2", w2);
    d.f64(pre + "score", score); d.i64(pre + "gap_after", ga); d.f64(pre + "gap_score_after", gsa);
}
static void dump_bonds(Dump& d, const std::string& pre, const std::vector<bond_interval_t>& bonds) {
    std::vector<uint64_t> off{0}, o1, o2, len;
    std::vector<double> sc;
    for (const auto& iv : bonds) {
        for (const auto& b : iv) { o1.push_back(b.offset1); o2.push_back(b.offset2); len.push_back(b.length); sc.push_back(b.score); }
        off.push_back(o1.size());
    }
    d.u64(pre + "interval_off", off); d.u64(pre + "offset1", o1); d.u64(pre + "offset2", o2); d.u64(pre + "length", len); d.f64(pre + "score", sc);
}
static void dump_alignment(Dump& d, const std::string& name, const Alignment& aln) {
    std::vector<uint64_t> flat;
    for (const auto& ap : aln) { flat.push_back(ap.node_id1); flat.push_back(ap.node_id2); }
    d.u64(name, flat);
}
struct CycCore : public Core {
    CycCore(std::vector<std::pair<std::string, std::string>>&& seqs, Tree&& tree) : Core(std::move(seqs), std::move(tree)) {}

    std::vector<std::pair<std::string, Alignment>> calibrate_and_bond(Dump* dump) {   /* src/core.cpp:96-299, cyclising, no restart */
        std::vector<double> intrinsic_scales;
        std::vector<std::pair<std::string, Alignment>> bond_alns;
        auto leaves = main_execution.leaf_subproblems();
        std::vector<std::pair<std::vector<match_set_t>, std::vector<anchor_t>>> memo(leaves.size());
        for (size_t i = 0; i < leaves.size(); ++i) {
            auto& sp = *leaves[i];
            reassign_sentinels(sp.graph, sp.tableau, 5, 6);
            SentinelTableau dummy = sp.tableau;
            dummy.src_sentinel = 7; dummy.snk_sentinel = 8;
            std::vector<match_set_t> matches = path_match_finder.find_matches(sp.graph, sp.graph, sp.tableau, dummy);
            std::vector<match_set_t> diagonal;
            diagonal.reserve(matches.size());
            for (const auto& ms : matches) {
                for (const auto& walk : ms.walks1) {
                    diagonal.emplace_back();
                    auto& m = diagonal.back();
                    m.walks1.emplace_back(walk); m.walks2.emplace_back(walk);
                    m.count1 = ms.count1; m.count2 = ms.count2; m.full_length = ms.full_length;
                }
            }
            ChainMerge chain_merge(sp.graph, sp.tableau);
            std::vector<anchor_t> chain;
            bool restrain = anchorer.max_num_match_pairs > memory_restraint_size;
            double scale = anchorer.estimate_score_scale(diagonal, sp.graph, sp.graph, sp.tableau, sp.tableau, chain_merge, chain_merge, restrain, &chain);
            intrinsic_scales.push_back(scale);
            memo[i].first = std::move(matches);
            memo[i].second = std::move(chain);
        }
        if (!skip_calibration) {
            double mean = 0.0;
            for (auto sc : intrinsic_scales) mean += sc;
            mean /= intrinsic_scales.size();
            score_function.score_scale = mean;
        }
        if (dump) { dump->f64("intrinsic_scales", intrinsic_scales); dump->f64("score_scale", std::vector<double>{score_function.score_scale}); }
        for (size_t i = 0; i < leaves.size(); ++i) {
            auto& sp = *leaves[i];
            const std::string lp = "leaf" + std::to_string(i) + ".";
            PathMerge<> path_merge(sp.graph, sp.tableau);
            auto matches = std::move(memo[i].first);
            auto chain = std::move(memo[i].second);
            if (dump) dump_chain(*dump, lp + "opt.", chain);
            auto mask = generate_diagonal_mask(matches);
            StepIndex step_index;
            size_t bonds_identified = 0, rounds = 0;
            for (size_t iter = 0; iter < max_tandem_duplication_search_rounds; ++iter) {
                auto secondary = anchorer.anchor_chain(matches, sp.graph, sp.graph, sp.tableau, sp.tableau, path_merge, path_merge,
                                                       anchorer.max_num_match_pairs * log2(anchorer.max_num_match_pairs) > memory_restraint_size,
                                                       &mask, &intrinsic_scales[i]);
                auto bonds = bonder.identify_bonds(sp.graph, sp.graph, sp.tableau, sp.tableau, path_merge, path_merge, chain, secondary);
                const std::string rp = lp + "r" + std::to_string(iter) + ".";
                if (dump) { dump_chain(*dump, rp + "sec.", secondary); dump_bonds(*dump, rp + "raw_bonds.", bonds); }
                bonder.deduplicate_self_bonds(bonds);
                if (dump) dump_bonds(*dump, rp + "bonds.", bonds);
                ++rounds;
                if (bonds.empty()) break;
                if (iter == 0) step_index = std::move(StepIndex(sp.graph));
                for (auto& bond : bonds) {
                    auto bond_chain = bonds_to_chain(sp.graph, bond);
                    bond_alns.emplace_back(sp.graph.path_name(0), stitcher.internal_stitch(bond_chain, sp.graph, path_merge));
                    for (auto& ap : bond_alns.back().second) {
                        if (ap.node_id1 != AlignedPair::gap) ap.node_id1 = step_index.path_steps(ap.node_id1).front().second;
                        if (ap.node_id2 != AlignedPair::gap) ap.node_id2 = step_index.path_steps(ap.node_id2).front().second;
                    }
                    if (dump) dump_alignment(*dump, lp + "bond_aln" + std::to_string(bonds_identified), bond_alns.back().second);
                    ++bonds_identified;
                }
                update_mask(matches, secondary, mask, true);
            }
            if (dump) dump->u64(lp + "counts", std::vector<uint64_t>{(uint64_t)rounds, (uint64_t)bonds_identified});
        }
        return bond_alns;
    }

    /* Core::polish_cyclized_graph (src/core.cpp:650-767) call for call, with the loop of do_execution (core.hpp:256-403) opened up so that every
     * realignment's subpaths, induced match sets and result can be recorded */
    void polish_dump(Subproblem& subproblem, Dump& dump) {
        auto inconsistencies = inconsistency_identifier.identify_inconsistencies(subproblem.graph, subproblem.tableau);
        if (inconsistencies.empty()) return;
        StepIndex step_index(subproblem.graph);
        reassign_sentinels(subproblem.graph, subproblem.tableau, 5, 6);
        SentinelTableau dummy = subproblem.tableau;
        dummy.src_sentinel = 7; dummy.snk_sentinel = 8;
        std::vector<match_set_t> full = path_match_finder.find_matches(subproblem.graph, subproblem.graph, subproblem.tableau, dummy);
        dump.u64("polish.full_match_sets", std::vector<uint64_t>{(uint64_t)full.size()});
        InducedMatchFinder induced(subproblem.graph, full, inconsistencies, step_index);
        std::vector<Subproblem> realigned;
        for (size_t i = 0; i < inconsistencies.size(); ++i) {
            auto inc = inconsistencies[i];
            std::unordered_map<uint64_t, std::pair<std::vector<size_t>, std::vector<size_t>>> loc;
            for (auto step : step_index.path_steps(inc.first)) loc[step.first].first.push_back(step.second);
            for (auto step : step_index.path_steps(inc.second)) loc[step.first].second.push_back(step.second);
            std::vector<uint64_t> path_ids;
            for (auto& kv : loc) path_ids.push_back(kv.first);
            std::sort(path_ids.begin(), path_ids.end());
            std::vector<std::tuple<uint64_t, size_t, size_t>> intervals;
            std::vector<std::pair<std::string, std::string>> subpaths;
            for (auto pid : path_ids) {
                const auto& l = loc[pid];
                for (size_t k = 0; k < l.first.size(); ++k) {
                    intervals.emplace_back(pid, l.first[k], l.second[k]);
                    subpaths.emplace_back();
                    subpaths.back().first = get_subpath_name(subproblem.graph.path_name(pid), l.first[k], l.second[k]);
                    for (size_t j = l.first[k]; j <= l.second[k]; ++j) subpaths.back().second.push_back(decode_base(subproblem.graph.label(subproblem.graph.path(pid)[j])));
                }
            }
            const std::string rp = "region" + std::to_string(i) + ".";
            {
                std::string names;
                for (const auto& sp : subpaths) names += sp.first + "\n";
                dump.str(rp + "names", names);
            }
            auto expanded = make_copy_expanded_tree(intervals, subpaths);
            dump.str(rp + "tree", expanded.to_newick());
            Execution realignment(std::move(subpaths), std::move(expanded));
            auto view = induced.component_view(i);
            size_t merge = 0;
            auto saved = logging::level;
            while (!realignment.finished()) {
                auto ptrs = realignment.next();
                auto& next_problem = *std::get<0>(ptrs);
                auto& sp1 = *std::get<1>(ptrs);
                auto& sp2 = *std::get<2>(ptrs);
                reassign_sentinels(sp1.graph, sp1.tableau, 5, 6);
                reassign_sentinels(sp2.graph, sp2.tableau, 7, 8);
                auto matches = view.find_matches(sp1.graph, sp2.graph, sp1.tableau, sp2.tableau);
                {
                    std::vector<uint64_t> flat;   /* per set: n walks1, n walks2, length, count1, count2, full_length, then the walks' first nodes */
                    for (const auto& ms : matches) {
                        flat.push_back(ms.walks1.size()); flat.push_back(ms.walks2.size()); flat.push_back(ms.walks1.empty() ? 0 : ms.walks1.front().size());
                        flat.push_back(ms.count1); flat.push_back(ms.count2); flat.push_back(ms.full_length);
                        for (const auto& w : ms.walks1) flat.push_back(w.front());
                        for (const auto& w : ms.walks2) flat.push_back(w.front());
                    }
                    dump.u64(rp + "m" + std::to_string(merge) + ".matches", flat);
                }
                PathMerge<> pm1(sp1.graph, sp1.tableau), pm2(sp2.graph, sp2.tableau);
                next_problem.alignment = align(matches, sp1, sp2, pm1, pm2, false);
                dump_alignment(dump, rp + "m" + std::to_string(merge) + ".alignment", next_problem.alignment);
                BaseGraph fused = std::move(sp1.graph);
                fuse(fused, sp2.graph, sp1.tableau, sp2.tableau, next_problem.alignment);
                next_problem.graph = std::move(fused);
                next_problem.tableau = sp1.tableau;
                next_problem.complete = true;
                ++merge;
            }
            logging::level = saved;
            dump_base_graph(dump, rp + "graph.", realignment.final_subproblem().graph, realignment.final_subproblem().tableau);
            realigned.emplace_back(std::move(realignment.final_subproblem()));
        }
        integrate_polished_subgraphs(subproblem, realigned);
    }

    void run(Dump* dump) {   /* src/core.cpp:63-94 + apply_bonds (:594-648) */
        auto bond_alignments = calibrate_and_bond(dump);
        do_execution(main_execution, this->path_match_finder, true);
        auto& root = main_execution.final_subproblem();
        if (dump) dump_base_graph(*dump, "msa.", root.graph, root.tableau);
        if (bond_alignments.empty()) return;
        std::vector<Alignment> to_fuse;
        for (auto& ba : bond_alignments) {
            uint64_t path_id = root.graph.path_id(ba.first);
            for (auto& ap : ba.second) {
                if (ap.node_id1 != AlignedPair::gap) ap.node_id1 = root.graph.path(path_id)[ap.node_id1];
                if (ap.node_id2 != AlignedPair::gap) ap.node_id2 = root.graph.path(path_id)[ap.node_id2];
            }
            to_fuse.emplace_back(std::move(ba.second));
        }
        SentinelTableau ct;
        Alignment ca;
        BaseGraph cyclized = internal_fuse(root.graph, to_fuse, &root.tableau, &ct, &root.alignment, &ca);
        if (dump) dump_base_graph(*dump, "fused.", cyclized, ct);
        simplify_bubbles(cyclized, ct);
        if (dump) dump_base_graph(*dump, "simplified.", cyclized, ct);
        root.graph = std::move(cyclized);
        root.tableau = ct;
        root.alignment.clear();
        if (dump) {
            auto inc = inconsistency_identifier.identify_inconsistencies(root.graph, root.tableau);
            std::vector<uint64_t> flat;
            for (const auto& b : inc) { flat.push_back(b.first); flat.push_back(b.second); }
            dump->u64("inconsistencies", flat);
        }
        if (dump) polish_dump(root, *dump);
        else polish_cyclized_graph(root);
        if (dump) dump_base_graph(*dump, "polished.", root.graph, root.tableau);
    }
};

int ref_cyclize_dump(const char* fasta_path, const char* newick_path, const char* dump_path, const char* out_path, const char* overrides, int verbosity) {
    try {
        Parameters params;
        params.set<std::string>("fasta_name", fasta_path);
        params.set<bool>("cyclize_tandem_duplications", true);
        apply_overrides(params, overrides);
        params.validate();
        logging::level = (logging::LoggingLevel)verbosity;
        std::ifstream fin(fasta_path);
        if (!fin) return -100;
        auto parsed = parse_fasta(fin);
        std::vector<std::string> names;
        for (const auto& p : parsed) names.push_back(p.first);
        std::string newick;
        if (newick_path && *newick_path) {
            std::ifstream tin(newick_path);
            std::stringstream ss;
            ss << tin.rdbuf();
            newick = ss.str();
        } else {
            newick = in_order_newick_string(names);
        }
        Tree tree(newick);
        CycCore core(std::move(parsed), std::move(tree));
        if (names.size() == 2) params.set<bool>("preserve_subproblems", true);
        params.apply(core);
        Dump dump;
        Dump* dp = nullptr;
        if (dump_path && *dump_path) {
            if (!dump.open(dump_path)) return -101;
            dp = &dump;
        }
        core.run(dp);
        std::stringstream ss;
        const auto& root = core.root_subproblem();
        write_gfa(root.graph, root.tableau, ss);
        if (dp) { dump.str("output", ss.str()); dump.close(); }
        if (out_path && *out_path) {
            std::ofstream fo(out_path);
            fo << ss.str();
        }
        return 0;
    } catch (std::exception& ex) {
        fprintf(stderr, "ref_cyclize_dump: %s\n", ex.what());
        return -102;
    }
}

/* Core::generate_diagonal_mask (mode 0) and Core::update_mask (mode 1, mask_reciprocal as given) on flat sets (src/core.cpp:301-372). */
struct MaskCore : public Core {
    MaskCore(std::vector<std::pair<std::string, std::string>>&& seqs, Tree&& tree) : Core(std::move(seqs), std::move(tree)) {}
    using Core::generate_diagonal_mask;
    using Core::update_mask;
};
int ref_masks(int mode, const clo_match_sets* ms, uint64_t n_chain, const uint64_t* chain_walk_off, const uint32_t* chain_walk1, const uint32_t* chain_walk2,
              int mask_reciprocal, const uint64_t* mask_in, uint64_t n_mask, uint64_t** mask_out, uint64_t* n_mask_out) {
    std::vector<match_set_t> sets(ms->n_sets);
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        for (uint64_t w = ms->set_off1[s]; w < ms->set_off1[s + 1]; ++w)
            sets[s].walks1.emplace_back(ms->nodes1 + ms->walk_off1[w], ms->nodes1 + ms->walk_off1[w + 1]);
        for (uint64_t w = ms->set_off2[s]; w < ms->set_off2[s + 1]; ++w)
            sets[s].walks2.emplace_back(ms->nodes2 + ms->walk_off2[w], ms->nodes2 + ms->walk_off2[w + 1]);
    }
    std::vector<std::pair<std::string, std::string>> seqs{{"a", "ACGT"}, {"b", "ACGT"}};
    MaskCore core(std::move(seqs), Tree("(a,b);"));
    std::unordered_set<std::tuple<size_t, size_t, size_t>> mask;
    if (mode == 0) mask = core.generate_diagonal_mask(sets);
    else {
        for (uint64_t i = 0; i < n_mask; ++i) mask.emplace(mask_in[3 * i], mask_in[3 * i + 1], mask_in[3 * i + 2]);
        std::vector<anchor_t> chain(n_chain);
        for (uint64_t a = 0; a < n_chain; ++a) {
            chain[a].walk1.assign(chain_walk1 + chain_walk_off[a], chain_walk1 + chain_walk_off[a + 1]);
            chain[a].walk2.assign(chain_walk2 + chain_walk_off[a], chain_walk2 + chain_walk_off[a + 1]);
        }
        core.update_mask(sets, chain, mask, mask_reciprocal != 0);
    }
    std::vector<std::tuple<size_t, size_t, size_t>> sorted(mask.begin(), mask.end());
    std::sort(sorted.begin(), sorted.end());
    *n_mask_out = sorted.size();
    *mask_out = (uint64_t*)malloc((sorted.size() ? sorted.size() : 1) * 3 * sizeof(uint64_t));
    for (size_t i = 0; i < sorted.size(); ++i) {
        (*mask_out)[3 * i] = std::get<0>(sorted[i]); (*mask_out)[3 * i + 1] = std::get<1>(sorted[i]); (*mask_out)[3 * i + 2] = std::get<2>(sorted[i]);
    }
    return 0;
}

/* Partitioner::partition_anchors (partitioner.hpp:72-213) on flat anchors; segments_out gets (first, past-the-last) anchor
 * index pairs (buffer for n_anchors pairs) */
int ref_partition_anchors(const cl_base_graph* g1, const cl_base_graph* g2, uint64_t n_anchors, const uint64_t* walk_off,
                          const uint32_t* walk1, const uint32_t* walk2, const uint64_t* count1, const uint64_t* count2,
                          const uint64_t* full_length, const uint64_t* match_set, const double* score, const clo_chain_params* cp,
                          int constraint_method, double minimum_segment_score, double minimum_segment_average, double window_length,
                          double generalized_length_mean, double boundary_score_factor, double score_scale, int score_boundaries,
                          int use_annotated_score, uint64_t* segments_out, uint64_t* n_segments_out) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    std::vector<anchor_t> anchors(n_anchors);
    for (uint64_t i = 0; i < n_anchors; ++i) {
        anchors[i].walk1.assign(walk1 + walk_off[i], walk1 + walk_off[i + 1]);
        anchors[i].walk2.assign(walk2 + walk_off[i], walk2 + walk_off[i + 1]);
        anchors[i].count1 = count1[i];
        anchors[i].count2 = count2[i];
        anchors[i].full_length = full_length[i];
        anchors[i].match_set = match_set[i];
        anchors[i].score = score[i];
        anchors[i].idx1 = i;   // not read by the partitioner: carries the identity through the move
    }
    ScoreFunction sf;
    sf.anchor_score_function = (ScoreFunction::AnchorScore)cp->anchor_score_function;
    sf.pair_count_power = cp->pair_count_power;
    sf.length_intercept = cp->length_intercept;
    sf.length_decay_power = cp->length_decay_power;
    sf.score_scale = score_scale;
    Partitioner part(sf);
    part.constraint_method = (Partitioner::ConstraintMethod)constraint_method;
    part.minimum_segment_score = minimum_segment_score;
    part.minimum_segment_average = minimum_segment_average;
    part.window_length = window_length;
    part.generalized_length_mean = generalized_length_mean;
    part.boundary_score_factor = boundary_score_factor;
    PathMerge<uint32_t, uint8_t> pm1(b1, t1), pm2(b2, t2);
    auto segments = part.partition_anchors(anchors, b1, b2, t1, t2, pm1, pm2, score_boundaries != 0, use_annotated_score != 0);
    *n_segments_out = segments.size();
    for (size_t i = 0; i < segments.size(); ++i) {
        if (segments[i].empty()) return -2;
        segments_out[2 * i] = segments[i].front().idx1;
        segments_out[2 * i + 1] = segments[i].back().idx1 + 1;
        for (size_t j = 0; j < segments[i].size(); ++j)
            if (segments[i][j].idx1 != segments[i].front().idx1 + j) return -3;
    }
    return 0;
}

/* Anchorer::split_branching_matches (anchorer.hpp:800-956) on flat inputs.  Output: the resulting sets as
 * (source set, n walks1, n walks2, walk length) rows plus every walk's nodes, graph-1 walks then graph-2 walks per set. */
int ref_split_branching_matches(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t anchor_split_limit,
                                uint64_t min_split_length, uint64_t min_path_length_spread, uint64_t max_split_match_set_size,
                                uint64_t* n_sets_out, uint64_t** rows_out, uint32_t** nodes_out, uint64_t* n_nodes_out) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    std::vector<match_set_t> sets(ms->n_sets);
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        for (uint64_t w = ms->set_off1[s]; w < ms->set_off1[s + 1]; ++w)
            sets[s].walks1.emplace_back(ms->nodes1 + ms->walk_off1[w], ms->nodes1 + ms->walk_off1[w + 1]);
        for (uint64_t w = ms->set_off2[s]; w < ms->set_off2[s + 1]; ++w)
            sets[s].walks2.emplace_back(ms->nodes2 + ms->walk_off2[w], ms->nodes2 + ms->walk_off2[w + 1]);
        sets[s].count1 = ms->count1[s];
        sets[s].count2 = ms->count2[s];
        sets[s].full_length = ms->full_length[s];
    }
    ScoreFunction sf;
    OpenAnchorer an(sf);
    an.anchor_split_limit = anchor_split_limit;
    an.min_split_length = min_split_length;
    an.min_path_length_spread = min_path_length_spread;
    an.max_split_match_set_size = max_split_match_set_size;
    an.split_branching_matches(sets, b1, b2, t1, t2, nullptr);
    *n_sets_out = sets.size();
    uint64_t total = 0;
    for (const auto& st : sets) { for (const auto& w : st.walks1) total += w.size(); for (const auto& w : st.walks2) total += w.size(); }
    *rows_out = (uint64_t*)malloc((sets.size() ? sets.size() : 1) * 6 * sizeof(uint64_t));
    *nodes_out = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    *n_nodes_out = total;
    uint64_t pos = 0;
    for (size_t s = 0; s < sets.size(); ++s) {
        uint64_t* row = *rows_out + 6 * s;
        row[0] = sets[s].walks1.size(); row[1] = sets[s].walks2.size(); row[2] = sets[s].walks1.front().size();
        row[3] = sets[s].count1; row[4] = sets[s].count2; row[5] = sets[s].full_length;
        for (const auto& w : sets[s].walks1) for (auto v : w) (*nodes_out)[pos++] = (uint32_t)v;
        for (const auto& w : sets[s].walks2) for (auto v : w) (*nodes_out)[pos++] = (uint32_t)v;
    }
    return 0;
}

/* PathMatchFinder::find_matches (match_finder.hpp:120-131 -> query_index :133-212 -> PathESA) on flat inputs; the sentinel
 * characters are the labels of the tableau nodes.  Output rows/nodes as ref_split_branching_matches. */
int ref_find_matches(const cl_base_graph* g1, const cl_base_graph* g2, const clo_chain_params* cp, uint64_t max_count, int use_css,
                     uint64_t* n_sets_out, uint64_t** rows_out, uint32_t** nodes_out, uint64_t* n_nodes_out) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    t1.src_sentinel = (char)g1->label[g1->src_id]; t1.snk_sentinel = (char)g1->label[g1->snk_id];
    t2.src_sentinel = (char)g2->label[g2->src_id]; t2.snk_sentinel = (char)g2->label[g2->snk_id];
    ScoreFunction sf;
    sf.anchor_score_function = (ScoreFunction::AnchorScore)cp->anchor_score_function;
    sf.pair_count_power = cp->pair_count_power;
    sf.length_intercept = cp->length_intercept;
    sf.length_decay_power = cp->length_decay_power;
    PathMatchFinder mf(sf);
    mf.max_count = max_count;
    mf.use_color_set_size = use_css != 0;
    std::vector<match_set_t> sets = mf.find_matches(b1, b2, t1, t2);
    *n_sets_out = sets.size();
    uint64_t total = 0;
    for (const auto& st : sets) { for (const auto& w : st.walks1) total += w.size(); for (const auto& w : st.walks2) total += w.size(); }
    *rows_out = (uint64_t*)malloc((sets.size() ? sets.size() : 1) * 6 * sizeof(uint64_t));
    *nodes_out = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    *n_nodes_out = total;
    uint64_t pos = 0;
    for (size_t s = 0; s < sets.size(); ++s) {
        uint64_t* row = *rows_out + 6 * s;
        row[0] = sets[s].walks1.size(); row[1] = sets[s].walks2.size(); row[2] = sets[s].walks1.front().size();
        row[3] = sets[s].count1; row[4] = sets[s].count2; row[5] = sets[s].full_length;
        for (const auto& w : sets[s].walks1) for (auto v : w) (*nodes_out)[pos++] = (uint32_t)v;
        for (const auto& w : sets[s].walks2) for (auto v : w) (*nodes_out)[pos++] = (uint32_t)v;
    }
    return 0;
}

/* the per-leaf step of Core::calibrate_anchor_scores_and_identify_bonds (src/core.cpp:122-166) with the reference's own
 * classes: self matches, the main-diagonal subset, Anchorer::estimate_score_scale over a ChainMerge */
int ref_leaf_intrinsic_scale(const cl_base_graph* g, const clo_chain_params* cp, uint64_t max_count, int global_anchoring,
                             uint64_t max_num_match_pairs, int fill_in, double* scale_out) {
    SentinelTableau t;
    BaseGraph b = build_base_graph(g, t);
    reassign_sentinels(b, t, 5, 6);
    SentinelTableau dummy = t;
    dummy.src_sentinel = 7;
    dummy.snk_sentinel = 8;
    ScoreFunction sf;
    sf.anchor_score_function = (ScoreFunction::AnchorScore)cp->anchor_score_function;
    sf.pair_count_power = cp->pair_count_power;
    sf.length_intercept = cp->length_intercept;
    sf.length_decay_power = cp->length_decay_power;
    PathMatchFinder mf(sf);
    mf.max_count = max_count;
    std::vector<match_set_t> matches = mf.find_matches(b, b, t, dummy);
    std::vector<match_set_t> diagonal;
    for (const auto& ms : matches)
        for (const auto& walk : ms.walks1) {
            diagonal.emplace_back();
            auto& m = diagonal.back();
            m.walks1.emplace_back(walk);
            m.walks2.emplace_back(walk);
            m.count1 = ms.count1;
            m.count2 = ms.count2;
            m.full_length = ms.full_length;
        }
    OpenAnchorer an(sf);
    for (int i = 0; i < 3; ++i) { an.gap_open[i] = cp->gap_open[i]; an.gap_extend[i] = cp->gap_extend[i]; }
    an.global_anchoring = global_anchoring != 0;
    an.max_num_match_pairs = max_num_match_pairs;
    an.do_fill_in_anchoring = fill_in != 0;
    an.chaining_algorithm = Anchorer::SparseAffine;
    ChainMerge cm(b, t);
    *scale_out = an.estimate_score_scale(diagonal, b, b, t, t, cm, cm, false, nullptr, nullptr);
    return 0;
}

/* flatten a BaseGraph the way ref_fuse documents */
static void flatten_graph(const BaseGraph& b1, void** out, uint64_t* sizes) {
    const uint64_t n = b1.node_size();
    uint64_t edges = 0, pn = 0;
    for (uint64_t v = 0; v < n; ++v) edges += b1.next_size(v);
    for (uint64_t p = 0; p < b1.path_size(); ++p) pn += b1.path(p).size();
    uint8_t* label = (uint8_t*)malloc(n ? n : 1);
    uint64_t* next_off = (uint64_t*)malloc((n + 1) * 8); uint32_t* next_idx = (uint32_t*)malloc((edges ? edges : 1) * 4);
    uint64_t* prev_off = (uint64_t*)malloc((n + 1) * 8); uint32_t* prev_idx = (uint32_t*)malloc((edges ? edges : 1) * 4);
    uint64_t* path_off = (uint64_t*)malloc((b1.path_size() + 1) * 8); uint32_t* path_nodes = (uint32_t*)malloc((pn ? pn : 1) * 4);
    uint64_t a = 0, b = 0, c = 0;
    next_off[0] = prev_off[0] = path_off[0] = 0;
    for (uint64_t v = 0; v < n; ++v) {
        label[v] = (uint8_t)b1.label(v);
        for (auto w : b1.next(v)) next_idx[a++] = (uint32_t)w;
        next_off[v + 1] = a;
        for (auto w : b1.previous(v)) prev_idx[b++] = (uint32_t)w;
        prev_off[v + 1] = b;
    }
    for (uint64_t p = 0; p < b1.path_size(); ++p) {
        for (auto v : b1.path(p)) path_nodes[c++] = (uint32_t)v;
        path_off[p + 1] = c;
    }
    out[0] = label; out[1] = next_off; out[2] = next_idx; out[3] = prev_off; out[4] = prev_idx; out[5] = path_off; out[6] = path_nodes;
    sizes[0] = n; sizes[1] = edges; sizes[2] = b1.path_size(); sizes[3] = pn;
}

/* make_base_graph + add_sentinels(graph, 5, 6) as Execution builds a leaf subproblem (src/execution.cpp:66-73); ids_out = src, snk */
int ref_leaf_graph(const char* sequence, uint64_t n, void** out, uint64_t* sizes, uint64_t* ids_out) {
    BaseGraph g = make_base_graph("leaf", std::string(sequence, sequence + n));
    SentinelTableau t = add_sentinels(g, 5, 6);
    flatten_graph(g, out, sizes);
    ids_out[0] = t.src_id; ids_out[1] = t.snk_id;
    return 0;
}

static char* dup_text(const std::string& s) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = '\0';
    return p;
}

/* write_gfa (gfa.hpp:46-157) of a flat graph whose paths get the given names */
int ref_write_gfa(const cl_base_graph* g, const char* const* names, int decode, char** text_out, uint64_t* len_out) {
    BaseGraph bg;
    for (uint64_t v = 0; v < g->n_nodes; ++v) bg.add_node((char)g->label[v]);
    clo_graph cg;
    cg.n = g->n_nodes; cg.label = g->label; cg.prev_off = g->prev_off; cg.prev_idx = g->prev_idx;
    cg.next_off = g->next_off; cg.next_idx = g->next_idx; cg.n_src = cg.n_snk = 0; cg.src = cg.snk = nullptr;
    BaseGraph b = build_graph(&cg);
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        auto id = b.add_path(names[p]);
        for (uint64_t i = g->path_off[p]; i < g->path_off[p + 1]; ++i) b.extend_path(id, g->path_nodes[i]);
    }
    SentinelTableau t;
    t.src_id = g->src_id; t.snk_id = g->snk_id;
    std::stringstream ss;
    write_gfa(b, t, ss, decode != 0);
    *text_out = dup_text(ss.str());
    *len_out = ss.str().size();
    return 0;
}

/* explicit_cigar(alignment, graph1, graph2) (alignment.hpp:2804-2843) */
int ref_explicit_cigar(const cl_base_graph* g1, const cl_base_graph* g2, const uint64_t* pairs, uint64_t n_pairs, char** text_out) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    Alignment aln;
    for (uint64_t i = 0; i < n_pairs; ++i) aln.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
    *text_out = dup_text(explicit_cigar(aln, b1, b2));
    return 0;
}

/* fuse (fuse.hpp:46-152) on flat inputs: graph 2 merged into graph 1 along the alignment; the fused graph flattened with its
 * adjacency lists in BaseGraph order.  out[]: label (u8), next_off, next_idx (u32), prev_off, prev_idx (u32), path_off,
 * path_nodes (u32) — malloc'ed, release with ref_free; sizes[]: nodes, edges, paths, path nodes. */
int ref_fuse(const cl_base_graph* g1, const cl_base_graph* g2, const uint64_t* pairs, uint64_t n_pairs, void** out, uint64_t* sizes) {
    SentinelTableau t1, t2;
    BaseGraph b1 = build_base_graph(g1, t1), b2 = build_base_graph(g2, t2);
    Alignment aln;
    for (uint64_t i = 0; i < n_pairs; ++i) aln.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
    fuse(b1, b2, t1, t2, aln);
    flatten_graph(b1, out, sizes);
    return 0;
}

/* Stitcher::despecify_indel_breakpoints (src/stitcher.cpp:265-310) on parallel arrays; same contract as
 * cl_despecify_indel_breakpoints */
/* internal_fuse (fuse.hpp:144-247) of one graph along one alignment; trans_out[old node] = new node */
int ref_internal_fuse(const cl_base_graph* g, const uint64_t* pairs, uint64_t n_pairs, void** out, uint64_t* sizes, uint64_t* ids_out, uint64_t* trans_out) {
    SentinelTableau t, t_out;
    BaseGraph b = build_base_graph(g, t);
    std::vector<Alignment> alns(1);
    for (uint64_t i = 0; i < n_pairs; ++i) alns[0].emplace_back(pairs[2 * i], pairs[2 * i + 1]);
    // an identity "alignment" over all nodes reports the translation
    Alignment ident, ident_out;
    for (uint64_t v = 0; v < b.node_size(); ++v) ident.emplace_back(v, v);
    BaseGraph fused = internal_fuse(b, alns, &t, &t_out, &ident, &ident_out);
    for (uint64_t v = 0; v < b.node_size(); ++v) trans_out[v] = ident_out[v].node_id1;
    flatten_graph(fused, out, sizes);
    ids_out[0] = t_out.src_id;
    ids_out[1] = t_out.snk_id;
    return 0;
}

/* simplify_bubbles (src/modify_graph.cpp:165-382) on a flat graph with sentinels (cyclic graphs welcome) */
int ref_simplify_bubbles(const cl_base_graph* g, void** out, uint64_t* sizes, uint64_t* ids_out) {
    try {
        SentinelTableau t;
        BaseGraph b = build_base_graph(g, t);
        simplify_bubbles(b, t);
        flatten_graph(b, out, sizes);
        ids_out[0] = t.src_id;
        ids_out[1] = t.snk_id;
        return 0;
    } catch (std::exception& ex) {
        fprintf(stderr, "ref_simplify_bubbles: %s\n", ex.what());
        return -1;
    }
}

/* InconsistencyIdentifier::identify_inconsistencies with the CLI's settings (src/parameters.cpp:98-103); *bounds_out malloc'ed [2 * *n_out] */
int ref_inconsistencies(const cl_base_graph* g, const uint64_t* settings /* [6] as cl_polish_params */, uint64_t** bounds_out, uint64_t* n_out) {
    try {
        SentinelTableau t;
        BaseGraph b = build_base_graph(g, t);
        InconsistencyIdentifier ii;
        ii.max_tight_cycle_size = settings[0];
        ii.max_bond_inconsistency_window = settings[1];
        ii.min_inconsistency_disjoint_length = settings[2];
        ii.min_inconsistency_total_length = settings[3];
        ii.padding_target_min_length = settings[4];
        ii.padding_max_length_limit = settings[5];
        auto inc = ii.identify_inconsistencies(b, t);
        *n_out = inc.size();
        *bounds_out = (uint64_t*)malloc((inc.size() ? inc.size() : 1) * 2 * sizeof(uint64_t));
        for (size_t i = 0; i < inc.size(); ++i) { (*bounds_out)[2 * i] = inc[i].first; (*bounds_out)[2 * i + 1] = inc[i].second; }
        return 0;
    } catch (std::exception& ex) {
        fprintf(stderr, "ref_inconsistencies: %s\n", ex.what());
        return -1;
    }
}

/* explicit_cigar(induced_pairwise_alignment(graph, p1, p2), seq1, seq2): the -A output (src/core.cpp:546-550) */
int ref_induced_pairwise_cigar(const cl_base_graph* g, uint64_t p1, uint64_t p2, char** text_out) {
    SentinelTableau t;
    BaseGraph b = build_base_graph(g, t);
    std::string text = explicit_cigar(induced_pairwise_alignment(b, p1, p2), path_to_string(b, b.path(p1)), path_to_string(b, b.path(p2)));
    *text_out = (char*)malloc(text.size() + 1);
    memcpy(*text_out, text.c_str(), text.size() + 1);
    return 0;
}

int ref_despecify(uint64_t n, const double* score, int64_t* gap_before, double* gap_score_before, int64_t* gap_after,
                  double* gap_score_after, int64_t min_len, double prop, uint8_t* keep_out, uint64_t* n_kept_out) {
    std::vector<anchor_t> anchors(n);
    for (uint64_t i = 0; i < n; ++i) {
        anchors[i].score = score[i];
        anchors[i].gap_before = gap_before[i];
        anchors[i].gap_score_before = gap_score_before[i];
        anchors[i].gap_after = gap_after[i];
        anchors[i].gap_score_after = gap_score_after[i];
        anchors[i].idx1 = i;  /* tag to recover which anchors survive */
    }
    Stitcher st;
    st.min_indel_fuzz_length = min_len;
    st.indel_fuzz_score_proportion = prop;
    st.despecify_indel_breakpoints(anchors);
    for (uint64_t i = 0; i < n; ++i) keep_out[i] = 0;
    for (size_t i = 0; i < anchors.size(); ++i) {
        keep_out[anchors[i].idx1] = 1;
        gap_before[i] = anchors[i].gap_before;
        gap_score_before[i] = anchors[i].gap_score_before;
        gap_after[i] = anchors[i].gap_after;
        gap_score_after[i] = anchors[i].gap_score_after;
    }
    *n_kept_out = anchors.size();
    return 0;
}

int ref_po_poa(const clo_graph* g1, const clo_graph* g2, int npw, const cl_align_params* prm, uint64_t* pairs_out,
               uint64_t* n_pairs_out, int64_t* score_out) {
    BaseGraph b1 = build_graph(g1), b2 = build_graph(g2);
    auto s1 = to_vec(g1->src, g1->n_src), s2 = to_vec(g2->src, g2->n_src);
    auto k1 = to_vec(g1->snk, g1->n_snk), k2 = to_vec(g2->snk, g2->n_snk);
    Alignment aln;
    int64_t score = 0;
    if (npw == 1) aln = po_poa(b1, b2, s1, s2, k1, k2, make_params<1>(prm), &score);
    else if (npw == 2) aln = po_poa(b1, b2, s1, s2, k1, k2, make_params<2>(prm), &score);
    else if (npw == 3) aln = po_poa(b1, b2, s1, s2, k1, k2, make_params<3>(prm), &score);
    else return CL_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < aln.size(); ++i) { pairs_out[2 * i] = aln[i].node_id1; pairs_out[2 * i + 1] = aln[i].node_id2; }
    *n_pairs_out = aln.size();
    if (score_out) *score_out = score;
    return 0;
}

/* Stitcher::subalign (or po_poa when force_num_pw is given) over a flat batch; same output contract as
 * clo_stitch_batch.  seconds_out (optional) receives the time spent inside the reference calls only. */
int ref_stitch_batch(const cl_stitch_batch* batch, const cl_stitch_params* sp, const uint8_t* force_num_pw,
                     cl_stitch_result* out, double* seconds_out) {
    uint64_t n = batch->n_problems;
    memset(out, 0, sizeof(*out));
    out->n_problems = n;
    out->aln_off = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
    out->score = (int64_t*)calloc(n ? n : 1, sizeof(int64_t));
    out->route = (uint8_t*)calloc(n ? n : 1, 1);
    out->num_pw = (uint8_t*)calloc(n ? n : 1, 1);
    DumpStitcher st;
    set_stitcher(st, sp);
    std::vector<uint64_t> pairs;
    double secs = 0;
    for (uint64_t k = 0; k < n; ++k) {
        SubGraphInfo i1 = make_info(&batch->side[0], k), i2 = make_info(&batch->side[1], k);
        Alignment aln;
        auto a = std::chrono::steady_clock::now();
        if (force_num_pw && i1.subgraph.node_size() && i2.subgraph.node_size()) {
            int64_t score = 0;
            int npw = force_num_pw[k];
            if (npw == 1) aln = po_poa(i1.subgraph, i2.subgraph, i1.sources, i2.sources, i1.sinks, i2.sinks, make_params<1>(&sp->alignment_params), &score);
            else if (npw == 2) aln = po_poa(i1.subgraph, i2.subgraph, i1.sources, i2.sources, i1.sinks, i2.sinks, make_params<2>(&sp->alignment_params), &score);
            else aln = po_poa(i1.subgraph, i2.subgraph, i1.sources, i2.sources, i1.sinks, i2.sinks, make_params<3>(&sp->alignment_params), &score);
            translate(aln, i1.back_translation, i2.back_translation);
            out->score[k] = score;
            out->num_pw[k] = (uint8_t)npw;
        } else {
            st.subalign(i1, i2, aln, batch->only_deletion_alns ? batch->only_deletion_alns[k] != 0 : false);
        }
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        for (const auto& ap : aln) { pairs.push_back(ap.node_id1); pairs.push_back(ap.node_id2); }
        out->aln_off[k + 1] = pairs.size() / 2;
    }
    out->pairs = (uint64_t*)malloc((pairs.size() ? pairs.size() : 1) * sizeof(uint64_t));
    if (!pairs.empty()) memcpy(out->pairs, pairs.data(), pairs.size() * sizeof(uint64_t));
    if (seconds_out) *seconds_out = secs;
    return 0;
}

/* Stitcher::internal_stitch (stitcher.hpp:209-234) on a flat anchor chain inside one graph. */
int ref_internal_stitch(const cl_base_graph* g, uint64_t n_anchors, const uint64_t* walk_off, const uint32_t* walk1, const uint32_t* walk2,
                        const cl_stitch_params* sp, uint64_t** pairs_out, uint64_t* n_pairs_out) {
    SentinelTableau t;
    BaseGraph b = build_base_graph(g, t);
    PathMerge<> pm(b, t);
    DumpStitcher st;
    set_stitcher(st, sp);
    std::vector<anchor_t> chain(n_anchors);
    for (uint64_t a = 0; a < n_anchors; ++a) {
        chain[a].walk1.assign(walk1 + walk_off[a], walk1 + walk_off[a + 1]);
        chain[a].walk2.assign(walk2 + walk_off[a], walk2 + walk_off[a + 1]);
    }
    Alignment aln = st.internal_stitch(chain, b, pm);
    *n_pairs_out = aln.size();
    *pairs_out = (uint64_t*)malloc((aln.size() ? aln.size() : 1) * 2 * sizeof(uint64_t));
    for (size_t i = 0; i < aln.size(); ++i) { (*pairs_out)[2 * i] = aln[i].node_id1; (*pairs_out)[2 * i + 1] = aln[i].node_id2; }
    return 0;
}

void ref_result_free(cl_stitch_result* r) {
    if (!r) return;
    free(r->aln_off); free(r->pairs); free(r->score); free(r->route); free(r->num_pw);
    memset(r, 0, sizeof(*r));
}

/* FASTA (+ optional Newick file) -> CIGAR (2 sequences) or GFA, with the CLI's parameters
 * (Parameters defaults, src/parameters.cpp:22-108; src/main.cpp:239-301).  dump_path may be NULL.
 * skip_calibration mirrors --skip-calibration; max_num_match_pairs <= 0 keeps the default.
 * timings_out[8] = calibration, match finding, chaining, partition, extraction, subalign, fuse, total */
/* The plan of a progressive MSA as the reference derives it: Tree(newick) (in_order_newick_string when newick is empty), then
 * Execution's constructor (prune to the FASTA's names, compact, binarize, small_first_postorder; src/execution.cpp:12-92).  Text out:
 * one "L <name>" line per leaf in Execution::leaf_subproblems order (the calibration order), then one "M <child1 leaves>;<child2
 * leaves>" line per merge in execution order, leaves sorted and comma separated.  An exception's message comes back as "E <what>". */
int ref_msa_plan(const char* newick, const char* const* names, uint64_t n_names, char** text_out) {
    std::stringstream ss;
    try {
        std::vector<std::pair<std::string, std::string>> seqs;
        std::vector<std::string> nm;
        for (uint64_t i = 0; i < n_names; ++i) { seqs.emplace_back(names[i], "ACGT"); nm.push_back(names[i]); }
        Tree tree(newick && *newick ? std::string(newick) : in_order_newick_string(nm));
        Execution ex(std::move(seqs), std::move(tree), true);
        for (auto* leaf : ex.leaf_subproblems()) ss << "L " << leaf->name << "\n";
        auto leaves = [&](const Subproblem& sp) {
            auto v = ex.leaf_descendents(sp);
            std::sort(v.begin(), v.end());
            std::string r;
            for (size_t i = 0; i < v.size(); ++i) r += (i ? "," : "") + v[i];
            return r;
        };
        while (!ex.finished()) {
            auto t = ex.next();
            ss << "M " << leaves(*std::get<1>(t)) << ";" << leaves(*std::get<2>(t)) << "\n";
        }
    } catch (std::exception& e) {
        ss.str("");
        ss << "E " << e.what() << "\n";
    }
    const std::string text = ss.str();
    *text_out = (char*)malloc(text.size() + 1);
    memcpy(*text_out, text.c_str(), text.size() + 1);
    return 0;
}

int ref_msa_dump(const char* fasta_path, const char* newick_path, const char* dump_path, const char* out_path,
                 int skip_calibration, long long max_num_match_pairs, int verbosity, double* timings_out) {
    try {
        auto T0 = std::chrono::steady_clock::now();
        Parameters params;
        params.set<std::string>("fasta_name", fasta_path);
        if (skip_calibration) params.set<bool>("skip_calibration", true);
        if (max_num_match_pairs > 0) params.set<int64_t>("max_num_match_pairs", (int64_t)max_num_match_pairs);
        params.validate();
        logging::level = (logging::LoggingLevel)verbosity;
        std::ifstream fin(fasta_path);
        if (!fin) return -100;
        auto parsed = parse_fasta(fin);
        std::vector<std::string> names;
        for (const auto& p : parsed) names.push_back(p.first);
        std::string newick;
        if (newick_path && *newick_path) {
            std::ifstream tin(newick_path);
            std::stringstream ss;
            ss << tin.rdbuf();
            newick = ss.str();
        } else {
            newick = in_order_newick_string(names);
        }
        Tree tree(newick);
        DumpCore core(std::move(parsed), std::move(tree));
        if (names.size() == 2) params.set<bool>("preserve_subproblems", true);
        params.apply(core);
        core.preserve_subproblems = true;
        Dump dump;
        Dump* dp = nullptr;
        if (dump_path && *dump_path) {
            if (!dump.open(dump_path)) return -101;
            dp = &dump;
        }
        core.run(dp);
        std::string text;
        if (names.size() == 2) {
            const auto& root = core.root_subproblem();
            text = explicit_cigar(root.alignment, core.leaf_subproblem(names.front()).graph, core.leaf_subproblem(names.back()).graph) + "\n";
        } else {
            std::stringstream ss;
            const auto& root = core.root_subproblem();
            write_gfa(root.graph, root.tableau, ss);
            text = ss.str();
        }
        if (dp) { dump.str("output", text); dump.close(); }
        if (out_path && *out_path) {
            std::ofstream fo(out_path);
            fo << text;
        }
        if (timings_out) {
            timings_out[0] = core.t_calib; timings_out[1] = core.t_match; timings_out[2] = core.t_chain;
            timings_out[3] = core.t_partition; timings_out[4] = core.t_extract; timings_out[5] = core.t_subalign;
            timings_out[6] = core.t_fuse;
            timings_out[7] = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
        }
        return 0;
    } catch (std::exception& ex) {
        fprintf(stderr, "ref_msa_dump: %s\n", ex.what());
        return -102;
    }
}

}  // extern "C"
