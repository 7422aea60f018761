// core_adapter.hpp — header-only C++11 glue between centrolign's own types and cl_core_align (include/centrolign_amd.h):
// what a centrolign maintainer includes to run the body of Core::align (include/centrolign/core.hpp:181-252) — anchor
// chain, partition, despecify, stitch — through the MI355X library; see INTEGRATION.md.
//
// Written against the reference's concepts, not its headers:
//   BaseGraphT  : node_size(), label(id), next(id), previous(id), path_size(), path(id)      (include/centrolign/graph.hpp:96-151)
//   TableauT    : .src_id, .snk_id                                                           (include/centrolign/modify_graph.hpp:33-38)
//   MatchSetT   : .walks1, .walks2 (vector<vector<uint64_t>>), .count1, .count2, .full_length (include/centrolign/match_finder.hpp:21-34)
//   CoreT       : .anchorer, .partitioner, .stitcher, .score_function with their public tunables (include/centrolign/core.hpp:70-103)
//   MatchFinderT: .max_count, .use_color_set_size                                             (include/centrolign/match_finder.hpp:52-54)
#ifndef CENTROLIGN_AMD_CORE_ADAPTER_HPP
#define CENTROLIGN_AMD_CORE_ADAPTER_HPP

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../centrolign_amd.h"
#include "stitch_adapter.hpp"

namespace centrolign_amd {

// owns the flat arrays a cl_base_graph points into
struct FlatBaseGraph {
    std::vector<uint8_t> label;
    std::vector<uint64_t> next_off{0}, prev_off{0}, path_off{0};
    std::vector<uint32_t> next_idx, prev_idx, path_nodes;
    cl_base_graph view;

    template <class BaseGraphT, class TableauT>
    FlatBaseGraph(const BaseGraphT& g, const TableauT& t) {
        const uint64_t n = g.node_size();
        label.resize(n);
        for (uint64_t v = 0; v < n; ++v) {
            label[v] = (uint8_t)g.label(v);
            for (auto w : g.next(v)) next_idx.push_back((uint32_t)w);
            next_off.push_back(next_idx.size());
            for (auto w : g.previous(v)) prev_idx.push_back((uint32_t)w);
            prev_off.push_back(prev_idx.size());
        }
        for (uint64_t p = 0; p < g.path_size(); ++p) {
            for (auto v : g.path(p)) path_nodes.push_back((uint32_t)v);
            path_off.push_back(path_nodes.size());
        }
        view.n_nodes = n;
        view.label = label.data();
        view.next_off = next_off.data(); view.next_idx = next_idx.data();
        view.prev_off = prev_off.data(); view.prev_idx = prev_idx.data();
        view.n_paths = g.path_size();
        view.path_off = path_off.data(); view.path_nodes = path_nodes.data();
        view.src_id = t.src_id;
        view.snk_id = t.snk_id;
    }
    FlatBaseGraph(const FlatBaseGraph&) = delete;
    FlatBaseGraph& operator=(const FlatBaseGraph&) = delete;
};

// owns the flat arrays a cl_match_sets points into
struct FlatMatchSets {
    std::vector<uint64_t> set_off1{0}, walk_off1{0}, set_off2{0}, walk_off2{0}, count1, count2, full_length;
    std::vector<uint32_t> nodes1, nodes2;
    cl_match_sets view;

    template <class MatchSetT>
    explicit FlatMatchSets(const std::vector<MatchSetT>& matches) {
        for (const auto& ms : matches) {
            for (const auto& w : ms.walks1) { for (auto v : w) nodes1.push_back((uint32_t)v); walk_off1.push_back(nodes1.size()); }
            set_off1.push_back(walk_off1.size() - 1);
            for (const auto& w : ms.walks2) { for (auto v : w) nodes2.push_back((uint32_t)v); walk_off2.push_back(nodes2.size()); }
            set_off2.push_back(walk_off2.size() - 1);
            count1.push_back(ms.count1);
            count2.push_back(ms.count2);
            full_length.push_back(ms.full_length);
        }
        view.n_sets = matches.size();
        view.set_off1 = set_off1.data(); view.walk_off1 = walk_off1.data(); view.nodes1 = nodes1.data();
        view.set_off2 = set_off2.data(); view.walk_off2 = walk_off2.data(); view.nodes2 = nodes2.data();
        view.count1 = count1.data(); view.count2 = count2.data(); view.full_length = full_length.data();
    }
    FlatMatchSets(const FlatMatchSets&) = delete;
    FlatMatchSets& operator=(const FlatMatchSets&) = delete;
};

// the tunables Core::align reads, from a configured Core
template <class CoreT>
cl_core_align_params core_align_params_of(const CoreT& core, bool is_main_execution) {
    cl_core_align_params p;
    cl_core_align_params_default(&p);
    const auto& an = core.anchorer;
    p.split_matches_at_branchpoints = an.split_matches_at_branchpoints ? 1 : 0;
    p.split.anchor_split_limit = an.anchor_split_limit;
    p.split.min_split_length = an.min_split_length;
    p.split.min_path_length_spread = an.min_path_length_spread;
    p.split.max_split_match_set_size = an.max_split_match_set_size;
    for (int k = 0; k < 3; ++k) { p.anchor.chain.gap_open[k] = an.gap_open[k]; p.anchor.chain.gap_extend[k] = an.gap_extend[k]; }
    p.anchor.chain.anchor_score_function = (int)core.score_function.anchor_score_function;
    p.anchor.chain.pair_count_power = core.score_function.pair_count_power;
    p.anchor.chain.length_intercept = core.score_function.length_intercept;
    p.anchor.chain.length_decay_power = core.score_function.length_decay_power;
    p.anchor.chain.global_anchoring = an.global_anchoring ? 1 : 0;
    p.anchor.max_num_match_pairs = an.max_num_match_pairs;
    p.anchor.score_scale = core.score_function.score_scale;
    p.anchor.autocalibrate_gap_penalties = an.autocalibrate_gap_penalties ? 1 : 0;
    p.anchor.do_fill_in_anchoring = an.do_fill_in_anchoring ? 1 : 0;
    const auto& pt = core.partitioner;
    p.partition.constraint_method = (int)pt.constraint_method;
    p.partition.minimum_segment_score = pt.minimum_segment_score;
    p.partition.minimum_segment_average = pt.minimum_segment_average;
    p.partition.window_length = pt.window_length;
    p.partition.generalized_length_mean = pt.generalized_length_mean;
    p.partition.boundary_score_factor = pt.boundary_score_factor;
    p.partition.score_boundaries = is_main_execution ? 0 : 1;
    p.min_indel_fuzz_length = core.stitcher.min_indel_fuzz_length;
    p.indel_fuzz_score_proportion = core.stitcher.indel_fuzz_score_proportion;
    p.stitch = stitch_params_of(core.stitcher);
    return p;
}

// Core::align on the device; throws std::runtime_error with the library's message on failure
template <class AlignedPairT, class BaseGraphT, class TableauT, class MatchSetT>
std::vector<AlignedPairT> core_align(Device& dev, const BaseGraphT& graph1, const TableauT& tableau1, const BaseGraphT& graph2,
                                     const TableauT& tableau2, const std::vector<MatchSetT>& matches, const cl_core_align_params& params) {
    static_assert(sizeof(AlignedPairT) == 2 * sizeof(uint64_t), "AlignedPair must be two uint64_t");
    FlatBaseGraph g1(graph1, tableau1), g2(graph2, tableau2);
    FlatMatchSets ms(matches);
    cl_core_align_result r;
    if (int rc = cl_core_align(dev.get(), &g1.view, &g2.view, &ms.view, &params, &r))
        throw std::runtime_error(std::string("cl_core_align failed (") + std::to_string(rc) + "): " + cl_last_error(dev.get()));
    std::vector<AlignedPairT> out;
    out.reserve(r.alignment.n_pairs);
    for (uint64_t i = 0; i < r.alignment.n_pairs; ++i) out.emplace_back(r.alignment.pairs[2 * i], r.alignment.pairs[2 * i + 1]);
    cl_core_align_result_free(&r);
    return out;
}

// the tunables PathMatchFinder::find_matches reads (match_finder.hpp:52-54) + the ScoreFunction of its weight filter
template <class MatchFinderT, class ScoreFunctionT>
cl_match_params match_params_of(const MatchFinderT& finder, const ScoreFunctionT& score_function) {
    cl_match_params p;
    cl_match_params_default(&p);
    p.max_count = finder.max_count;
    p.use_color_set_size = finder.use_color_set_size ? 1 : 0;
    p.score.anchor_score_function = (int)score_function.anchor_score_function;
    p.score.pair_count_power = score_function.pair_count_power;
    p.score.length_intercept = score_function.length_intercept;
    p.score.length_decay_power = score_function.length_decay_power;
    return p;
}

// PathMatchFinder::find_matches (match_finder.hpp:120-131) through the library: same arguments, same std::vector<match_set_t>
template <class MatchSetT, class BaseGraphT, class TableauT>
std::vector<MatchSetT> find_matches(Device& dev, const BaseGraphT& graph1, const BaseGraphT& graph2, const TableauT& tableau1,
                                    const TableauT& tableau2, const cl_match_params& params) {
    FlatBaseGraph g1(graph1, tableau1), g2(graph2, tableau2);
    // the sentinel characters travel as the labels of the sentinel nodes (Core::do_execution reassigns them before every
    // merge, core.hpp:283-286)
    if (g1.view.n_nodes) { g1.label[tableau1.src_id] = (uint8_t)tableau1.src_sentinel; g1.label[tableau1.snk_id] = (uint8_t)tableau1.snk_sentinel; }
    if (g2.view.n_nodes) { g2.label[tableau2.src_id] = (uint8_t)tableau2.src_sentinel; g2.label[tableau2.snk_id] = (uint8_t)tableau2.snk_sentinel; }
    cl_owned_match_sets* owned = nullptr;
    if (int rc = cl_find_matches(dev.get(), &g1.view, &g2.view, &params, &owned, nullptr))
        throw std::runtime_error(std::string("cl_find_matches failed (") + std::to_string(rc) + "): " + cl_last_error(dev.get()));
    cl_match_sets v;
    cl_owned_match_sets_view(owned, &v);
    std::vector<MatchSetT> out(v.n_sets);
    for (uint64_t s = 0; s < v.n_sets; ++s) {
        MatchSetT& m = out[s];
        for (uint64_t w = v.set_off1[s]; w < v.set_off1[s + 1]; ++w) m.walks1.emplace_back(v.nodes1 + v.walk_off1[w], v.nodes1 + v.walk_off1[w + 1]);
        for (uint64_t w = v.set_off2[s]; w < v.set_off2[s + 1]; ++w) m.walks2.emplace_back(v.nodes2 + v.walk_off2[w], v.nodes2 + v.walk_off2[w + 1]);
        m.count1 = v.count1[s];
        m.count2 = v.count2[s];
        m.full_length = v.full_length[s];
    }
    cl_owned_match_sets_free(owned);
    return out;
}

// fuse (fuse.hpp:46-152) through the library, checked against an already fused reference graph: true when cl_fuse of
// (dest, source, alignment) is node for node, list for list and path for path the graph `fused_by_reference`
template <class BaseGraphT, class TableauT, class AlignmentT>
bool fuse_equals(const BaseGraphT& dest, const BaseGraphT& source, const TableauT& dest_table, const TableauT& source_table,
                 const AlignmentT& alignment, const BaseGraphT& fused_by_reference) {
    FlatBaseGraph g1(dest, dest_table), g2(source, source_table), want(fused_by_reference, dest_table);
    std::vector<uint64_t> pairs;
    pairs.reserve(2 * alignment.size());
    for (const auto& ap : alignment) { pairs.push_back(ap.node_id1); pairs.push_back(ap.node_id2); }
    cl_owned_base_graph* owned = nullptr;
    if (int rc = cl_fuse(&g1.view, &g2.view, pairs.data(), alignment.size(), &owned))
        throw std::runtime_error("cl_fuse failed (" + std::to_string(rc) + ")");
    cl_base_graph v;
    cl_owned_base_graph_view(owned, &v);
    auto same = [](const void* a, const void* b, size_t bytes) { return bytes == 0 || std::memcmp(a, b, bytes) == 0; };
    bool ok = v.n_nodes == want.view.n_nodes && v.n_paths == want.view.n_paths && same(v.label, want.view.label, v.n_nodes) &&
              same(v.next_off, want.view.next_off, (v.n_nodes + 1) * 8) && same(v.prev_off, want.view.prev_off, (v.n_nodes + 1) * 8) &&
              same(v.next_idx, want.view.next_idx, v.next_off[v.n_nodes] * 4) && same(v.prev_idx, want.view.prev_idx, v.prev_off[v.n_nodes] * 4) &&
              same(v.path_off, want.view.path_off, (v.n_paths + 1) * 8) && same(v.path_nodes, want.view.path_nodes, v.path_off[v.n_paths] * 4);
    cl_owned_base_graph_free(owned);
    return ok;
}

}  // namespace centrolign_amd

#endif
