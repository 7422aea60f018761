// seam_wrappers.hpp — the reference's own call signatures at seams S1 and S3, implemented on the C ABI of
// include/centrolign_amd.h, so that code shaped like include/centrolign/core.hpp / stitcher.hpp compiles against them unchanged:
//
//   S1  po_poa<NumPW>(graph1, graph2, sources1, sources2, sinks1, sinks2, params, score_out)          include/centrolign/alignment.hpp:78-85
//   S2' Stitcher::internal_stitch(anchor_chain, graph, xmerge)                                         include/centrolign/stitcher.hpp:41-43
//   S3  Anchorer::anchor_chain(matches, graph1, graph2, tableau1, tableau2, xmerge1, xmerge2,
//                              restrain_memory, masked_matches, override_scale)                        include/centrolign/anchorer.hpp:135-145
//       (here the Anchorer and its ScoreFunction come first, as arguments)
//
// Header-only C++11, written against the reference's concepts (see stitch_adapter.hpp / core_adapter.hpp), not its headers.  The
// XMerge arguments are accepted and ignored: the library builds its own PathMerge tables from the graphs (the reference's XMerge is a
// cache of the same reachability information).  The device is passed first (the reference has no such object); everything else is
// positional as in the reference.  Errors throw std::runtime_error, as the reference's own code does (src/stitcher.cpp:36).
#ifndef CENTROLIGN_AMD_SEAM_WRAPPERS_HPP
#define CENTROLIGN_AMD_SEAM_WRAPPERS_HPP

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_set>
#include <utility>
#include <unordered_map>
#include <vector>

#include "../centrolign_amd.h"
#include "core_adapter.hpp"
#include "stitch_adapter.hpp"

namespace centrolign_amd {

// ---- S1 ------------------------------------------------------------------------------------------------------------------
// AlignmentT = std::vector<AlignedPair>; ParamsT = AlignmentParameters<NumPW> (.match, .mismatch, .gap_open[k], .gap_extend[k])
template <int NumPW, class AlignmentT, class Graph, class ParamsT>
AlignmentT po_poa(Device& dev, const Graph& graph1, const Graph& graph2, const std::vector<uint64_t>& sources1,
                  const std::vector<uint64_t>& sources2, const std::vector<uint64_t>& sinks1, const std::vector<uint64_t>& sinks2,
                  const ParamsT& params, int64_t* score_out = nullptr) {
    static_assert(NumPW >= 1 && NumPW <= 3, "NumPW must be 1, 2 or 3");
    // one subproblem, identity back-translation: the ids that come back are the graphs' own
    struct Info {
        const Graph& subgraph;
        std::vector<uint64_t> back_translation;
        const std::vector<uint64_t>& sources;
        const std::vector<uint64_t>& sinks;
    };
    auto identity = [](uint64_t n) { std::vector<uint64_t> v(n); for (uint64_t i = 0; i < n; ++i) v[i] = i; return v; };
    Info i1{graph1, identity(graph1.node_size()), sources1, sinks1}, i2{graph2, identity(graph2.node_size()), sources2, sinks2};
    StitchBatchBuilder batch;
    batch.add(i1, i2, false);
    cl_align_params ap;
    ap.match = params.match;
    ap.mismatch = params.mismatch;
    for (int k = 0; k < 3; ++k) {   // the unused pieces repeat the last one (truncate_parameters reads only the leading NumPW)
        const int j = k < NumPW ? k : NumPW - 1;
        ap.gap_open[k] = params.gap_open[j];
        ap.gap_extend[k] = params.gap_extend[j];
    }
    const uint8_t num_pw = (uint8_t)NumPW;
    cl_stitch_batch b = batch.view();
    cl_stitch_result r;
    if (int rc = cl_po_poa_batch(dev.get(), &b, &num_pw, &ap, &r))
        throw std::runtime_error(std::string("cl_po_poa_batch failed (") + std::to_string(rc) + "): " + cl_last_error(dev.get()));
    AlignmentT out;
    out.reserve(r.aln_off[1]);
    for (uint64_t i = 0; i < r.aln_off[1]; ++i) out.emplace_back(r.pairs[2 * i], r.pairs[2 * i + 1]);
    if (score_out) *score_out = r.score[0];
    cl_stitch_result_free(&r);
    return out;
}

// ---- S3 ------------------------------------------------------------------------------------------------------------------
// AnchorerT: the public tunables of Anchorer (anchorer.hpp:152-175); ScoreFunctionT: the ScoreFunction the Anchorer was constructed on (its
// pointer to it is protected, :179; Core holds the object, core.hpp:84); AnchorT = anchor_t; MaskT = the reference's
// std::unordered_set<std::tuple<size_t, size_t, size_t>> (its hash comes from the reference's utility.hpp).
// `matches` is reordered in place and grown by the branch splitting exactly as the reference does it (:971-973, :1108-1173); the mask,
// when given, is re-indexed to the new order in place (:1159-1166).
template <class AnchorT, class AnchorerT, class ScoreFunctionT, class MatchSetT, class BGraph, class TableauT, class XMerge,
          class MaskT = std::unordered_set<std::tuple<size_t, size_t, size_t>>>
std::vector<AnchorT> anchor_chain(Device& dev, const AnchorerT& anchorer, const ScoreFunctionT& score_function, std::vector<MatchSetT>& matches,
                                  const BGraph& graph1, const BGraph& graph2, const TableauT& tableau1, const TableauT& tableau2,
                                  const XMerge& /*xmerge1*/, const XMerge& /*xmerge2*/, bool /*restrain_memory*/,
                                  MaskT* masked_matches = nullptr, double* override_scale = nullptr) {
    FlatBaseGraph g1(graph1, tableau1), g2(graph2, tableau2);
    cl_anchor_params ap{};
    cl_chain_params_default(&ap.chain);
    ap.chaining_algorithm_plus_one = (int)anchorer.chaining_algorithm + 1;   // (Sparse: the library builds ChainMerge tables itself, as core.hpp:350-357 passes them)
    for (int k = 0; k < 3; ++k) { ap.chain.gap_open[k] = anchorer.gap_open[k]; ap.chain.gap_extend[k] = anchorer.gap_extend[k]; }
    ap.chain.anchor_score_function = (int)score_function.anchor_score_function;
    ap.chain.pair_count_power = score_function.pair_count_power;
    ap.chain.length_intercept = score_function.length_intercept;
    ap.chain.length_decay_power = score_function.length_decay_power;
    ap.chain.global_anchoring = anchorer.global_anchoring ? 1 : 0;
    ap.max_num_match_pairs = anchorer.max_num_match_pairs;
    ap.score_scale = score_function.score_scale;
    ap.autocalibrate_gap_penalties = anchorer.autocalibrate_gap_penalties ? 1 : 0;
    ap.do_fill_in_anchoring = anchorer.do_fill_in_anchoring ? 1 : 0;

    auto fail = [&](const char* what, int rc) { throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + cl_last_error(dev.get())); };
    if (anchorer.split_matches_at_branchpoints) {
        cl_split_params sp;
        sp.anchor_split_limit = anchorer.anchor_split_limit;
        sp.min_split_length = anchorer.min_split_length;
        sp.min_path_length_spread = anchorer.min_path_length_spread;
        sp.max_split_match_set_size = anchorer.max_split_match_set_size;
        FlatMatchSets before(matches);
        cl_owned_match_sets* owned = nullptr;
        if (int rc = cl_split_branching_matches(&g1.view, &g2.view, &before.view, &sp, &owned)) fail("cl_split_branching_matches", rc);
        cl_match_sets v;
        cl_owned_match_sets_view(owned, &v);
        std::vector<MatchSetT> split(v.n_sets);
        for (uint64_t s = 0; s < v.n_sets; ++s) {
            MatchSetT& m = split[s];
            for (uint64_t w = v.set_off1[s]; w < v.set_off1[s + 1]; ++w) m.walks1.emplace_back(v.nodes1 + v.walk_off1[w], v.nodes1 + v.walk_off1[w + 1]);
            for (uint64_t w = v.set_off2[s]; w < v.set_off2[s + 1]; ++w) m.walks2.emplace_back(v.nodes2 + v.walk_off2[w], v.nodes2 + v.walk_off2[w + 1]);
            m.count1 = v.count1[s];
            m.count2 = v.count2[s];
            m.full_length = v.full_length[s];
        }
        cl_owned_match_sets_free(owned);
        // the mask follows the new sets (anchorer.hpp:816-820, 911-918): every piece cut off set i is appended behind the original sets and
        // inherits set i's masked (idx1, idx2) pairs; set i itself keeps only its first piece.  The pieces are appended in the order of the
        // sets they come from, and the pieces of one set add up to what it lost, which identifies every appended set's origin.
        if (masked_matches && !masked_matches->empty() && split.size() != matches.size()) {
            std::unordered_map<size_t, std::vector<std::pair<size_t, size_t>>> by_set;
            for (const auto& m : *masked_matches) by_set[std::get<0>(m)].emplace_back(std::get<1>(m), std::get<2>(m));
            size_t next = matches.size();
            for (size_t i = 0; i < matches.size(); ++i) {
                if (matches[i].walks1.empty()) continue;
                size_t lost = matches[i].walks1.front().size() - split[i].walks1.front().size();
                while (lost > 0 && next < split.size()) {
                    auto it = by_set.find(i);
                    if (it != by_set.end())
                        for (const auto& ab : it->second) masked_matches->emplace(next, ab.first, ab.second);
                    lost -= split[next].walks1.front().size();
                    ++next;
                }
            }
        }
        matches.swap(split);
    }
    FlatMatchSets ms(matches);
    std::vector<uint64_t> mask;
    if (masked_matches)
        for (const auto& m : *masked_matches) { mask.push_back(std::get<0>(m)); mask.push_back(std::get<1>(m)); mask.push_back(std::get<2>(m)); }
    cl_anchor_chain_result r;
    const int rc = masked_matches || override_scale
                       ? cl_anchor_chain_masked(dev.get(), &g1.view, &g2.view, &ms.view, &ap, mask.data(), mask.size() / 3, override_scale, &r)
                       : cl_anchor_chain(dev.get(), &g1.view, &g2.view, &ms.view, &ap, &r);
    if (rc) fail("cl_anchor_chain", rc);
    // the reference leaves `matches` in the order of its budgeted selection (anchorer.hpp:1168) and the mask re-indexed to it
    {
        std::vector<MatchSetT> reordered(matches.size());
        std::vector<uint64_t> position(matches.size());
        for (uint64_t k = 0; k < r.n_sets; ++k) { reordered[k] = std::move(matches[r.set_order[k]]); position[r.set_order[k]] = k; }
        matches.swap(reordered);
        if (masked_matches) {
            MaskT re;
            re.reserve(masked_matches->size());
            for (const auto& m : *masked_matches) re.emplace(position[std::get<0>(m)], std::get<1>(m), std::get<2>(m));
            masked_matches->swap(re);
        }
    }
    std::vector<AnchorT> chain(r.n_anchors);
    for (uint64_t i = 0; i < r.n_anchors; ++i) {
        AnchorT& a = chain[i];
        a.walk1.assign(r.walk1 + r.walk_off[i], r.walk1 + r.walk_off[i + 1]);
        a.walk2.assign(r.walk2 + r.walk_off[i], r.walk2 + r.walk_off[i + 1]);
        a.count1 = r.count1[i];
        a.count2 = r.count2[i];
        a.full_length = r.full_length[i];
        a.match_set = r.anchors[3 * i];
        a.idx1 = r.anchors[3 * i + 1];
        a.idx2 = r.anchors[3 * i + 2];
        a.score = r.score[i];
        a.gap_before = r.gap_before[i];
        a.gap_after = r.gap_after[i];
        a.gap_score_before = r.gap_score_before[i];
        a.gap_score_after = r.gap_score_after[i];
    }
    cl_anchor_chain_result_free(&r);
    return chain;
}

// ---- S2, second entry --------------------------------------------------------------------------------------------------------
// StitcherT: the public tunables of Stitcher (stitcher.hpp:48-71); AnchorT: .walk1, .walk2
template <class AlignmentT, class StitcherT, class AnchorT, class BGraph, class TableauT, class XMerge>
AlignmentT internal_stitch(Device& dev, const StitcherT& stitcher, const std::vector<AnchorT>& anchor_chain, const BGraph& graph,
                           const TableauT& tableau, const XMerge& /*xmerge*/) {
    FlatBaseGraph g(graph, tableau);
    std::vector<uint64_t> walk_off{0};
    std::vector<uint32_t> w1, w2;
    for (const auto& a : anchor_chain) {
        for (auto v : a.walk1) w1.push_back((uint32_t)v);
        for (auto v : a.walk2) w2.push_back((uint32_t)v);
        walk_off.push_back(w1.size());
    }
    const cl_stitch_params sp = stitch_params_of(stitcher);
    cl_alignment out;
    if (int rc = cl_internal_stitch(dev.get(), &g.view, anchor_chain.size(), walk_off.data(), w1.data(), w2.data(), &sp, &out))
        throw std::runtime_error(std::string("cl_internal_stitch failed (") + std::to_string(rc) + "): " + cl_last_error(dev.get()));
    AlignmentT aln;
    aln.reserve(out.n_pairs);
    for (uint64_t i = 0; i < out.n_pairs; ++i) aln.emplace_back(out.pairs[2 * i], out.pairs[2 * i + 1]);
    cl_alignment_free(&out);
    return aln;
}

}  // namespace centrolign_amd

#endif
