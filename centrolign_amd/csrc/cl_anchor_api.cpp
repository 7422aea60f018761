// cl_anchor_api.cpp — host-side anchor-chain post-processing of the stitch path.
//   cl_despecify_indel_breakpoints <-> Stitcher::despecify_indel_breakpoints (src/stitcher.cpp:265-310) over
//                                      identify_despecification_partition (src/stitcher.cpp:115-263) and
//                                      PartitionClient::traceback (include/centrolign/partition_client.hpp:31-54)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <tuple>
#include <vector>

#include "cl_internal.hpp"
#include "search_trees.hpp"

namespace {

using Val = std::tuple<int64_t, double, size_t>;

// src/stitcher.cpp:115-263
std::vector<std::pair<size_t, size_t>> despecification_partition(uint64_t n, const double* score, const int64_t* gap_before,
                                                                 int64_t min_len, double prop) {
    static const double inf = std::numeric_limits<double>::max();
    static const double mininf = std::numeric_limits<double>::lowest();
    std::vector<std::pair<int64_t, int64_t>> limit(n, {0, 0});
    int64_t prev_indel = -1, before_prev = -1;
    for (size_t i = 0; i < n; ++i) {
        if (i != 0 && std::llabs(gap_before[i]) >= min_len) { before_prev = prev_indel; prev_indel = (int64_t)i; }
        if (before_prev != -1 && prev_indel != -1) {
            limit[i].first = before_prev + 1;
            limit[i].second = std::min<int64_t>((int64_t)i, prev_indel + 1);
        } else if (prev_indel != -1) {
            limit[i].first = std::min<int64_t>(1, (int64_t)i);
            limit[i].second = std::min<int64_t>(prev_indel + 1, (int64_t)i);
        }
    }
    std::vector<double> prefix(n + 1, 0.0);
    for (size_t i = 0; i < n; ++i) prefix[i + 1] = prefix[i] + score[i];
    std::vector<double> index_key(n, mininf);
    for (size_t i = 1; i < n; ++i) index_key[i] = prefix[i] + prop * score[i - 1];
    std::vector<std::tuple<int64_t, double, Val>> data;
    data.reserve(n + 1);
    for (size_t i = 0; i < n; ++i) data.emplace_back((int64_t)i, index_key[i], Val(0, 0.0, 0));
    clhost::OrthoTree<int64_t, double, Val> tree(data);
    std::vector<std::pair<Val, Val>> dp(n + 1, {Val(-1, 0.0, 0), Val(-1, 0.0, 0)});
    std::vector<size_t> back(n + 1, (size_t)-1);
    std::get<0>(dp.front().first) = 0;
    size_t opt = 0;
    for (size_t i = 1; i + 1 < dp.size(); ++i) {
        dp[i].first = std::max(dp[i - 1].first, dp[i - 1].second);
        const double query_key = prefix[i] - prop * score[i];
        const size_t it = tree.range_max(limit[i].first, limit[i].second, query_key, inf);
        if (it != tree.end()) {
            const Val& v = tree.val[it];
            // (the second field is computed from get<0> in the reference, src/stitcher.cpp:221-223; kept as is)
            dp[i].second = Val(std::get<0>(v) + 1, (double)std::get<0>(v) - prefix[i] + prefix[(size_t)tree.key1[it]], i);
            back[i] = (size_t)tree.key1[it];
            if (dp[i].second > dp[opt].second) opt = i;
        }
        const size_t at = tree.find((int64_t)i, index_key[i]);
        tree.update(at, Val(std::get<0>(dp[i].first), std::get<1>(dp[i].first), i));
    }
    // partition_client.hpp:31-54
    std::vector<std::pair<size_t, size_t>> partition;
    bool in_interval = true;
    size_t tb = opt;
    while (tb > 0) {
        if (in_interval) {
            const size_t prev = back[tb];
            partition.emplace_back(prev, tb);
            tb = prev;
            in_interval = false;
        } else {
            in_interval = (dp[tb].first == dp[tb - 1].second);
            --tb;
        }
    }
    std::reverse(partition.begin(), partition.end());
    return partition;
}

}  // namespace

extern "C" {

int cl_despecify_indel_breakpoints(uint64_t n_anchors, const double* score, int64_t* gap_before, double* gap_score_before,
                                   int64_t* gap_after, double* gap_score_after, int64_t min_indel_fuzz_length,
                                   double indel_fuzz_score_proportion, uint8_t* keep_out, uint64_t* n_kept_out) {
    if ((n_anchors && (!score || !gap_before || !gap_score_before || !gap_after || !gap_score_after || !keep_out)) || !n_kept_out) {
        cl_set_error(nullptr, "null argument");
        return CL_ERR_INVALID_ARGUMENT;
    }
    const auto removal = despecification_partition(n_anchors, score, gap_before, min_indel_fuzz_length, indel_fuzz_score_proportion);
    // src/stitcher.cpp:282-309, on parallel arrays: kept anchors are compacted to the front of the gap arrays
    size_t removed = 0, d = 0;
    int64_t gap = 0;
    double gap_score = 0.0;
    std::vector<int64_t> gb(gap_before, gap_before + n_anchors), ga(gap_after, gap_after + n_anchors);
    std::vector<double> gsb(gap_score_before, gap_score_before + n_anchors), gsa(gap_score_after, gap_score_after + n_anchors);
    std::vector<size_t> origin(n_anchors);
    for (size_t i = 0; i < n_anchors; ++i) origin[i] = i;
    for (size_t i = 0; i < n_anchors; ++i) keep_out[i] = 1;
    for (size_t i = 0; i < n_anchors; ++i) {
        if (d < removal.size() && i >= removal[d].first && i < removal[d].second) {
            gap += gb[i];
            gap_score += gsb[i];
            keep_out[origin[i]] = 0;
            ++removed;
        } else if (removed != 0) {
            gb[i - removed] = gb[i]; ga[i - removed] = ga[i]; gsb[i - removed] = gsb[i]; gsa[i - removed] = gsa[i];
            origin[i - removed] = origin[i];
        }
        if (d < removal.size() && i == removal[d].second) {
            ga[i - removed - 1] = gap;
            gsa[i - removed - 1] = gap_score;
            gb[i - removed] = gap;
            gsb[i - removed] = gap_score;
            gap = 0;
            gap_score = 0.0;
            ++d;
        }
    }
    const size_t kept = n_anchors - removed;
    for (size_t i = 0; i < kept; ++i) { gap_before[i] = gb[i]; gap_after[i] = ga[i]; gap_score_before[i] = gsb[i]; gap_score_after[i] = gsa[i]; }
    *n_kept_out = kept;
    return CL_OK;
}

}  // extern "C"

// =====================================================================================================================
// Anchorer::split_branching_matches (include/centrolign/anchorer.hpp:800-956)
// =====================================================================================================================
#include "stitch_host.hpp"


namespace {

// SuperbubbleTree(graph, tableau) + SuperbubbleDistances reduced to what the splitter asks: which superbubble begins /
// ends at a node and the spread between the longest and the shortest path through it.  Superbubbles by the sweep of
// superbubbles.hpp:63-170 over the reference's topological order; bubbles that touch a sentinel are dropped
// (structure_tree.hpp:164-169).  On a BaseGraph every label has size 1, so the min / max "distance" of
// structure_distances.hpp:107-150 is the number of nodes on the shortest / longest path between the two boundaries.
struct Bubbles {
    std::vector<uint32_t> begins, ends;   // node -> bubble id or none
    std::vector<uint64_t> spread;         // max - min path length through the bubble
    static constexpr uint32_t none = 0xFFFFFFFFu;
    bool build(const cl_base_graph& g) {
        const uint64_t n = g.n_nodes;
        std::vector<uint32_t> order;
        if (!clhost::topological_order(g, order)) return false;
        std::vector<int64_t> index(n);
        for (uint64_t i = 0; i < n; ++i) index[order[i]] = (int64_t)i;
        begins.assign(n, none);
        ends.assign(n, none);
        std::vector<int64_t> stack, backward(n, INT64_MAX);
        std::vector<std::pair<uint32_t, uint32_t>> found;
        for (int64_t i = (int64_t)n - 1; i >= 0; --i) {
            const uint32_t v = order[i];
            int64_t forward = -1;
            for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) forward = std::max(forward, index[g.next_idx[e]]);
            if (forward == i + 1) stack.push_back(i + 1);
            while (!stack.empty() && forward > stack.back()) {
                const int64_t bad = stack.back();
                stack.pop_back();
                if (!stack.empty()) backward[stack.back()] = std::min(backward[stack.back()], backward[bad]);
            }
            if (!stack.empty() && backward[stack.back()] == i) {
                const int64_t ok = stack.back();
                found.emplace_back(v, order[ok]);
                stack.pop_back();
                if (!stack.empty()) backward[stack.back()] = std::min(backward[stack.back()], backward[ok]);
            }
            for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) backward[i] = std::min(backward[i], index[g.prev_idx[e]]);
            if (!stack.empty()) backward[stack.back()] = std::min(backward[stack.back()], backward[i]);
        }
        std::vector<int64_t> dmin(n), dmax(n);
        for (const auto& b : found) {
            if (b.first == g.src_id || b.second == g.snk_id || b.first == g.snk_id || b.second == g.src_id) continue;
            const uint32_t id = (uint32_t)spread.size();
            begins[b.first] = id;
            ends[b.second] = id;
            // the bubble's nodes occupy the positions index[begin] .. index[end] of the order
            const int64_t lo = index[b.first], hi = index[b.second];
            for (int64_t i = lo; i <= hi; ++i) { dmin[order[i]] = INT64_MAX; dmax[order[i]] = -1; }
            dmin[b.first] = dmax[b.first] = 1;
            for (int64_t i = lo; i < hi; ++i) {
                const uint32_t v = order[i];
                if (dmax[v] < 0) continue;
                for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) {
                    const uint32_t w = g.next_idx[e];
                    dmin[w] = std::min(dmin[w], dmin[v] + 1);
                    dmax[w] = std::max(dmax[w], dmax[v] + 1);
                }
            }
            spread.push_back((uint64_t)(dmax[b.second] - dmin[b.second]));
        }
        return true;
    }
};

}  // namespace

extern "C" {

void cl_split_params_default(cl_split_params* p) {
    p->anchor_split_limit = 5;
    p->min_split_length = 128;
    p->min_path_length_spread = 50;
    p->max_split_match_set_size = 16;
}

void cl_owned_match_sets_view(const cl_owned_match_sets* o, cl_match_sets* v) {
    *v = cl_match_sets{o->count1.size(), o->set_off1.data(), o->walk_off1.data(), o->nodes1.data(), o->set_off2.data(), o->walk_off2.data(),
                       o->nodes2.data(), o->count1.data(), o->count2.data(), o->full_length.data()};
}

void cl_owned_match_sets_free(cl_owned_match_sets* o) { delete o; }

}  // extern "C"

namespace {

bool build_bubbles(const cl_base_graph& g1, const cl_base_graph& g2, Bubbles& b1, Bubbles& b2) {
    bool ok1 = false, ok2 = false;
    // (small graphs — the thousands of realignments of a polishing step — one after the other on the spot: a pool hand-off costs more than both decompositions)
    if (g1.n_nodes + g2.n_nodes < (1u << 16)) { ok1 = b1.build(g1); ok2 = b2.build(g2); return ok1 && ok2; }
    cl_pool_run(2, [&](unsigned t) { if (t) ok2 = b2.build(g2); else ok1 = b1.build(g1); });
    return ok1 && ok2;
}

bool no_wide_bubble(const Bubbles& b1, const Bubbles& b2, const cl_split_params& sp) {
    for (const Bubbles* b : {&b1, &b2})
        for (uint64_t spread : b->spread)
            if (spread >= sp.min_path_length_spread) return false;
    return true;
}

// the body of cl_split_branching_matches; with skip_identity, *out stays null when no match can be cut (no superbubble of either graph is
// wide enough: always the case for the leaf graphs of a pairwise merge) and the caller goes on with its own sets, without the 100-MB copy
int split_matches(const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms, const cl_split_params* sp, bool skip_identity,
                  cl_owned_match_sets** out) {
    *out = nullptr;
    // a set during the edit: walks as (begin, end) windows into the caller's node arrays
    struct Set { uint64_t src; uint64_t from, to; };   // piece [from, to) of every walk of original set src
    const uint64_t n_orig = ms->n_sets;
    std::vector<Set> sets(n_orig);
    for (uint64_t s = 0; s < n_orig; ++s) {
        const uint64_t w0 = ms->set_off1[s];
        const uint64_t len = ms->set_off1[s + 1] > w0 ? ms->walk_off1[w0 + 1] - ms->walk_off1[w0] : 0;
        sets[s] = Set{s, 0, len};
    }
    if (sp->anchor_split_limit != 0) {
        // two chains (every leaf merge) have no bubble to split at: no snarl decomposition of a million nodes to find that out (30 ms per merge)
        auto is_chain = [](const cl_base_graph& g) {
            for (uint64_t v = 0; v < g.n_nodes; ++v)
                if (g.next_off[v + 1] - g.next_off[v] > 1 || g.prev_off[v + 1] - g.prev_off[v] > 1) return false;
            return true;
        };
        if (skip_identity && is_chain(*g1) && is_chain(*g2)) return CL_OK;
        Bubbles b1, b2;
        if (!build_bubbles(*g1, *g2, b1, b2)) return CL_ERR_CYCLIC_GRAPH;
        if (skip_identity && no_wide_bubble(b1, b2, *sp)) return CL_OK;
        for (uint64_t s = 0; s < n_orig; ++s)
            if (ms->set_off1[s + 1] == ms->set_off1[s]) return CL_ERR_INVALID_ARGUMENT;   // the reference reads walks1.front()
        // the cuts of every set, side by side; the pieces are appended in the order of the sets afterwards
        std::vector<std::vector<uint64_t>> cuts(n_orig);
        cl_parallel_for(n_orig, [&](uint64_t sb, uint64_t se) {
            for (uint64_t s = sb; s < se; ++s) {
                const uint64_t n1 = ms->set_off1[s + 1] - ms->set_off1[s], n2 = ms->set_off2[s + 1] - ms->set_off2[s];
                const uint64_t len = sets[s].to;
                if (n1 * n2 > sp->max_split_match_set_size || len < sp->min_split_length) continue;
                auto branch = [&](uint64_t j, bool backwards) {
                    for (int side = 0; side < 2; ++side) {
                        const Bubbles& b = side ? b2 : b1;
                        const uint64_t* so = side ? ms->set_off2 : ms->set_off1;
                        const uint64_t* wo = side ? ms->walk_off2 : ms->walk_off1;
                        const uint32_t* nd = side ? ms->nodes2 : ms->nodes1;
                        for (uint64_t w = so[s]; w < so[s + 1]; ++w) {
                            const uint32_t id = (backwards ? b.ends : b.begins)[nd[wo[w] + j]];
                            if (id != Bubbles::none && b.spread[id] >= sp->min_path_length_spread) return true;
                        }
                    }
                    return false;
                };
                std::vector<uint64_t>& division = cuts[s];
                // note: stops early, a branch after the final position is not problematic
                for (uint64_t j = 0; j < len; ++j) {
                    if (j == sp->anchor_split_limit && j + sp->anchor_split_limit < len) j = len - sp->anchor_split_limit;   // skip to the suffix
                    if (j != 0 && (division.empty() || division.back() != j) && branch(j, true)) division.push_back(j);
                    if (j + 1 != len && branch(j, false)) division.push_back(j + 1);
                }
            }
        }, 2048);
        for (uint64_t s = 0; s < n_orig; ++s) {
            const std::vector<uint64_t>& division = cuts[s];
            if (division.empty()) continue;
            uint64_t end = sets[s].to;
            for (size_t q = division.size(); q-- > 0;) {
                sets.push_back(Set{s, division[q], end});
                end = division[q];
            }
            sets[s].to = division.front();
        }
    }
    // lay the pieces out: sizes, offsets, then the copy side by side
    std::unique_ptr<cl_owned_match_sets> o(new cl_owned_match_sets());
    const uint64_t m = sets.size();
    o->count1.resize(m); o->count2.resize(m); o->full_length.resize(m);
    std::vector<uint64_t> node_base[2];
    for (int side = 0; side < 2; ++side) {
        const uint64_t* so = side ? ms->set_off2 : ms->set_off1;
        auto& oso = side ? o->set_off2 : o->set_off1;
        oso.resize(m + 1);
        node_base[side].resize(m + 1);
        oso[0] = 0; node_base[side][0] = 0;
        for (uint64_t k = 0; k < m; ++k) {
            const uint64_t nw = so[sets[k].src + 1] - so[sets[k].src];
            oso[k + 1] = oso[k] + nw;
            node_base[side][k + 1] = node_base[side][k] + nw * (sets[k].to - sets[k].from);
        }
        (side ? o->walk_off2 : o->walk_off1).resize(oso[m] + 1);
        (side ? o->nodes2 : o->nodes1).resize(node_base[side][m]);
        (side ? o->walk_off2 : o->walk_off1)[0] = 0;
    }
    cl_parallel_for(m, [&](uint64_t kb, uint64_t ke) {
        for (uint64_t k = kb; k < ke; ++k) {
            const Set& st = sets[k];
            const uint64_t s = st.src, len = st.to - st.from;
            for (int side = 0; side < 2; ++side) {
                const uint64_t* so = side ? ms->set_off2 : ms->set_off1;
                const uint64_t* wo = side ? ms->walk_off2 : ms->walk_off1;
                const uint32_t* nd = side ? ms->nodes2 : ms->nodes1;
                uint64_t* owo = (side ? o->walk_off2 : o->walk_off1).data() + (side ? o->set_off2 : o->set_off1)[k];
                uint32_t* ond = (side ? o->nodes2 : o->nodes1).data();
                uint64_t at = node_base[side][k];
                for (uint64_t w = so[s], i = 0; w < so[s + 1]; ++w, ++i) {
                    if (len) std::memcpy(ond + at, nd + wo[w] + st.from, len * sizeof(uint32_t));
                    at += len;
                    owo[i + 1] = at;
                }
            }
            o->count1[k] = ms->count1[s];
            o->count2[k] = ms->count2[s];
            o->full_length[k] = ms->full_length[s];
        }
    }, 4096);
    *out = o.release();
    return CL_OK;
}

}  // namespace

// no superbubble of either graph is wide enough to cut a match at: cl_split_branching_matches would hand back a copy of its input
bool cl_split_is_identity(const cl_base_graph* g1, const cl_base_graph* g2, const cl_split_params* sp) {
    if (sp->anchor_split_limit == 0) return true;
    Bubbles b1, b2;
    if (!build_bubbles(*g1, *g2, b1, b2)) return false;   // let the real call report the cycle
    return no_wide_bubble(b1, b2, *sp);
}

// cl_split_branching_matches for callers inside the library: *out stays null when the split changes nothing
int cl_split_branching_matches_unless_identity(const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms, const cl_split_params* sp,
                                               cl_owned_match_sets** out) {
    if (sp->anchor_split_limit == 0) { *out = nullptr; return CL_OK; }
    return split_matches(g1, g2, ms, sp, true, out);
}

extern "C" {

int cl_split_branching_matches(const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms, const cl_split_params* sp,
                               cl_owned_match_sets** out) {
    if (!g1 || !g2 || !ms || !sp || !out) return CL_ERR_INVALID_ARGUMENT;
    return split_matches(g1, g2, ms, sp, false, out);
}

}  // extern "C"

// =====================================================================================================================
// Partitioner::partition_anchors (include/centrolign/partitioner.hpp:72-213)
// =====================================================================================================================
namespace {

typedef std::vector<std::pair<size_t, size_t>> Intervals;
const double kPartMinInf = std::numeric_limits<double>::lowest();

struct PartCtx {
    const cl_partition_params* pp;
    double min_score() const { return pp->minimum_segment_score * pp->score_scale; }
    double min_average() const { return pp->minimum_segment_average * pp->score_scale; }
    double adjust(double score, size_t i, size_t n) const {
        if (pp->score_boundaries) {
            if (i == 0) score += pp->boundary_score_factor * min_score();
            if (i + 1 == n) score += pp->boundary_score_factor * min_score();
        }
        return score;
    }
};

// ---- segment selection over the chain's anchors: the three rules of Partitioner (partitioner.hpp:215-684) ---------------------------------------
// What all three rules share, in this library's own terms.  A CUT is a position between two items (0 .. n).  Per cut i the table keeps
//   rest[i]: the best total of segments chosen among items 0 .. i - 1 with item i - 1 in no segment (or i == 0),
//   shut[i]: the best total with a segment ENDING at cut i, and from[i], the cut where that segment starts.
// A segment [a, i) is worth (prefix[i] - prefix[a]) - min_score on top of rest[a]; the rules differ in which starts a are admissible for an end i.
// The results have to be the reference's to the last bit — exact ties between candidate segments (a cut on either side of a zero-score item) are decided
// by rounding — so the sums are formed in the operation order of the reference AS BUILT (-O3 -ffast-math regroups the source's expressions; read off
// the disassembly of oracle/_ref): a segment's total is  (start term - min_score) + prefix[end]  with start term = rest[a] - prefix[a].
struct SegmentTable {
    std::vector<double> rest, shut;
    std::vector<size_t> from;
    size_t best_end = 0;
    explicit SegmentTable(size_t n_items) : rest(n_items + 1, kPartMinInf), shut(n_items + 1, kPartMinInf), from(n_items + 1, (size_t)-1) { rest[0] = 0; shut[0] = 0; }
    size_t cuts() const { return rest.size(); }
    void carry(size_t i) { rest[i] = std::max(rest[i - 1], shut[i - 1]); }
    void close_at(size_t i, double total, size_t start) { shut[i] = total; from[i] = start; }
    void note_end(size_t i) { if (shut[i] > shut[best_end]) best_end = i; }
    double start_term(size_t a, const std::vector<double>& prefix) const { return rest[a] - prefix[a]; }
    // back from the best end: a segment, then cuts without one until the value says a segment ends again (PartitionClient::traceback, partition_client.hpp:29-53)
    Intervals segments() const {
        Intervals out;
        bool at_segment_end = true;
        for (size_t i = best_end; i > 0;) {
            if (at_segment_end) {
                out.emplace_back(from[i], i);
                i = from[i];
                at_segment_end = false;
            } else {
                at_segment_end = rest[i] == shut[i - 1];
                --i;
            }
        }
        std::reverse(out.begin(), out.end());
        return out;
    }
};

// rule 1 (partitioner.hpp:215-270): any start; the best start term so far is carried along
Intervals maximum_weight_partition(const PartCtx& pc, const std::vector<double>& data) {
    const double min_score = pc.min_score();
    const size_t n = data.size();
    std::vector<double> prefix(n + 1, 0);
    for (size_t i = 0; i < n; ++i) prefix[i + 1] = prefix[i] + pc.adjust(data[i], i, n);
    SegmentTable T(n);
    size_t best_start = 0;
    for (size_t i = 1; i < T.cuts(); ++i) {
        T.carry(i);
        T.close_at(i, (prefix[i] - min_score) + T.start_term(best_start, prefix), best_start);
        if (T.start_term(i, prefix) > T.start_term(best_start, prefix)) best_start = i;
        T.note_end(i);
    }
    return T.segments();
}

// Starts indexed by their "surplus" — prefix of (score - weight x minimum average) — so that "segments whose average meets the minimum" is a key range:
// a static-shape max-tree over (surplus, cut) holding the start terms that are currently admissible (search_trees.hpp)
typedef std::pair<double, size_t> SurplusKey;
typedef clhost::MaxTree<SurplusKey, double> StartTree;

// rule 2 (partitioner.hpp:272-351): starts whose segment to i has at least the minimum average
Intervals average_constrained_partition(const PartCtx& pc, const std::vector<std::pair<double, double>>& data) {
    const double min_score = pc.min_score(), min_average = pc.min_average();
    const size_t n = data.size();
    auto score = [&](size_t i) { return pc.adjust(data[i].first, i, n); };
    // (item-indexed here, as the reference has them: prefix[i] / surplus[i] include item i; the first item enters unadjusted)
    std::vector<double> prefix(n), surplus(n);
    if (n) {
        prefix[0] = data[0].first;
        surplus[0] = data[0].first - data[0].second * min_average;
    }
    for (size_t i = 1; i < n; ++i) {
        prefix[i] = prefix[i - 1] + score(i);
        surplus[i] = surplus[i - 1] + score(i) - data[i].second * min_average;
    }
    std::vector<std::pair<SurplusKey, double>> leaves;
    leaves.reserve(n + 1);
    for (size_t i = 0; i < n; ++i) leaves.emplace_back(SurplusKey(surplus[i], i + 1), kPartMinInf);
    leaves.emplace_back(SurplusKey(0, 0), 0);
    StartTree starts(leaves);
    SegmentTable T(n);
    for (size_t i = 1; i < T.cuts(); ++i) {
        T.carry(i);
        const size_t hit = starts.range_max(SurplusKey(kPartMinInf, 0), SurplusKey(surplus[i - 1], (size_t)-1));
        if (hit != starts.end() && starts.val[hit] != kPartMinInf) {
            T.close_at(i, (starts.val[hit] - min_score) + prefix[i - 1], starts.key[hit].second);
            T.note_end(i);
        }
        starts.update(starts.find(SurplusKey(surplus[i - 1], i)), T.rest[i] - prefix[i - 1]);
    }
    return T.segments();
}

// Where a window of `length` units of weight that begins at item i (going right) or ends at item i (going left) stops, and whether that window — completed, when the
// sequence runs out first, by what the item next to the run-out would contribute (partitioner.hpp:383-443) — has the minimum average
struct WindowReach {
    std::vector<int64_t> stop[2];     // [0]: first item beyond the window going right, [1]: going left (-1 = ran off the front)
    std::vector<char> meets[2];
    template <class Score>
    WindowReach(const std::vector<std::pair<double, double>>& data, Score score, double length, double min_average) {
        const int64_t n = (int64_t)data.size();
        for (int dir = 0; dir < 2; ++dir) {
            stop[dir].resize(data.size());
            meets[dir].resize(data.size());
            const int64_t step = dir == 0 ? 1 : -1;
            double in_score = 0.0, in_weight = 0.0;
            int64_t edge = dir == 0 ? 0 : n - 1;
            for (int64_t i = edge; i < n && i >= 0; i += step) {
                while (edge < n && edge >= 0 && in_weight < length) {
                    in_score += score((size_t)edge);
                    in_weight += data[edge].second;
                    edge += step;
                }
                stop[dir][i] = edge;
                if ((edge < 0 || edge >= n) && in_weight < length) {
                    // the sequence ran out inside the window: as the neighbour we came from, or the window as far as it goes
                    if (i - step >= 0 && i - step < n) meets[dir][i] = meets[dir][i - step];
                    else meets[dir][i] = in_score >= min_average * in_weight;
                } else {
                    const double last_score = data[edge - step].first, last_weight = data[edge - step].second;
                    meets[dir][i] = last_weight * in_score + (length - in_weight) * last_score >= last_weight * min_average * length;
                }
                in_score -= score((size_t)i);
                in_weight -= data[i].second;
            }
        }
    }
};

// rule 3 (partitioner.hpp:353-684): minimum average over the segment AND over every window of window_length inside it.  Starts closer than a window to the end stay in
// the tree (rule 2 decides for them); a start that falls a window behind leaves the tree and competes as the ONE best "far" start, valid as long as no window
// beginning between it and the end's last full window, and none ending between its first full window and the end, misses the average (counts of failing windows)
Intervals window_average_constrained_partition(const PartCtx& pc, const std::vector<std::pair<double, double>>& data) {
    const double min_score = pc.min_score(), min_average = pc.min_average(), window_length = pc.pp->window_length;
    const size_t n = data.size();
    auto score = [&](size_t i) { return pc.adjust(data[i].first, i, n); };
    const WindowReach reach(data, score, window_length, min_average);
    const std::vector<int64_t>& right_stop = reach.stop[0];
    const std::vector<int64_t>& left_stop = reach.stop[1];
    std::vector<double> prefix(n + 1), surplus(n + 1);
    std::vector<int> fails_right(n + 1), fails_left(n + 1);      // windows that miss the average among those beginning (going right) / ending (going left) before each cut
    for (size_t i = 0; i < n; ++i) {
        prefix[i + 1] = prefix[i] + score(i);
        surplus[i + 1] = surplus[i] + score(i) - data[i].second * min_average;
        fails_right[i + 1] = fails_right[i] + (int)!reach.meets[0][i];
        fails_left[i + 1] = fails_left[i] + (int)!reach.meets[1][i];
    }
    std::vector<std::pair<SurplusKey, double>> leaves;
    leaves.reserve(n + 2);
    for (size_t i = 0; i <= n; ++i) leaves.emplace_back(SurplusKey(surplus[i], i), kPartMinInf);
    leaves.front().second = 0;
    StartTree starts(leaves);
    SegmentTable T(n);
    size_t behind = 0;                 // the first start still inside the current window
    double behind_weight = 0.0;        // weight between `behind` and the end
    size_t far_start = (size_t)-1, far_start_reach = (size_t)-1;   // the best start a window or more behind, and the item its first full window stops at
    size_t last_window = 0, reach_cursor = 0;
    size_t no_window_from = n;         // items from here on have no full window to their right
    for (double tail = 0.0; no_window_from != 0 && tail + data[no_window_from - 1].second < window_length; --no_window_from) tail += data[no_window_from - 1].second;
    for (size_t i = 1; i < T.cuts(); ++i) {
        while (last_window < no_window_from && right_stop[last_window] <= (int64_t)i) ++last_window;
        if (far_start != (size_t)-1 && (fails_right[far_start] != fails_right[last_window] || fails_left[far_start_reach] != fails_left[i])) far_start = (size_t)-1;
        behind_weight += data[i - 1].second;
        while (behind < n && behind_weight > window_length) {
            behind_weight -= data[behind].second;
            const size_t leaf = starts.find(SurplusKey(surplus[behind], behind));
            starts.update(leaf, kPartMinInf);
            const size_t a = starts.key[leaf].second;
            while (reach_cursor < n && left_stop[reach_cursor] + 1 < (int64_t)a) ++reach_cursor;
            if (fails_right[a] == fails_right[last_window] && fails_left[reach_cursor] == fails_left[i] &&
                (far_start == (size_t)-1 || T.start_term(a, prefix) > T.start_term(far_start, prefix))) {
                far_start = a;
                far_start_reach = reach_cursor;
            }
            ++behind;
        }
        T.carry(i);
        const size_t hit = starts.range_max(SurplusKey(kPartMinInf, 0), SurplusKey(surplus[i], (size_t)-1));
        if (hit != starts.end() && starts.val[hit] != kPartMinInf) T.close_at(i, (starts.val[hit] - min_score) + prefix[i], starts.key[hit].second);
        if (far_start != (size_t)-1) {
            const double total = T.start_term(far_start, prefix) + (prefix[i] - min_score);
            if (total > T.shut[i]) T.close_at(i, total, far_start);
        }
        T.note_end(i);
        starts.update(starts.find(SurplusKey(surplus[i], i)), T.start_term(i, prefix));
    }
    return T.segments();
}

// utility.hpp:255-285
double add_log(double log_x, double log_y) {
    return log_x < log_y ? log_y + log1p(exp(log_x - log_y)) : log_x + log1p(exp(log_y - log_x));
}
double generalized_mean(const std::vector<double>& v, double p) {
    double n = 0.0;
    if (p == 0.0) {
        double sum_log = 0.0;
        for (double x : v) { sum_log += log(x); n += 1.0; }
        return exp(sum_log / n);
    }
    double log_sum = std::numeric_limits<double>::lowest() / 2.0;
    for (double x : v) { log_sum = add_log(log_sum, p * log(x)); n += 1.0; }
    return exp((log_sum - log(n)) / p);
}

}  // namespace

extern "C" {

void cl_partition_params_default(cl_partition_params* p) {
    p->constraint_method = 3;
    p->minimum_segment_score = 15000.0;
    p->minimum_segment_average = 0.1;
    p->window_length = 10000.0;
    p->generalized_length_mean = -0.5;
    p->boundary_score_factor = 0.95;
    p->score_scale = 1.0;
    p->score_boundaries = 0;
    p->use_annotated_score = 0;
    const double go[3] = {1.25, 50.0, 5000.0}, ge[3] = {2.5, 0.1, 0.0015};
    for (int i = 0; i < 3; ++i) { p->score_function.gap_open[i] = go[i]; p->score_function.gap_extend[i] = ge[i]; }
    p->score_function.anchor_score_function = 2;
    p->score_function.pair_count_power = 0.5;
    p->score_function.length_intercept = 2250.0;
    p->score_function.length_decay_power = 2.0;
    p->score_function.global_anchoring = 1;
}

int cl_partition_anchors(const cl_base_graph* g1, const cl_base_graph* g2, const cl_anchor_fields* an, const cl_partition_params* pp,
                         uint64_t** segments_out, uint64_t* n_segments_out) {
    if (!g1 || !g2 || !an || !pp || !segments_out || !n_segments_out) return CL_ERR_INVALID_ARGUMENT;
    *segments_out = nullptr;
    *n_segments_out = 0;
    const size_t n = an->n_anchors;
    // partitioner.hpp:81-103: the count penalty is reduced for match sets used several times in this chain
    std::vector<uint64_t> num_from_set;
    if (!pp->use_annotated_score)
        for (size_t i = 0; i < n; ++i) {
            if (num_from_set.size() <= an->match_set[i]) num_from_set.resize(an->match_set[i] + 1, 0);
            ++num_from_set[an->match_set[i]];
        }
    auto anchor_score = [&](size_t i) -> double {
        if (pp->use_annotated_score) return an->score[i];
        const uint64_t used = num_from_set[an->match_set[i]];
        return clhost::anchor_weight(pp->score_function, an->count1[i] - used + 1, an->count2[i] - used + 1,
                                     an->walk_off[i + 1] - an->walk_off[i], an->full_length[i]);
    };
    const PartCtx pc{pp};
    Intervals partition;
    if (pp->constraint_method == 0) {
        partition.emplace_back(0, n);
    } else if (pp->constraint_method == 1) {
        std::vector<double> data(n);
        for (size_t i = 0; i < n; ++i) data[i] = anchor_score(i);
        partition = maximum_weight_partition(pc, data);
    } else if (pp->constraint_method == 2 || pp->constraint_method == 3) {
        // the gaps between the anchors: extract_graphs_between + minimum distance across either graph (:124-151)
        std::vector<uint64_t> seg_off{0, n};
        cl_anchor_segments sg{n ? 1u : 0u, seg_off.data(), an->walk_off, an->walk1, an->walk2};
        clhost::OwnedBatch ob;
        const int rc = clhost::extract_stitch_batch(*g1, *g2, sg, ob, cl_shared_table(g1), cl_shared_table(g2));
        if (rc) return rc;
        const size_t n_gaps = ob.only_del.size();
        std::vector<std::pair<double, double>> data(n + n_gaps);
        cl_parallel_for(data.size(), [&](uint64_t i_begin, uint64_t i_end) {   // (every entry is independent: 26 000 gaps and anchors per merge of 2 x 1 Mbp)
        for (size_t i = i_begin; i < i_end; ++i) {
            if (i % 2 == 0) {
                std::vector<double> sizes;
                for (int side = 0; side < 2; ++side) {
                    const auto& sd = ob.side[side];
                    const size_t kk = i / 2;
                    if (sd.node_off[kk + 1] == sd.node_off[kk]) sizes.push_back(0.00001);
                    else sizes.push_back((double)((uint64_t)clhost::min_source_sink(sd, kk) + 1));
                }
                data[i].first = 0.0;
                data[i].second = generalized_mean(sizes, pp->generalized_length_mean);
            } else {
                data[i].first = anchor_score(i / 2);
                data[i].second = (double)(an->walk_off[i / 2 + 1] - an->walk_off[i / 2]);
            }
        }
        }, 1024);
        partition = pp->constraint_method == 2 ? average_constrained_partition(pc, data) : window_average_constrained_partition(pc, data);
        for (auto& iv : partition) {
            iv.first /= 2;
            iv.second = std::min((iv.second + 1) / 2, n);
        }
        if (partition.size() == 1 && partition.front().first == partition.front().second) partition.pop_back();
    } else {
        return CL_ERR_INVALID_ARGUMENT;
    }
    uint64_t* out = (uint64_t*)malloc((partition.size() ? partition.size() : 1) * 2 * sizeof(uint64_t));
    if (!out) return CL_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < partition.size(); ++i) { out[2 * i] = partition[i].first; out[2 * i + 1] = partition[i].second; }
    *segments_out = out;
    *n_segments_out = partition.size();
    return CL_OK;
}

}  // extern "C"
