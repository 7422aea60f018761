// stitch_host.hpp — host-side producers of the stitch kernels' inputs.
//
//   PathMergeTable           <-> PathMerge<UIntSize,UIntChain>  include/centrolign/path_merge.hpp:96-277
//                                (the reachability / "last index on path p that reaches me" table; also the
//                                 coordinate system of the chaining DP)
//   extract_connecting_graph <-> include/centrolign/subgraph_extraction.hpp:52-125
//   extract_stitch_batch     <-> Extractor::extract_graphs_between(segments, ...)  include/centrolign/anchorer.hpp:494-585,
//                                flattened in the order Stitcher::stitch consumes the pairs (stitcher.hpp:157-203)
//
// Subgraph node numbering (DFS discovery order) and edge insertion order are reproduced exactly: the
// reference's traceback tie-breaks read previous() lists of these subgraphs (SURVEY.md §7 "hard parts").
#ifndef CL_STITCH_HOST_HPP
#define CL_STITCH_HOST_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/centrolign_amd.h"

#include <functional>
void cl_pool_run(unsigned n_tasks, const std::function<void(unsigned)>& task);   // cl_api.cpp

namespace clhost {

// Kahn's algorithm with a LIFO stack seeded in ascending id order (topological_order.hpp:12-60)
inline bool topological_order(const cl_base_graph& g, std::vector<uint32_t>& order) {
    const uint64_t n = g.n_nodes;
    order.clear();
    order.reserve(n);
    std::vector<uint32_t> stack, indeg(n);
    for (uint64_t v = 0; v < n; ++v) {
        indeg[v] = (uint32_t)(g.prev_off[v + 1] - g.prev_off[v]);
        if (!indeg[v]) stack.push_back((uint32_t)v);
    }
    while (!stack.empty()) {
        uint32_t v = stack.back();
        stack.pop_back();
        order.push_back(v);
        for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e)
            if (--indeg[g.next_idx[e]] == 0) stack.push_back(g.next_idx[e]);
    }
    return order.size() == n;
}

class PathMergeTable {
public:
    static constexpr uint32_t kNone = std::numeric_limits<uint32_t>::max();

    // ChainMerge (include/centrolign/chain_merge.hpp:100-160), the reachability structure Core::execute hands to Core::align when the chaining
    // algorithm is not SparseAffine (core.hpp:350-357, the CLI's -g 0 / 1): a node belongs to ONE chain — the first path that covers it, with its
    // index among that path's first-covered nodes — plus the sentinel chain; table[node][chain] = the last index on the chain that reaches the
    // node, by propagation along the edges in topological order.  Same queries as PathMerge below, so every user of the table works on either
    bool build_chain_merge(const cl_base_graph& g, bool tableau = true) {
        g_ = &g;
        n_ = g.n_nodes;
        chains_ = g.n_paths + (tableau ? 1 : 0);
        chain_mode_ = true;
        path_head_.assign(n_, kNone);
        index_.assign(chains_ * n_, kNone);
        next_chain_.assign(chains_ * n_, kNone);
        table_.assign(n_ * chains_, kNone);
        chain_off_.assign(1, 0);
        chain_nodes_.clear();
        std::vector<uint32_t> own_index(n_, kNone);
        for (uint64_t p = 0; p < g.n_paths; ++p) {
            uint32_t index = 0;
            for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i) {
                const uint32_t v = g.path_nodes[i];
                if (path_head_[v] != kNone) continue;
                path_head_[v] = (uint32_t)p;
                own_index[v] = index;
                index_[p * n_ + v] = index++;
                chain_nodes_.push_back(v);
            }
            chain_off_.push_back(chain_nodes_.size());
        }
        if (tableau) {
            const uint64_t pp = g.n_paths;
            path_head_[g.src_id] = (uint32_t)pp; own_index[g.src_id] = 0; index_[pp * n_ + g.src_id] = 0;
            path_head_[g.snk_id] = (uint32_t)pp; own_index[g.snk_id] = 1; index_[pp * n_ + g.snk_id] = 1;
            chain_nodes_.push_back((uint32_t)g.src_id);
            chain_nodes_.push_back((uint32_t)g.snk_id);
            chain_off_.push_back(chain_nodes_.size());
        }
        std::vector<uint32_t> order;
        if (!topological_order(g, order)) return false;
        for (uint32_t v : order) {
            const uint32_t cv = path_head_[v];
            if (cv == kNone) continue;                 // not covered by a path
            const uint32_t* row = &table_[(uint64_t)v * chains_];
            for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) {
                const uint32_t w = g.next_idx[e];
                if (path_head_[w] == kNone) continue;
                uint32_t* nrow = &table_[(uint64_t)w * chains_];
                for (uint64_t c = 0; c < chains_; ++c) {   // (signed comparison: "none" is overwritten, chain_merge.hpp:148-156)
                    const uint32_t cand = c == cv ? own_index[v] : row[c];
                    if (cand != kNone && (nrow[c] == kNone || cand > nrow[c])) nrow[c] = cand;
                }
            }
        }
        return true;
    }
    bool chain_mode() const { return chain_mode_; }

    // path_merge.hpp:96-163; with a tableau the sentinel pseudo-path is chain number path_size()
    bool build(const cl_base_graph& g, bool tableau = true) {
        g_ = &g;
        chain_mode_ = false;
        n_ = g.n_nodes;
        chains_ = g.n_paths + (tableau ? 1 : 0);
        path_head_.assign(n_, kNone);
        index_.assign(chains_ * n_, kNone);
        next_chain_.assign(chains_ * n_, kNone);
        table_.assign(n_ * chains_, kNone);
        for (uint64_t p = 0; p < g.n_paths; ++p) {
            uint32_t index = 0;
            for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i, ++index) {
                const uint32_t v = g.path_nodes[i];
                for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) table_[(uint64_t)g.next_idx[e] * chains_ + p] = index;
                index_[p * n_ + v] = index;
                next_chain_[p * n_ + v] = path_head_[v];
                path_head_[v] = (uint32_t)p;
            }
        }
        std::vector<uint32_t> order;
        if (!topological_order(g, order)) return false;
        for (uint32_t v : order) {
            uint32_t* row = &table_[(uint64_t)v * chains_];
            for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) {
                const uint32_t* prow = &table_[(uint64_t)g.prev_idx[e] * chains_];
                for (uint64_t p = 0; p < g.n_paths; ++p) {
                    if (row[p] == kNone) row[p] = prow[p];
                    else if (prow[p] != kNone) row[p] = std::max(prow[p], row[p]);
                }
            }
        }
        if (!tableau) return true;
        const uint64_t pp = g.n_paths;
        index_[pp * n_ + g.src_id] = 0;
        index_[pp * n_ + g.snk_id] = 1;
        path_head_[g.src_id] = (uint32_t)pp;   // path_merge.hpp:155-156: overwrites, sentinels are on no real path
        path_head_[g.snk_id] = (uint32_t)pp;
        for (uint64_t v = 0; v < n_; ++v)
            if (v != g.src_id) table_[v * chains_ + pp] = 0;
        return true;
    }
    uint64_t chain_size() const { return chains_; }
    uint32_t predecessor_index(uint64_t node, uint64_t chain) const { return table_[node * chains_ + chain]; }
    uint32_t index_on(uint64_t node, uint64_t chain) const { return index_[chain * n_ + node]; }
    // path_merge.hpp:235-248
    bool reachable(uint64_t from, uint64_t to) const {
        const uint32_t c = path_head_[from];
        if (c == kNone) return false;
        const uint32_t idx_from = index_[(uint64_t)c * n_ + from];
        const uint32_t last = table_[to * chains_ + c];
        return last != kNone && idx_from <= last;
    }
    // path_merge.hpp:255-264: most recently added path first
    template <class F>
    void for_each_chain_on(uint64_t node, F f) const {
        for (uint32_t p = path_head_[node]; p != kNone; p = next_chain_[(uint64_t)p * n_ + node]) f(p);
    }
    uint64_t node_at(uint64_t chain, uint64_t index) const {
        if (chain_mode_) return chain_nodes_[chain_off_[chain] + index];
        if (chain == g_->n_paths) return index ? g_->snk_id : g_->src_id;
        return g_->path_nodes[g_->path_off[chain] + index];
    }

private:
    const cl_base_graph* g_ = nullptr;
    uint64_t n_ = 0, chains_ = 0;
    std::vector<uint32_t> path_head_, index_, next_chain_, table_;
    bool chain_mode_ = false;                       // built by build_chain_merge
    std::vector<uint64_t> chain_off_;
    std::vector<uint32_t> chain_nodes_;
};

// flat batch under construction (the cl_stitch_batch layout, owned)
struct OwnedBatch {
    struct Side {
        std::vector<uint64_t> node_off{0}, prev_off{0}, next_off{0}, src_off{0}, snk_off{0}, back;
        std::vector<uint8_t> label;
        std::vector<uint32_t> prev_idx, next_idx, src_idx, snk_idx;
    } side[2];
    std::vector<uint8_t> only_del;
    cl_stitch_batch view_;
    const cl_stitch_batch* view() {
        view_.n_problems = only_del.size();
        for (int s = 0; s < 2; ++s) {
            Side& d = side[s];
            view_.side[s] = cl_graph_side{d.node_off.data(), d.label.data(), d.prev_off.data(), d.prev_idx.data(),
                                          d.next_off.data(), d.next_idx.data(), d.src_off.data(), d.src_idx.data(),
                                          d.snk_off.data(), d.snk_idx.data(), d.back.data()};
        }
        view_.only_deletion_alns = only_del.data();
        return &view_;
    }
};

// subgraph_extraction.hpp:52-125, appended to one side of the batch
class Extractor {
public:
    void extract(const cl_base_graph& g, const PathMergeTable& pm, uint64_t from_id, uint64_t to_id, OwnedBatch::Side& out) {
        fwd_.clear();
        edges_.clear();
        stack_.assign(1, from_id);
        const uint64_t node_base = out.label.size();
        uint32_t n_sub = 0;
        while (!stack_.empty()) {
            const uint64_t node = stack_.back();
            stack_.pop_back();
            for (uint64_t e = g.next_off[node]; e < g.next_off[node + 1]; ++e) {
                const uint64_t nxt = g.next_idx[e];
                if (nxt == to_id && node != from_id) {
                    out.snk_idx.push_back(fwd_[node]);
                    continue;
                }
                if (!pm.reachable(nxt, to_id)) continue;
                auto it = fwd_.find(nxt);
                if (it == fwd_.end()) {
                    it = fwd_.emplace(nxt, n_sub++).first;
                    out.back.push_back(nxt);
                    out.label.push_back(g.label[nxt]);
                    stack_.push_back(nxt);
                }
                if (node != from_id) edges_.emplace_back(fwd_.at(node), it->second);
                else out.src_idx.push_back(it->second);
            }
        }
        // adjacency lists in edge-insertion order (BaseGraph::add_edge appends to next[from] and prev[to])
        deg_.assign(2 * (size_t)n_sub + 2, 0);
        uint64_t* nd = deg_.data();            // next degrees
        uint64_t* pd = deg_.data() + n_sub + 1;  // prev degrees
        for (auto& ed : edges_) { nd[ed.first]++; pd[ed.second]++; }
        const uint64_t nb = out.next_idx.size(), pb = out.prev_idx.size();
        out.next_idx.resize(nb + edges_.size());
        out.prev_idx.resize(pb + edges_.size());
        cur_.assign(2 * (size_t)n_sub, 0);
        uint64_t an = nb, ap = pb;
        for (uint32_t v = 0; v < n_sub; ++v) {
            cur_[v] = an; an += nd[v]; out.next_off.push_back(an);
            cur_[n_sub + v] = ap; ap += pd[v]; out.prev_off.push_back(ap);
        }
        for (auto& ed : edges_) {
            out.next_idx[cur_[ed.first]++] = ed.second;
            out.prev_idx[cur_[n_sub + ed.second]++] = ed.first;
        }
        out.node_off.push_back(node_base + n_sub);
        out.src_off.push_back(out.src_idx.size());
        out.snk_off.push_back(out.snk_idx.size());
    }

private:
    std::unordered_map<uint64_t, uint32_t> fwd_;
    std::vector<std::pair<uint32_t, uint32_t>> edges_;
    std::vector<uint64_t> stack_, deg_, cur_;
};

// Extractor::extract_graphs_between(segments, ...) (anchorer.hpp:494-585) in Stitcher::stitch's consumption order:
// before-first, then per segment its within-segment gaps followed by the gap to the next segment / the sink.
// with_ends = false: only the gaps between consecutive anchors, the overload without tableaus that Stitcher::internal_stitch uses
// (anchorer.hpp:423-431; no gap in front of the first anchor or behind the last one)
inline int extract_stitch_batch(const cl_base_graph& g1, const cl_base_graph& g2, const cl_anchor_segments& sg, OwnedBatch& out,
                                const PathMergeTable* have1 = nullptr, const PathMergeTable* have2 = nullptr, bool with_ends = true) {
    PathMergeTable own1, own2;
    if ((!have1 && !own1.build(g1)) || (!have2 && !own2.build(g2))) return CL_ERR_CYCLIC_GRAPH;
    const PathMergeTable& pm1 = have1 ? *have1 : own1;
    const PathMergeTable& pm2 = have2 ? *have2 : own2;
    // the gaps are independent: list them in consumption order, extract chunks of the list side by side, append the chunks in order
    struct Gap { uint64_t f1, t1, f2, t2; bool only_del; };
    std::vector<Gap> gaps;
    auto add = [&](uint64_t f1, uint64_t t1, uint64_t f2, uint64_t t2, bool only_del) { gaps.push_back(Gap{f1, t1, f2, t2, only_del}); };
    struct Finish {   // runs when the function returns: every path below only fills `gaps`
        const cl_base_graph& g1; const cl_base_graph& g2; const PathMergeTable& pm1; const PathMergeTable& pm2;
        std::vector<Gap>& gaps; OwnedBatch& out;
        ~Finish() {
            const uint64_t n = gaps.size();
            unsigned hw = std::thread::hardware_concurrency();
            const uint64_t nt = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw ? hw : 1, 16), n / 512));
            std::vector<OwnedBatch> part(nt);
            auto work = [&](uint64_t t) {
                Extractor ex;
                OwnedBatch& o = nt == 1 ? out : part[t];
                for (uint64_t k = n * t / nt; k < n * (t + 1) / nt; ++k) {
                    ex.extract(g1, pm1, gaps[k].f1, gaps[k].t1, o.side[0]);
                    ex.extract(g2, pm2, gaps[k].f2, gaps[k].t2, o.side[1]);
                    o.only_del.push_back(gaps[k].only_del ? 1 : 0);
                }
            };
            if (nt == 1) { work(0); return; }
            cl_pool_run((unsigned)nt, [&](unsigned t) { work(t); });   // the process's thread pool (cl_internal.hpp)
            for (uint64_t t = 0; t < nt; ++t) {
                out.only_del.insert(out.only_del.end(), part[t].only_del.begin(), part[t].only_del.end());
                for (int sd = 0; sd < 2; ++sd) {
                    OwnedBatch::Side& d = out.side[sd];
                    const OwnedBatch::Side& p = part[t].side[sd];
                    auto append_off = [](std::vector<uint64_t>& dst, const std::vector<uint64_t>& src, uint64_t base) {
                        for (size_t i = 1; i < src.size(); ++i) dst.push_back(src[i] + base);   // both start with a leading 0
                    };
                    append_off(d.node_off, p.node_off, d.label.size());
                    append_off(d.next_off, p.next_off, d.next_idx.size());
                    append_off(d.prev_off, p.prev_off, d.prev_idx.size());
                    append_off(d.src_off, p.src_off, d.src_idx.size());
                    append_off(d.snk_off, p.snk_off, d.snk_idx.size());
                    d.label.insert(d.label.end(), p.label.begin(), p.label.end());
                    d.back.insert(d.back.end(), p.back.begin(), p.back.end());
                    d.next_idx.insert(d.next_idx.end(), p.next_idx.begin(), p.next_idx.end());
                    d.prev_idx.insert(d.prev_idx.end(), p.prev_idx.begin(), p.prev_idx.end());
                    d.src_idx.insert(d.src_idx.end(), p.src_idx.begin(), p.src_idx.end());
                    d.snk_idx.insert(d.snk_idx.end(), p.snk_idx.begin(), p.snk_idx.end());
                }
            }
        }
    } finish{g1, g2, pm1, pm2, gaps, out};
    auto first1 = [&](uint64_t a) { return (uint64_t)sg.walk1[sg.walk_off[a]]; };
    auto last1 = [&](uint64_t a) { return (uint64_t)sg.walk1[sg.walk_off[a + 1] - 1]; };
    auto first2 = [&](uint64_t a) { return (uint64_t)sg.walk2[sg.walk_off[a]]; };
    auto last2 = [&](uint64_t a) { return (uint64_t)sg.walk2[sg.walk_off[a + 1] - 1]; };
    if (sg.n_segments == 0) {
        if (with_ends) add(g1.src_id, g1.snk_id, g2.src_id, g2.snk_id, true);
        return CL_OK;
    }
    for (uint64_t s = 0; s < sg.n_segments; ++s)
        if (sg.seg_off[s + 1] <= sg.seg_off[s]) return CL_ERR_INVALID_ARGUMENT;  // the reference assumes non-empty segments
    const uint64_t a0 = sg.seg_off[0];
    if (with_ends) add(g1.src_id, first1(a0), g2.src_id, first2(a0), true);
    for (uint64_t s = 0; s < sg.n_segments; ++s) {
        for (uint64_t a = sg.seg_off[s] + 1; a < sg.seg_off[s + 1]; ++a) add(last1(a - 1), first1(a), last2(a - 1), first2(a), false);
        const uint64_t al = sg.seg_off[s + 1] - 1;
        if (s + 1 < sg.n_segments) add(last1(al), first1(sg.seg_off[s + 1]), last2(al), first2(sg.seg_off[s + 1]), true);
        else if (with_ends) add(last1(al), g1.snk_id, last2(al), g2.snk_id, true);
    }
    return CL_OK;
}

// minimum source->sink distance in nodes of one extracted subgraph, as source_sink_minmax(...).first (src/anchorer.cpp:14-23)
inline int64_t min_source_sink(const clhost::OwnedBatch::Side& sd, uint64_t k) {
    const uint64_t b = sd.node_off[k], n = sd.node_off[k + 1] - b;
    std::vector<int64_t> dist(n, INT64_MAX);
    std::vector<uint32_t> indeg(n), st, order;
    for (uint64_t v = 0; v < n; ++v) { indeg[v] = (uint32_t)(sd.prev_off[b + v + 1] - sd.prev_off[b + v]); if (!indeg[v]) st.push_back((uint32_t)v); }
    while (!st.empty()) {
        uint32_t v = st.back(); st.pop_back(); order.push_back(v);
        for (uint64_t e = sd.next_off[b + v]; e < sd.next_off[b + v + 1]; ++e) if (--indeg[sd.next_idx[e]] == 0) st.push_back(sd.next_idx[e]);
    }
    for (uint64_t i = sd.src_off[k]; i < sd.src_off[k + 1]; ++i) dist[sd.src_idx[i]] = 0;
    for (uint32_t v : order)
        if (dist[v] != INT64_MAX)
            for (uint64_t e = sd.next_off[b + v]; e < sd.next_off[b + v + 1]; ++e) dist[sd.next_idx[e]] = std::min(dist[sd.next_idx[e]], dist[v] + 1);
    int64_t mn = INT64_MAX;
    for (uint64_t i = sd.snk_off[k]; i < sd.snk_off[k + 1]; ++i) mn = std::min(mn, dist[sd.snk_idx[i]]);
    return mn;
}


inline double anchor_weight(const cl_chain_params& cp, uint64_t count1, uint64_t count2, uint64_t length, uint64_t full_length) {
    // ScoreFunction::anchor_weight (score_function.hpp:51-75) in the operation order of the reference AS BUILT: its
    // CMakeLists.txt:9 compiles with -O3 -ffast-math, under which gcc turns "x / pow(c, p)" into "x * pow(c, -p)" and
    // regroups the products (disassembly of oracle/_ref).  Mathematically tied weights (e.g. lengths symmetric about the
    // vertex of the concave length term) order match sets in the budget selection (anchorer.hpp:1130-1134), so the
    // last bit matters.
    const double count = (double)(count1 * count2);
    const double fraction = double(length) / double(full_length);
    switch (cp.anchor_score_function) {
    case 0: return pow(count, -cp.pair_count_power) * fraction;
    case 1: return (fraction * (double)length) * pow(count, -cp.pair_count_power);
    case 2: {
        const double inv = pow(count, -cp.pair_count_power);
        const double decay = pow((double)length / cp.length_intercept, cp.length_decay_power);
        return (inv * (double)length - cp.length_intercept * decay) * fraction;
    }
    default: {
        const double decay = pow((double)length / cp.length_intercept, cp.length_decay_power);
        return ((double)length - (cp.length_intercept * count) * decay) * fraction;
    }
    }
}


}  // namespace clhost

#endif
