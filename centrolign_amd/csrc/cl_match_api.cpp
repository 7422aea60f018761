// cl_match_api.cpp — PathMatchFinder::find_matches (include/centrolign/match_finder.hpp:120-212): the minimal rare matches
// between the embedded paths of two graphs.
//
// The reference builds an enhanced suffix array over the joined path sequences (PathESA, path_esa.hpp:81-170: SA-IS, Kasai
// LCP, child table, suffix links, Hui's colour-set-size index) and walks its LCP-interval tree bottom-up
// (esa.hpp:284-494).  Here the suffix array and the LCP array come from the device (match_kernels.hip); the tree itself is
// one linear stack pass on the host that produces, per internal node: interval, string depth, parent, and the number of
// DISTINCT start nodes per graph below it (leaves minus duplicates, the duplicates charged to the lowest common ancestor
// of consecutive occurrences of a node — Hui 1992, as src/esa.cpp:149-300 does with an Euler tour + RMQ; here the LCA of
// leaf i with an earlier leaf is simply the deepest entry of the traversal stack that starts at or before that leaf).
// No child table and no suffix-link table are materialised: the only suffix-link use (esa.hpp:352-362: "the sibling
// reached by dropping the first character") is the highest ancestor of leaf ISA[SA[begin] + 1] whose depth is still
// >= the parent's depth, found by climbing parent pointers.
//
// A match is reported for an internal node C with parent P (esa.hpp:296-431) when
//   0 < count1(C) * count2(C) <= max_count, and, unless P is the root, some graph sees more start nodes under P than
//   under C (one character shorter on the right is more frequent) and some graph sees more under the sibling reached by
//   dropping the first character than under C (one shorter on the left is more frequent);
// its length is depth(P) + 1.  The "child had a too-frequent descendant" flags of esa.hpp:291,308,371 only skip work: counts
// grow towards the root, so such a child fails the max_count test anyway.  Matches come out in the order the reference
// emits them (parents in the order the stack pass closes them, children left to right), are filtered by
// ScoreFunction::anchor_weight > 0 (match_finder.hpp:155-167) and walked out (esa.hpp:610-665: suffix-array order, one
// walk per distinct (graph, start node), the joined ids of the next `length` text positions).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "cl_internal.hpp"
#include "match_device.h"
#include "stitch_host.hpp"

namespace {

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// the joined text of PathESA's constructor (path_esa.hpp:92-118)
struct JoinedText {
    std::vector<uint8_t> text;    // label + 1; sentinel characters around every path; a final 0
    std::vector<uint32_t> id;     // joined_ids
    uint64_t first2 = 0;          // text positions >= first2 belong to graph 2 (index_ranges, the final 0 included)
    uint64_t n_ids[2] = {0, 0};   // ids of graph c are < n_ids[c]
    int build(const cl_base_graph& g1, const cl_base_graph& g2) {
        const cl_base_graph* gs[2] = {&g1, &g2};
        uint64_t total = 1;
        for (auto g : gs) {
            if (g->n_nodes && (g->src_id >= g->n_nodes || g->snk_id >= g->n_nodes)) return CL_ERR_INVALID_ARGUMENT;
            if (g->n_paths && g->n_nodes == 0) return CL_ERR_INVALID_ARGUMENT;
            total += (g->n_paths ? g->path_off[g->n_paths] : 0) + 2 * g->n_paths;
        }
        if (total >= 0x7FFFFFFFull) return CL_ERR_INVALID_ARGUMENT;
        text.reserve(total);
        id.reserve(total);
        for (int c = 0; c < 2; ++c) {
            const cl_base_graph& g = *gs[c];
            for (uint64_t p = 0; p < g.n_paths; ++p) {
                text.push_back((uint8_t)(g.label[g.src_id] + 1));
                id.push_back((uint32_t)g.src_id);
                for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i) {
                    const uint32_t v = g.path_nodes[i];
                    if (v >= g.n_nodes || g.label[v] == 0xFF) return CL_ERR_INVALID_ARGUMENT;
                    text.push_back((uint8_t)(g.label[v] + 1));
                    id.push_back(v);
                }
                text.push_back((uint8_t)(g.label[g.snk_id] + 1));
                id.push_back((uint32_t)g.snk_id);
            }
            if (c == 0) first2 = text.size();
            n_ids[c] = g.n_nodes + 1;
        }
        text.push_back(0);                       // the sentinel SA-IS needs (:113-114) ...
        id.push_back((uint32_t)g2.n_nodes);      // ... arbitrarily assigned to the last graph (:115-117)
        return CL_OK;
    }
    int comp(uint32_t pos) const { return pos >= first2 ? 1 : 0; }
};

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Found {
    uint32_t node, length;
    uint64_t count[2];
};

int matches_from_esa(const cl_base_graph& g1, const cl_base_graph& g2, const cl_match_params& prm, const JoinedText& T, const uint32_t* sa,
                     const uint32_t* lcp, const uint32_t* isa, cl_owned_match_sets& out, cl_match_stats* st) {
    const uint32_t n = (uint32_t)T.text.size();
    auto t0 = clk::now();
    // ---- the LCP-interval tree, bottom-up (the stack pass of esa.hpp:436-494), with Hui's duplicate counts — cut into ranges of suffix-array
    //      positions that run side by side.  A node is NAMED by the position at which the pass opens it (0 = the root; at most one node opens
    //      per position), so every range names the nodes alike without talking to the others:
    //        * the intervals still open in front of a range are the strict minima of the LCP values met going left from it (depth = the
    //          minimum, left end = the position of the next smaller value, opened at the leftmost position that holds the minimum);
    //        * the earlier occurrence of a leaf's start node (Hui's "previous leaf of the same colour") is tabulated beforehand from the
    //          inverse suffix array instead of carried along the pass;
    //        * the nodes below a node are exactly those opened at positions (l, r] of its interval, so the subtree totals of the
    //          duplicates (src/esa.cpp:235-300) are differences of one prefix sum over positions.
    //      Closing order (children before parents, the order the reference meets the nodes in) = the ranges' closing lists one after another.
    //      (ClRawVec: the big arrays come uninitialised from the thread's block cache — fresh zeroed vectors cost a page fault per 4 KB, and
    //      those wait while another thread of the process locks pages)
    uint32_t grain = 1u << 16;
    uint64_t max_ranges = cl_pool_width();
    if (const char* e = getenv("CL_MATCH_TREE_GRAIN")) { grain = (uint32_t)std::max(1, atoi(e)); max_ranges = 4096; }   // test hook: many small ranges
    const unsigned n_ranges = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(max_ranges, n / grain));
    std::vector<uint32_t> cut(n_ranges + 1);
    for (unsigned t = 0; t <= n_ranges; ++t) cut[t] = (uint32_t)((uint64_t)n * t / n_ranges);
    auto comp_of = [&](uint32_t i) { return T.comp(sa[i]); };
    // leaves of graph 0 at suffix-array positions < p (graph 1: p minus that)
    ClRawVec<uint32_t> before0((size_t)n + 1);
    {
        std::vector<uint32_t> part(n_ranges + 1, 0);
        cl_pool_run(n_ranges, [&](unsigned t) {
            uint32_t k = 0;
            for (uint32_t i = cut[t]; i < cut[t + 1]; ++i) k += comp_of(i) == 0;
            part[t + 1] = k;
        });
        for (unsigned t = 0; t < n_ranges; ++t) part[t + 1] += part[t];
        cl_pool_run(n_ranges, [&](unsigned t) {
            uint32_t k = part[t];
            for (uint32_t i = cut[t]; i < cut[t + 1]; ++i) { before0[i] = k; k += comp_of(i) == 0; }
            if (t + 1 == n_ranges) before0[n] = k;
        });
    }
    auto before = [&](int c, uint32_t p) { return c ? p - before0[p] : before0[p]; };
    // prev_of[i]: the largest suffix-array position < i whose suffix starts at the same node of the same graph
    ClRawVec<uint32_t> prev_of(n);
    {
        const uint64_t n_keys = T.n_ids[0] + T.n_ids[1];
        auto key_of = [&](uint32_t pos) { return (uint64_t)(T.comp(pos) ? T.n_ids[0] + T.id[pos] : T.id[pos]); };
        ClRawVec<uint32_t> key_off(n_keys + 1), fill(n_keys), occ(n);
        cl_parallel_for(n_keys, [&](uint64_t b, uint64_t e) { std::memset(fill.data() + b, 0, (e - b) * sizeof(uint32_t)); });
        cl_parallel_for(n, [&](uint64_t b, uint64_t e) { for (uint64_t pos = b; pos < e; ++pos) __atomic_fetch_add(&fill[key_of((uint32_t)pos)], 1u, __ATOMIC_RELAXED); });
        {   // exclusive prefix sum of the counts
            const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(cl_pool_width(), n_keys / 65536));
            std::vector<uint64_t> kc(nt + 1), sum(nt + 1, 0);
            for (unsigned t = 0; t <= nt; ++t) kc[t] = n_keys * t / nt;
            cl_pool_run(nt, [&](unsigned t) { uint64_t a = 0; for (uint64_t k = kc[t]; k < kc[t + 1]; ++k) a += fill[k]; sum[t + 1] = a; });
            for (unsigned t = 0; t < nt; ++t) sum[t + 1] += sum[t];
            cl_pool_run(nt, [&](unsigned t) { uint32_t a = (uint32_t)sum[t]; for (uint64_t k = kc[t]; k < kc[t + 1]; ++k) { key_off[k] = a; a += fill[k]; } });
            key_off[n_keys] = n;
        }
        cl_parallel_for(n, [&](uint64_t b, uint64_t e) {
            for (uint64_t pos = b; pos < e; ++pos) {
                const uint64_t k = key_of((uint32_t)pos);
                occ[key_off[k] + __atomic_sub_fetch(&fill[k], 1u, __ATOMIC_RELAXED)] = isa[pos];
            }
        });
        cl_parallel_for(n_keys, [&](uint64_t b, uint64_t e) {
            for (uint64_t k = b; k < e; ++k) {
                uint32_t* lo = occ.data() + key_off[k];
                uint32_t* hi = occ.data() + key_off[k + 1];
                if (lo == hi) continue;
                if (hi - lo > 1) std::sort(lo, hi);
                prev_of[*lo] = kNone;
                for (uint32_t* q = lo + 1; q < hi; ++q) prev_of[*q] = q[-1];
            }
        });
    }
    ClRawVec<uint32_t> node_l(n), node_r(n), node_parent(n);   // by node = opening position
    ClRawVec<uint32_t> leaf_parent(n);                          // deepest internal node above every leaf
    ClRawVec<uint32_t> dup[2];                                   // duplicates whose LCA is the node; prefix sums over positions afterwards
    for (int c = 0; c < 2; ++c) {
        dup[c].resize(n);
        cl_parallel_for(n, [&](uint64_t b, uint64_t e) { std::memset(dup[c].data() + b, 0, (e - b) * sizeof(uint32_t)); });
    }
    auto depth_of = [&](uint32_t v) -> uint32_t { return v ? lcp[v] : 0; };
    node_l[0] = 0; node_parent[0] = kNone;
    struct Open { uint32_t node, l, depth; };
    std::vector<ClRawVec<uint32_t>> closed_part(n_ranges);     // node ids in the order the pass closes them (children before parents)
    std::vector<uint32_t> late_leaf_parent(n_ranges, kNone);   // a node that opens at a range's first position sits above the leaf in front of it
    cl_pool_run(n_ranges, [&](unsigned t) {
        const uint32_t a = cut[t], b = cut[t + 1];
        ClRawVec<uint32_t>& closed = closed_part[t];
        closed.reserve((size_t)(b - a) + 64);
        std::vector<Open> stack;
        if (a == 0) stack.push_back(Open{0, 0, 0});
        else {
            std::vector<Open> deepest_first;
            uint32_t cur = kNone, leftmost = 0;
            for (uint32_t j = a - 1; j >= 1; --j) {
                const uint32_t v = lcp[j];
                if (cur == kNone) { cur = v; leftmost = j; }
                else if (v < cur) { deepest_first.push_back(Open{leftmost, j, cur}); cur = v; leftmost = j; }
                else if (v == cur) leftmost = j;
                if (cur == 0) break;
            }
            if (cur != kNone && cur != 0) deepest_first.push_back(Open{leftmost, 0, cur});
            deepest_first.push_back(Open{0, 0, 0});
            stack.assign(deepest_first.rbegin(), deepest_first.rend());
        }
        if (a == 0) leaf_parent[0] = 0;
        for (uint32_t i = std::max<uint32_t>(a, 1); i < b; ++i) {
            const uint32_t li = lcp[i];
            uint32_t last = kNone, left = i - 1;
            bool fresh = true;
            while (stack.back().depth > li) {
                const Open v = stack.back();
                stack.pop_back();
                node_r[v.node] = i - 1;
                closed.push_back(v.node);
                left = v.l;
                fresh = false;
                if (stack.back().depth >= li) { node_parent[v.node] = stack.back().node; last = kNone; }
                else last = v.node;
            }
            if (stack.back().depth < li) {
                stack.push_back(Open{i, left, li});
                node_l[i] = left;
                if (last != kNone) node_parent[last] = i;
                if (fresh) {   // the interval opens at leaf i-1
                    if (i - 1 >= a) leaf_parent[i - 1] = i; else late_leaf_parent[t] = i;
                }
            }
            leaf_parent[i] = stack.back().node;
            const uint32_t prev = prev_of[i];
            if (prev != kNone) {
                // deepest open interval that also holds the previous occurrence: interval starts grow along the stack.  Copies of a node with
                // a long shared context are neighbours in the suffix array, so the answer is mostly at or just under the top: look there first
                size_t hi = stack.size() - 1;
                int probes = 6;
                while (hi > 0 && probes-- > 0 && stack[hi].l > prev) --hi;
                if (hi > 0 && stack[hi].l > prev) {
                    size_t lo = 0;
                    --hi;
                    while (lo < hi) {
                        const size_t mid = (lo + hi + 1) / 2;
                        if (stack[mid].l <= prev) lo = mid; else hi = mid - 1;
                    }
                    hi = lo;
                }
                __atomic_fetch_add(&dup[comp_of(i)][stack[hi].node], 1u, __ATOMIC_RELAXED);
            }
        }
        if (b == n)
            while (!stack.empty()) {
                const Open v = stack.back();
                stack.pop_back();
                node_r[v.node] = n - 1;
                closed.push_back(v.node);
                if (!stack.empty()) node_parent[v.node] = stack.back().node;
            }
    });
    for (unsigned t = 1; t < n_ranges; ++t)
        if (late_leaf_parent[t] != kNone) leaf_parent[cut[t] - 1] = late_leaf_parent[t];
    std::vector<uint64_t> closed_off(n_ranges + 1, 0);
    for (unsigned t = 0; t < n_ranges; ++t) closed_off[t + 1] = closed_off[t] + closed_part[t].size();
    const uint32_t n_nodes = (uint32_t)closed_off[n_ranges];
    ClRawVec<uint32_t> closed(n_nodes), close_rank(n);
    cl_pool_run(n_ranges, [&](unsigned t) {
        const ClRawVec<uint32_t>& part = closed_part[t];
        uint32_t k = (uint32_t)closed_off[t];
        for (uint32_t v : part) { closed[k] = v; close_rank[v] = k; ++k; }
    });
    closed_part.clear();
    // dup[c][p] := duplicates charged to the nodes opened at positions <= p
    for (int c = 0; c < 2; ++c) {
        std::vector<uint64_t> sum(n_ranges + 1, 0);
        cl_pool_run(n_ranges, [&](unsigned t) { uint64_t a = 0; for (uint32_t i = cut[t]; i < cut[t + 1]; ++i) a += dup[c][i]; sum[t + 1] = a; });
        for (unsigned t = 0; t < n_ranges; ++t) sum[t + 1] += sum[t];
        cl_pool_run(n_ranges, [&](unsigned t) { uint32_t a = (uint32_t)sum[t]; for (uint32_t i = cut[t]; i < cut[t + 1]; ++i) { a += dup[c][i]; dup[c][i] = a; } });
    }
    // distinct start nodes of graph c below node v = leaves - duplicates
    auto distinct = [&](uint32_t v, int c) -> uint64_t {
        const uint32_t l = node_l[v], r = node_r[v];
        const uint32_t dups = v ? dup[c][r] - dup[c][l] : dup[c][r];
        return (uint64_t)(before(c, r + 1) - before(c, l)) - dups;
    };
    if (st) { st->n_internal_nodes = n_nodes; st->tree_ms = ms_since(t0); }

    // ---- the query (esa.hpp:290-431): every (parent, child) pair is independent.  The reference meets the children grouped under their
    //      parents, parents in closing order, children in closing order: evaluate them all side by side in closing order of the child and
    //      sort the few that pass by (closing rank of the parent, closing rank of the child)
    t0 = clk::now();
    struct Kept { uint64_t key; Found f; };
    std::vector<std::vector<Kept>> kept_parts;
    std::mutex kept_mutex;
    cl_parallel_for(n_nodes, [&](uint64_t b, uint64_t e) {
        std::vector<Kept> mine;
        for (uint64_t x = b; x < e; ++x) {
            const uint32_t C = closed[x], P = node_parent[C];
            if (P == kNone) continue;
            const uint64_t c0 = distinct(C, 0);
            if (c0 == 0) continue;                   // esa.hpp:391-393: counts stay 0, total 0
            const uint64_t c1 = distinct(C, 1);
            if (c1 == 0) continue;
            unsigned __int128 total = (unsigned __int128)c0 * c1;   // sat_mult (:399-402)
            if (total > prm.max_count) continue;
            const uint32_t d = depth_of(P);
            if (d != 0) {
                if (!(c0 < distinct(P, 0) || c1 < distinct(P, 1))) continue;            // parent_more_frequent (:397)
                // the sibling under the parent's suffix link (:352-362): drop the first character of C's string
                const uint32_t q = isa[sa[node_l[C]] + 1];
                uint32_t L = leaf_parent[q];
                while (node_parent[L] != kNone && depth_of(node_parent[L]) >= d) L = node_parent[L];
                if (depth_of(L) < d) continue;   // a leaf: cannot be more frequent than an internal node's string
                if (!(c0 < distinct(L, 0) || c1 < distinct(L, 1))) continue;            // link_more_frequent (:396)
            }
            if (!(clhost::anchor_weight(prm.score, c0, c1, (uint64_t)d + 1, (uint64_t)d + 1) > 0.0)) continue;   // match_finder.hpp:162
            mine.push_back(Kept{((uint64_t)close_rank[P] << 32) | x, Found{C, d + 1, {c0, c1}}});
        }
        std::lock_guard<std::mutex> lock(kept_mutex);
        kept_parts.push_back(std::move(mine));
    }, 4096);
    std::vector<Kept> kept;
    for (auto& part : kept_parts) kept.insert(kept.end(), part.begin(), part.end());
    std::sort(kept.begin(), kept.end(), [](const Kept& x, const Kept& y) { return x.key < y.key; });
    std::vector<Found> matches(kept.size());
    for (size_t i = 0; i < kept.size(); ++i) matches[i] = kept[i].f;
    if (st) { st->n_candidates = n_nodes ? n_nodes - 1 : 0; st->query_ms = ms_since(t0); }

    // ---- walk the matches out (esa.hpp:610-665, match_finder.hpp:186-205)
    t0 = clk::now();
    const uint64_t m = matches.size();
    std::vector<uint64_t> w1(m), w2(m);
    out.count1.resize(m); out.count2.resize(m); out.full_length.resize(m);
    out.set_off1.assign(m + 1, 0); out.set_off2.assign(m + 1, 0);
    // pass 1: the leaves that start a walk (first occurrence of a (graph, start node) in suffix-array order)
    std::vector<uint64_t> leaf_off(m + 1, 0);
    for (uint64_t s = 0; s < m; ++s) leaf_off[s + 1] = leaf_off[s] + (node_r[matches[s].node] - node_l[matches[s].node] + 1);
    ClRawVec<uint8_t> starts(leaf_off[m]);
    cl_parallel_for(m, [&](uint64_t b, uint64_t e) {
        std::vector<uint32_t> stamp[2];
        for (int c = 0; c < 2; ++c) stamp[c].assign(T.n_ids[c], kNone);
        for (uint64_t s = b; s < e; ++s) {
            const struct { uint32_t l, r; } nd{node_l[matches[s].node], node_r[matches[s].node]};
            uint64_t k1 = 0, k2 = 0;
            for (uint32_t i = nd.l; i <= nd.r; ++i) {
                const uint32_t pos = sa[i];
                const int c = T.comp(pos);
                uint32_t& seen = stamp[c][T.id[pos]];
                const bool first = seen != (uint32_t)s;
                seen = (uint32_t)s;
                starts[leaf_off[s] + (i - nd.l)] = first ? (uint8_t)(1 + c) : 0;
                if (first) { if (c) ++k2; else ++k1; }
            }
            w1[s] = k1; w2[s] = k2;
        }
    }, 256);
    for (uint64_t s = 0; s < m; ++s) {
        out.set_off1[s + 1] = out.set_off1[s] + w1[s];
        out.set_off2[s + 1] = out.set_off2[s] + w2[s];
        out.count1[s] = w1[s];
        out.count2[s] = w2[s];
        out.full_length[s] = matches[s].length;
    }
    std::vector<uint64_t> node_off1(m + 1, 0), node_off2(m + 1, 0);
    for (uint64_t s = 0; s < m; ++s) {
        node_off1[s + 1] = node_off1[s] + w1[s] * matches[s].length;
        node_off2[s + 1] = node_off2[s] + w2[s] * matches[s].length;
    }
    out.walk_off1.resize(out.set_off1[m] + 1);
    out.walk_off2.resize(out.set_off2[m] + 1);
    out.nodes1.resize(node_off1[m]);
    out.nodes2.resize(node_off2[m]);
    out.walk_off1[out.set_off1[m]] = node_off1[m];
    out.walk_off2[out.set_off2[m]] = node_off2[m];
    cl_parallel_for(m, [&](uint64_t b, uint64_t e) {
        for (uint64_t s = b; s < e; ++s) {
            const struct { uint32_t l, r; } nd{node_l[matches[s].node], node_r[matches[s].node]};
            const uint32_t len = matches[s].length;
            uint64_t a1 = out.set_off1[s], a2 = out.set_off2[s], p1 = node_off1[s], p2 = node_off2[s];
            for (uint32_t i = nd.l; i <= nd.r; ++i) {
                const uint8_t f = starts[leaf_off[s] + (i - nd.l)];
                if (!f) continue;
                const uint32_t* src = T.id.data() + sa[i];
                if (f == 1) { out.walk_off1[a1++] = p1; std::memcpy(out.nodes1.data() + p1, src, (size_t)len * 4); p1 += len; }
                else        { out.walk_off2[a2++] = p2; std::memcpy(out.nodes2.data() + p2, src, (size_t)len * 4); p2 += len; }
            }
        }
    }, 256);
    if (st) st->walk_ms = ms_since(t0);
    (void)g1; (void)g2;
    return CL_OK;
}

}  // namespace

extern "C" {

void cl_match_params_default(cl_match_params* p) {
    if (!p) return;
    p->max_count = 3000;
    p->use_color_set_size = 1;
    cl_chain_params_default(&p->score);
}

int cl_find_matches(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_params* prm, cl_owned_match_sets** out,
                    cl_match_stats* stats) {
    return cl_find_matches_hooked(ctx, g1, g2, prm, out, stats, nullptr);
}

}  // extern "C"

// cl_find_matches with a callback between its device half (suffix array, LCP) and its host half (interval tree, queries, walks): cl_merge
// starts work there that would get in the device half's way (page-locking the traceback's download area stalls every HIP call of the process)
int cl_find_matches_hooked(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_params* prm, cl_owned_match_sets** out,
                           cl_match_stats* stats, const std::function<void()>* after_device_half) {
    cl_bind_device(ctx);
    if (!ctx || !g1 || !g2 || !prm || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (stats) *stats = cl_match_stats{};
    auto t0 = clk::now();
    JoinedText T;
    int rc = T.build(*g1, *g2);
    if (rc) { cl_set_error(ctx, "cl_find_matches: malformed graph (path node or sentinel id out of range, or joined text >= 2^31)"); return rc; }
    const uint32_t n = (uint32_t)T.text.size();
    const double text_ms = ms_since(t0);
    t0 = clk::now();
    ClRawVec<uint32_t> sa(n), lcp(n), isa(n);
    ClSuffixStats ss;
    if ((rc = cl_match_suffix_array(ctx, T.text.data(), n, sa.data(), lcp.data(), isa.data(), &ss))) return rc;
    const double suffix_wall_ms = ms_since(t0);
    if (after_device_half) (*after_device_half)();
    if (stats) { stats->text_length = n; stats->doubling_rounds = ss.rounds; stats->sa_ms = ss.sort_ms; stats->lcp_ms = ss.lcp_ms; }
    std::unique_ptr<cl_owned_match_sets> o(new cl_owned_match_sets());
    if ((rc = matches_from_esa(*g1, *g2, *prm, T, sa.data(), lcp.data(), isa.data(), *o, stats))) return rc;
    if (stats) { stats->text_ms = text_ms; stats->suffix_wall_ms = suffix_wall_ms; }
    *out = o.release();
    return CL_OK;
}

extern "C" {

int cl_match_joined_text(const cl_base_graph* g1, const cl_base_graph* g2, uint8_t** text_out, uint64_t* n_out) {
    if (!g1 || !g2 || !text_out || !n_out) return CL_ERR_INVALID_ARGUMENT;
    JoinedText T;
    int rc = T.build(*g1, *g2);
    if (rc) return rc;
    *n_out = T.text.size();
    *text_out = (uint8_t*)malloc(T.text.size());
    if (!*text_out) return CL_ERR_OUT_OF_MEMORY;
    std::memcpy(*text_out, T.text.data(), T.text.size());
    return CL_OK;
}

int cl_suffix_array_lcp(cl_context* ctx, const uint8_t* text, uint64_t n, uint32_t* sa, uint32_t* lcp, uint32_t* isa, uint32_t* rounds_out) {
    cl_bind_device(ctx);
    if (!ctx || (n && (!text || !sa || !lcp || !isa)) || n >= 0x7FFFFFFFull) return CL_ERR_INVALID_ARGUMENT;
    ClSuffixStats ss;
    int rc = cl_match_suffix_array(ctx, text, (uint32_t)n, sa, lcp, isa, &ss);
    if (rounds_out) *rounds_out = ss.rounds;
    return rc;
}

int cl_matches_from_suffix_array(const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_params* prm, const uint32_t* sa,
                                 const uint32_t* lcp, uint64_t n, cl_owned_match_sets** out, cl_match_stats* stats) {
    if (!g1 || !g2 || !prm || !sa || !lcp || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (stats) *stats = cl_match_stats{};
    JoinedText T;
    int rc = T.build(*g1, *g2);
    if (rc) return rc;
    if (n != T.text.size()) return CL_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> isa(n);
    for (uint64_t p = 0; p < n; ++p) {
        if (sa[p] >= n) return CL_ERR_INVALID_ARGUMENT;
        isa[sa[p]] = (uint32_t)p;
    }
    if (stats) stats->text_length = n;
    std::unique_ptr<cl_owned_match_sets> o(new cl_owned_match_sets());
    if ((rc = matches_from_esa(*g1, *g2, *prm, T, sa, lcp, isa.data(), *o, stats))) return rc;
    *out = o.release();
    return CL_OK;
}

}  // extern "C"
