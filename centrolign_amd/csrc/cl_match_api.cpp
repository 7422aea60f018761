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

struct Node {
    uint32_t l, r, depth, parent;
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
    // ---- the LCP-interval tree, bottom-up (the stack pass of esa.hpp:436-494), with Hui's duplicate counts
    std::vector<Node> nodes;
    std::vector<uint32_t> closed;            // node ids in the order the pass closes them (children before parents)
    std::vector<uint32_t> own_dup[2];        // duplicates whose LCA is the node
    std::vector<uint32_t> leaf_parent(n);    // deepest internal node above every leaf
    std::vector<uint32_t> stack;
    std::vector<uint32_t> prev_occ[2];
    for (int c = 0; c < 2; ++c) prev_occ[c].assign(T.n_ids[c], kNone);
    std::vector<uint32_t> before[2];         // leaves of graph c at suffix-array positions < p
    for (int c = 0; c < 2; ++c) before[c].resize((size_t)n + 1);
    nodes.reserve(n);
    closed.reserve(n);
    auto new_node = [&](uint32_t l, uint32_t depth) {
        nodes.push_back(Node{l, 0, depth, kNone});
        own_dup[0].push_back(0);
        own_dup[1].push_back(0);
        return (uint32_t)(nodes.size() - 1);
    };
    auto visit_leaf = [&](uint32_t i) {
        const uint32_t pos = sa[i];
        const int c = T.comp(pos);
        before[0][i + 1] = before[0][i] + (c == 0);
        before[1][i + 1] = before[1][i] + (c == 1);
        uint32_t& prev = prev_occ[c][T.id[pos]];
        if (prev != kNone) {
            // deepest open interval that also holds the previous occurrence: interval starts grow along the stack
            size_t lo = 0, hi = stack.size() - 1;
            while (lo < hi) {
                const size_t mid = (lo + hi + 1) / 2;
                if (nodes[stack[mid]].l <= prev) lo = mid; else hi = mid - 1;
            }
            ++own_dup[c][stack[lo]];
        }
        prev = i;
    };
    stack.push_back(new_node(0, 0));
    before[0][0] = before[1][0] = 0;
    leaf_parent[0] = stack[0];
    visit_leaf(0);
    // the pass is a chain of cache misses — id[sa[i]], then prev_occ[...][that id] — on addresses known far ahead: fetch them early
    constexpr uint32_t kAheadId = 48, kAheadOcc = 24;
    for (uint32_t i = 1; i < n; ++i) {
        if (i + kAheadId < n) __builtin_prefetch(&T.id[sa[i + kAheadId]]);
        if (i + kAheadOcc < n) { const uint32_t pa = sa[i + kAheadOcc]; __builtin_prefetch(&prev_occ[T.comp(pa)][T.id[pa]], 1); }
        uint32_t last = kNone, left = i - 1;
        bool fresh = true;
        while (nodes[stack.back()].depth > lcp[i]) {
            const uint32_t v = stack.back();
            stack.pop_back();
            nodes[v].r = i - 1;
            closed.push_back(v);
            left = nodes[v].l;
            fresh = false;
            if (nodes[stack.back()].depth >= lcp[i]) { nodes[v].parent = stack.back(); last = kNone; }
            else last = v;
        }
        if (nodes[stack.back()].depth < lcp[i]) {
            const uint32_t u = new_node(left, lcp[i]);
            stack.push_back(u);
            if (last != kNone) nodes[last].parent = u;
            if (fresh) leaf_parent[i - 1] = u;   // the interval opens at leaf i-1
        }
        leaf_parent[i] = stack.back();
        visit_leaf(i);
    }
    while (!stack.empty()) {
        const uint32_t v = stack.back();
        stack.pop_back();
        nodes[v].r = n - 1;
        closed.push_back(v);
        if (!stack.empty()) nodes[v].parent = stack.back();
    }
    // subtree totals of the duplicates (src/esa.cpp:235-300), then distinct start nodes = leaves - duplicates
    std::vector<uint32_t>& dup0 = own_dup[0];
    std::vector<uint32_t>& dup1 = own_dup[1];
    for (size_t k = 0; k < closed.size(); ++k) {
        if (k + 32 < closed.size()) __builtin_prefetch(&nodes[closed[k + 32]]);
        if (k + 16 < closed.size()) {
            const uint32_t pa = nodes[closed[k + 16]].parent;
            if (pa != kNone) { __builtin_prefetch(&dup0[pa], 1); __builtin_prefetch(&dup1[pa], 1); }
        }
        const uint32_t v = closed[k];
        if (nodes[v].parent != kNone) { dup0[nodes[v].parent] += dup0[v]; dup1[nodes[v].parent] += dup1[v]; }
    }
    auto distinct = [&](uint32_t v, int c) -> uint64_t {
        return (uint64_t)(before[c][nodes[v].r + 1] - before[c][nodes[v].l]) - (c ? dup1[v] : dup0[v]);
    };
    if (st) { st->n_internal_nodes = nodes.size(); st->tree_ms = ms_since(t0); }

    // ---- the query (esa.hpp:290-431): every (parent, child) pair is independent.  The reference meets the children grouped under their
    //      parents, parents in closing order, children in closing order: evaluate them all side by side in closing order of the child and
    //      sort the few that pass by (closing rank of the parent, closing rank of the child)
    t0 = clk::now();
    const uint32_t n_nodes = (uint32_t)nodes.size();
    std::vector<uint32_t> close_rank(n_nodes);
    for (uint32_t k = 0; k < n_nodes; ++k) {
        if (k + 32 < n_nodes) __builtin_prefetch(&close_rank[closed[k + 32]], 1);
        close_rank[closed[k]] = k;
    }
    struct Kept { uint64_t key; Found f; };
    std::vector<std::vector<Kept>> kept_parts;
    std::mutex kept_mutex;
    cl_parallel_for(n_nodes, [&](uint64_t b, uint64_t e) {
        std::vector<Kept> mine;
        for (uint64_t x = b; x < e; ++x) {
            const uint32_t C = closed[x], P = nodes[C].parent;
            if (P == kNone) continue;
            const uint64_t c0 = distinct(C, 0);
            if (c0 == 0) continue;                   // esa.hpp:391-393: counts stay 0, total 0
            const uint64_t c1 = distinct(C, 1);
            if (c1 == 0) continue;
            unsigned __int128 total = (unsigned __int128)c0 * c1;   // sat_mult (:399-402)
            if (total > prm.max_count) continue;
            const uint32_t d = nodes[P].depth;
            if (d != 0) {
                if (!(c0 < distinct(P, 0) || c1 < distinct(P, 1))) continue;            // parent_more_frequent (:397)
                // the sibling under the parent's suffix link (:352-362): drop the first character of C's string
                const uint32_t q = isa[sa[nodes[C].l] + 1];
                uint32_t L = leaf_parent[q];
                while (nodes[L].parent != kNone && nodes[nodes[L].parent].depth >= d) L = nodes[L].parent;
                if (nodes[L].depth < d) continue;   // a leaf: cannot be more frequent than an internal node's string
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
    for (uint64_t s = 0; s < m; ++s) leaf_off[s + 1] = leaf_off[s] + (nodes[matches[s].node].r - nodes[matches[s].node].l + 1);
    std::vector<uint8_t> starts(leaf_off[m]);
    cl_parallel_for(m, [&](uint64_t b, uint64_t e) {
        std::vector<uint32_t> stamp[2];
        for (int c = 0; c < 2; ++c) stamp[c].assign(T.n_ids[c], kNone);
        for (uint64_t s = b; s < e; ++s) {
            const Node& nd = nodes[matches[s].node];
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
            const Node& nd = nodes[matches[s].node];
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
    std::vector<uint32_t> sa(n), lcp(n), isa(n);
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
