// cl_fuse_api.cpp — fuse (include/centrolign/fuse.hpp:46-152): merge the second graph of a merge into the first along their
// alignment, what Core::do_execution does with the result of Core::align (include/centrolign/core.hpp:366-388), and
// cl_merge, the whole loop body of do_execution for one merge: find_matches -> align -> fuse.
//
// Host code, linear in the graphs and the alignment.  BaseGraph::add_node / add_edge append (src/graph.cpp:221-229), so the
// fused graph's node ids and the ORDER of every adjacency list follow from the order of the reference's calls; that order
// is kept, because the next merge's reachability tables, topological orders and predecessor tie-breaks read the lists in
// order.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <thread>
#include <vector>

#include "cl_internal.hpp"


extern "C" {

void cl_owned_base_graph_view(const cl_owned_base_graph* g, cl_base_graph* v) {
    v->n_nodes = g->label.size();
    v->label = g->label.data();
    v->next_off = g->next_off.data(); v->next_idx = g->next_idx.data();
    v->prev_off = g->prev_off.data(); v->prev_idx = g->prev_idx.data();
    v->n_paths = g->path_off.size() - 1;
    v->path_off = g->path_off.data(); v->path_nodes = g->path_nodes.data();
    v->src_id = g->src_id; v->snk_id = g->snk_id;
}

void cl_owned_base_graph_free(cl_owned_base_graph* g) { delete g; }

int cl_fuse(const cl_base_graph* dest, const cl_base_graph* source, const uint64_t* pairs, uint64_t n_pairs, cl_owned_base_graph** out) {
    if (!dest || !source || (n_pairs && !pairs) || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const uint64_t n1 = dest->n_nodes, n2 = source->n_nodes;
    const uint64_t gap = ~(uint64_t)0;
    if (n1 == 0 || n2 == 0 || dest->src_id >= n1 || dest->snk_id >= n1 || source->src_id >= n2 || source->snk_id >= n2) return CL_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < n_pairs; ++i)
        if ((pairs[2 * i] != gap && pairs[2 * i] >= n1) || (pairs[2 * i + 1] != gap && pairs[2 * i + 1] >= n2)) return CL_ERR_INVALID_ARGUMENT;
    // record match nodes (fuse.hpp:58-66), join the sentinels (:68-70), add the unmatched nodes (:72-77)
    std::vector<uint8_t> label(dest->label, dest->label + n1);
    std::vector<uint64_t> trans(n2, gap);
    for (uint64_t i = 0; i < n_pairs; ++i) {
        const uint64_t a = pairs[2 * i], b = pairs[2 * i + 1];
        if (a != gap && b != gap && dest->label[a] == source->label[b]) trans[b] = a;
    }
    trans[source->src_id] = dest->src_id;
    trans[source->snk_id] = dest->snk_id;
    for (uint64_t b = 0; b < n2; ++b) {
        if (trans[b] != gap) continue;
        trans[b] = label.size();
        label.push_back(source->label[b]);
    }
    const uint64_t n = label.size();
    if (n >= 0xFFFFFFFFull) return CL_ERR_INVALID_ARGUMENT;
    // The reference appends edges to per-node lists (BaseGraph::add_edge, src/graph.cpp:226-229) in two phases; here the appended
    // edges are kept as one list in call order and the lists are put together at the end: next(v) = dest's next(v), then the
    // appended edges out of v in call order; previous(w) likewise.
    std::vector<std::pair<uint32_t, uint32_t>> added;
    // phase 1 — substitution edges from the destination graph (:83-108): a mismatched pair hangs between the nearest aligned
    // destination nodes to its right and to its left
    {
        std::vector<uint64_t> left_of(n_pairs), right_of(n_pairs);   // nearest index with node_id1 != gap, exclusive
        uint64_t last = gap;
        for (uint64_t i = 0; i < n_pairs; ++i) { left_of[i] = last; if (pairs[2 * i] != gap) last = i; }
        last = gap;
        for (uint64_t i = n_pairs; i-- > 0;) { right_of[i] = last; if (pairs[2 * i] != gap) last = i; }
        for (uint64_t i = 0; i < n_pairs; ++i) {
            const uint64_t a = pairs[2 * i], b = pairs[2 * i + 1];
            if (a == gap || b == gap || dest->label[a] == source->label[b]) continue;
            if (right_of[i] != gap) added.emplace_back((uint32_t)trans[b], (uint32_t)pairs[2 * right_of[i]]);
            if (left_of[i] != gap) added.emplace_back((uint32_t)pairs[2 * left_of[i]], (uint32_t)trans[b]);
        }
    }
    // the phase-1 edges grouped by their tail (stable): what next(v) holds beyond dest's list when phase 2 looks at v
    const uint64_t n_sub = added.size();
    std::vector<uint64_t> sub_off(n + 1, 0);
    for (uint64_t e = 0; e < n_sub; ++e) ++sub_off[added[e].first + 1];
    for (uint64_t v = 0; v < n; ++v) sub_off[v + 1] += sub_off[v];
    std::vector<uint32_t> sub_to(n_sub);
    {
        std::vector<uint64_t> fill(sub_off.begin(), sub_off.end() - 1);
        for (uint64_t e = 0; e < n_sub; ++e) sub_to[fill[added[e].first]++] = added[e].second;
    }
    // phase 2 — the source's edges that are not there yet (:114-129).  The reference tests membership in next(new_id) as it stood
    // BEFORE this source node's own additions: dest's list, the phase-1 edges out of it, and what phase 2 appended for EARLIER
    // source nodes that map to the same node (in a valid alignment only the two sentinels collect several source nodes)
    std::vector<uint8_t> hits(n, 0);   // source nodes per fused node, saturating at 2
    for (uint64_t b = 0; b < n2; ++b) if (hits[trans[b]] < 2) ++hits[trans[b]];
    std::unordered_map<uint64_t, std::vector<uint32_t>> earlier;   // phase-2 heads appended so far out of a node several source nodes map to
    for (uint64_t b = 0; b < n2; ++b) {
        const uint64_t v = trans[b];
        const uint32_t* d0 = v < n1 ? dest->next_idx + dest->next_off[v] : nullptr;
        const uint32_t* d1 = v < n1 ? dest->next_idx + dest->next_off[v + 1] : nullptr;
        const uint32_t* s0 = sub_to.data() + sub_off[v];
        const uint32_t* s1 = sub_to.data() + sub_off[v + 1];
        std::vector<uint32_t>* shared = hits[v] > 1 ? &earlier[v] : nullptr;
        const size_t seen = shared ? shared->size() : 0;
        for (uint64_t e = source->next_off[b]; e < source->next_off[b + 1]; ++e) {
            const uint32_t w = (uint32_t)trans[source->next_idx[e]];
            bool have = std::find(d0, d1, w) != d1 || std::find(s0, s1, w) != s1;
            if (!have && shared) have = std::find(shared->begin(), shared->begin() + seen, w) != shared->begin() + seen;
            if (have) continue;
            added.emplace_back((uint32_t)v, w);
            if (shared) shared->push_back(w);
        }
    }
    std::unique_ptr<cl_owned_base_graph> g(new cl_owned_base_graph());
    g->label = std::move(label);
    for (int dir = 0; dir < 2; ++dir) {
        const uint64_t* doff = dir ? dest->prev_off : dest->next_off;
        const uint32_t* didx = dir ? dest->prev_idx : dest->next_idx;
        std::vector<uint64_t>& off = dir ? g->prev_off : g->next_off;
        std::vector<uint32_t>& idx = dir ? g->prev_idx : g->next_idx;
        off.assign(n + 1, 0);
        for (uint64_t v = 0; v < n1; ++v) off[v + 1] = doff[v + 1] - doff[v];
        for (const auto& e : added) ++off[(dir ? e.second : e.first) + 1];
        for (uint64_t v = 0; v < n; ++v) off[v + 1] += off[v];
        idx.resize(off[n]);
        std::vector<uint64_t> fill(off.begin(), off.end() - 1);
        for (uint64_t v = 0; v < n1; ++v)
            for (uint64_t e = doff[v]; e < doff[v + 1]; ++e) idx[fill[v]++] = didx[e];
        for (const auto& e : added) idx[fill[dir ? e.second : e.first]++] = dir ? e.first : e.second;
    }
    // the paths: the destination's, then the source's translated (:137-143)
    g->path_off.assign(1, 0);
    for (uint64_t p = 0; p < dest->n_paths; ++p) {
        g->path_nodes.insert(g->path_nodes.end(), dest->path_nodes + dest->path_off[p], dest->path_nodes + dest->path_off[p + 1]);
        g->path_off.push_back(g->path_nodes.size());
    }
    for (uint64_t p = 0; p < source->n_paths; ++p) {
        for (uint64_t i = source->path_off[p]; i < source->path_off[p + 1]; ++i) {
            if (source->path_nodes[i] >= n2) return CL_ERR_INVALID_ARGUMENT;
            g->path_nodes.push_back((uint32_t)trans[source->path_nodes[i]]);
        }
        g->path_off.push_back(g->path_nodes.size());
    }
    g->src_id = dest->src_id;   // next_problem.tableau = subproblem1.tableau (core.hpp:388)
    g->snk_id = dest->snk_id;
    *out = g.release();
    return CL_OK;
}

int cl_leaf_intrinsic_scale(cl_context* ctx, const cl_base_graph* leaf, const cl_match_params* mp, const cl_anchor_params* ap, double* scale_out) {
    cl_bind_device(ctx);
    if (!ctx || !leaf || !mp || !ap || !scale_out) return CL_ERR_INVALID_ARGUMENT;
    if (leaf->n_nodes == 0 || leaf->src_id >= leaf->n_nodes || leaf->snk_id >= leaf->n_nodes) return CL_ERR_INVALID_ARGUMENT;
    // the leaf against itself, the second copy under its own sentinel characters (src/core.cpp:128-133)
    std::vector<uint8_t> lab1(leaf->label, leaf->label + leaf->n_nodes), lab2(lab1);
    lab1[leaf->src_id] = 5; lab1[leaf->snk_id] = 6;
    lab2[leaf->src_id] = 7; lab2[leaf->snk_id] = 8;
    cl_base_graph a = *leaf, b = *leaf;
    a.label = lab1.data();
    b.label = lab2.data();
    cl_owned_match_sets* ms = nullptr;
    int rc = cl_find_matches(ctx, &a, &b, mp, &ms, nullptr);
    if (rc) return rc;
    cl_match_sets v;
    cl_owned_match_sets_view(ms, &v);
    // the main-diagonal subset (:135-148): one set per graph-1 walk, matched to itself, counts and full length retained
    const uint64_t n_walks = v.set_off1[v.n_sets];
    std::vector<uint64_t> set_off(n_walks + 1), count1(n_walks), count2(n_walks), full_length(n_walks);
    for (uint64_t w = 0; w <= n_walks; ++w) set_off[w] = w;
    for (uint64_t s = 0; s < v.n_sets; ++s)
        for (uint64_t w = v.set_off1[s]; w < v.set_off1[s + 1]; ++w) { count1[w] = v.count1[s]; count2[w] = v.count2[s]; full_length[w] = v.full_length[s]; }
    cl_match_sets diag{n_walks, set_off.data(), v.walk_off1, v.nodes1, set_off.data(), v.walk_off1, v.nodes1, count1.data(), count2.data(), full_length.data()};
    // Anchorer::estimate_score_scale on (graph, graph) (:150-156; a leaf has one path, its ChainMerge is that path)
    rc = cl_estimate_score_scale(ctx, &a, &a, &diag, ap, scale_out);
    cl_owned_match_sets_free(ms);
    return rc;
}

void cl_merge_params_default(cl_merge_params* p) {
    if (!p) return;
    cl_match_params_default(&p->match);
    cl_core_align_params_default(&p->align);
}

void cl_merge_result_free(cl_merge_result* r) {
    if (!r) return;
    cl_core_align_result_free(&r->align);
    cl_owned_base_graph_free(r->fused);
    memset(r, 0, sizeof(*r));
}

// internal_fuse (include/centrolign/fuse.hpp:144-247): the nodes of ONE graph that the alignments pair up — transitively — become one node
// per label; the graph that comes out may have cycles (Core::apply_bonds, src/core.cpp:631-636: the tandem duplications of a cyclised
// alignment).  Node numbering: UnionFind groups in ascending order of their final head (union by rank as union_find.hpp:55-72 does it, the
// pairs in the given order), inside a group one node per label in ascending label order; edges in the order a walk over the old nodes and
// their next lists first meets them; paths translated node by node.
int cl_internal_fuse(const cl_base_graph* g, const uint64_t* pairs, uint64_t n_pairs, cl_owned_base_graph** out, uint64_t* trans_out) {
    if (!g || (n_pairs && !pairs) || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const uint64_t n = g->n_nodes, gap = ~(uint64_t)0;
    if (n == 0 || g->src_id >= n || g->snk_id >= n) return CL_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < 2 * n_pairs; ++i) if (pairs[i] != gap && pairs[i] >= n) return CL_ERR_INVALID_ARGUMENT;
    std::vector<uint64_t> head(n), rank(n, 0);
    for (uint64_t v = 0; v < n; ++v) head[v] = v;
    auto find = [&](uint64_t i) {   // union_find.hpp:42-53: the path is pointed at the root, except its last element
        std::vector<uint64_t> path;
        while (head[i] != i) { path.push_back(i); i = head[i]; }
        for (size_t p = 1; p < path.size(); ++p) head[path[p - 1]] = i;
        return i;
    };
    for (uint64_t i = 0; i < n_pairs; ++i) {
        const uint64_t a = pairs[2 * i], b = pairs[2 * i + 1];
        if (a == gap || b == gap) continue;
        const uint64_t ha = find(a), hb = find(b);
        if (ha == hb) continue;
        if (rank[ha] > rank[hb]) head[hb] = ha;
        else { head[ha] = hb; if (rank[hb] == rank[ha]) ++rank[hb]; }
    }
    std::vector<std::vector<uint64_t>> groups(n);
    for (uint64_t v = 0; v < n; ++v) groups[find(v)].push_back(v);
    std::unique_ptr<cl_owned_base_graph> f(new cl_owned_base_graph());
    std::vector<uint64_t> trans(n, gap);
    for (uint64_t hd = 0; hd < n; ++hd) {
        const auto& grp = groups[hd];
        if (grp.empty()) continue;
        // std::map<char, ...> (fuse.hpp:176): ascending (signed) label
        std::vector<std::pair<int, uint64_t>> by_label;
        for (uint64_t v : grp) by_label.emplace_back((int)(signed char)g->label[v], v);
        std::stable_sort(by_label.begin(), by_label.end(), [](const std::pair<int, uint64_t>& x, const std::pair<int, uint64_t>& y) { return x.first < y.first; });
        for (size_t i = 0; i < by_label.size(); ++i) {
            if (i == 0 || by_label[i].first != by_label[i - 1].first) f->label.push_back((uint8_t)by_label[i].first);
            trans[by_label[i].second] = f->label.size() - 1;
        }
    }
    const uint64_t m = f->label.size();
    std::vector<std::vector<uint32_t>> next(m), prev(m);
    {
        std::vector<std::unordered_set<uint64_t>> seen(m);
        for (uint64_t v = 0; v < n; ++v) {
            const uint64_t fv = trans[v];
            for (uint64_t e = g->next_off[v]; e < g->next_off[v + 1]; ++e) {
                const uint64_t fw = trans[g->next_idx[e]];
                if (seen[fv].insert(fw).second) { next[fv].push_back((uint32_t)fw); prev[fw].push_back((uint32_t)fv); }
            }
        }
    }
    f->next_off.assign(1, 0);
    f->prev_off.assign(1, 0);
    for (uint64_t v = 0; v < m; ++v) {
        f->next_idx.insert(f->next_idx.end(), next[v].begin(), next[v].end());
        f->next_off.push_back(f->next_idx.size());
        f->prev_idx.insert(f->prev_idx.end(), prev[v].begin(), prev[v].end());
        f->prev_off.push_back(f->prev_idx.size());
    }
    f->path_off.assign(1, 0);
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        for (uint64_t i = g->path_off[p]; i < g->path_off[p + 1]; ++i) f->path_nodes.push_back((uint32_t)trans[g->path_nodes[i]]);
        f->path_off.push_back(f->path_nodes.size());
    }
    f->src_id = trans[g->src_id];
    f->snk_id = trans[g->snk_id];
    if (trans_out) memcpy(trans_out, trans.data(), n * sizeof(uint64_t));
    *out = f.release();
    return CL_OK;
}

int cl_merge(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_merge_params* prm, cl_merge_result* out) {
    cl_bind_device(ctx);
    if (!ctx || !g1 || !g2 || !prm || !out) return CL_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    if (g1->n_nodes == 0 || g2->n_nodes == 0 || g1->src_id >= g1->n_nodes || g1->snk_id >= g1->n_nodes || g2->src_id >= g2->n_nodes ||
        g2->snk_id >= g2->n_nodes) {
        cl_set_error(ctx, "cl_merge: sentinel ids out of range");
        return CL_ERR_INVALID_ARGUMENT;
    }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return (float)std::chrono::duration<double, std::milli>(now() - t).count(); };
    // reassign_sentinels(…, 5, 6) / (…, 7, 8) (core.hpp:283-284): on private copies of the label arrays
    std::vector<uint8_t> lab1(g1->label, g1->label + g1->n_nodes), lab2(g2->label, g2->label + g2->n_nodes);
    lab1[g1->src_id] = 5; lab1[g1->snk_id] = 6;
    lab2[g2->src_id] = 7; lab2[g2->snk_id] = 8;
    cl_base_graph a = *g1, b = *g2;
    a.label = lab1.data();
    b.label = lab2.data();
    auto t0 = now();
    // the chaining DPs bring their per-pair query results back through the context's page-locked area: up to 28 bytes per match pair and
    // (chain of graph 1, chain of graph 2) combination — 0.9 GB at the root of a 10-sequence tree.  Locking that many pages takes ≈ 0.25 s
    // the first time a context needs them: do it beside the match finding instead of in front of the first traceback
    const uint64_t want_pinned = (uint64_t)prm->align.anchor.max_num_match_pairs * 28ull * std::max<uint64_t>(1, g1->n_paths) * std::max<uint64_t>(1, g2->n_paths);
    // — and beside the HOST half of it: while pages are being locked every HIP call of the process stalls (the suffix sort of the root took
    // 226 ms instead of 20 when the two ran side by side)
    std::thread pin;
    const std::function<void()> start_pin = [&] {
        if (want_pinned > ctx->pinned_bytes && want_pinned <= (8ull << 30)) {
            const int device = ctx->device;
            pin = std::thread([ctx, want_pinned, device] { (void)hipSetDevice(device); (void)cl_pinned(ctx, want_pinned); });
        }
    };
    // the PathMerge tables of the alignment need the graphs only: built beside the match finding
    ClPathMergeTables tables;
    tables.start(&a, &b, prm->align.anchor.chaining_algorithm_plus_one == 2);
    cl_owned_match_sets* ms = nullptr;
    cl_match_stats mst;
    int rc = cl_find_matches_hooked(ctx, &a, &b, &prm->match, &ms, &mst, &start_pin);
    if (pin.joinable()) pin.join();
    if (rc) return rc;
    out->match_ms = ms_since(t0);
    if (getenv("CL_CHAIN_TIMING"))
        fprintf(stderr, "[cl_merge] find_matches %.1f ms: text %.1f, device half %.1f (suffix sort %.1f, lcp %.1f), interval tree %.1f, queries %.1f, walks %.1f\n",
                out->match_ms, mst.text_ms, mst.suffix_wall_ms, mst.sa_ms, mst.lcp_ms, mst.tree_ms, mst.query_ms, mst.walk_ms);
    cl_match_sets view;
    cl_owned_match_sets_view(ms, &view);
    out->n_match_sets = view.n_sets;
    t0 = now();
    rc = cl_core_align_prepared(ctx, &a, &b, &view, &prm->align, &out->align, &tables);
    cl_owned_match_sets_free(ms);
    if (rc) return rc;
    out->align_ms = ms_since(t0);
    t0 = now();
    rc = cl_fuse(&a, &b, out->align.alignment.pairs, out->align.alignment.n_pairs, &out->fused);
    out->fuse_ms = ms_since(t0);
    if (rc) { cl_set_error(ctx, "cl_fuse failed"); cl_merge_result_free(out); return rc; }
    return CL_OK;
}

}  // extern "C"
