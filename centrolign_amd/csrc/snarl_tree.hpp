// snarl_tree.hpp — the snarl / chain tree of a (possibly cyclic) sequence graph with source and sink sentinels, as the polishing step of
// cyclisation reads it (InconsistencyIdentifier, include/centrolign/inconsistency_identifier.hpp:66-187): which node pairs bound a snarl,
// how snarls line up in chains, which chains sit inside which snarl, which net graphs have cycles, and the shortest / longest walk through
// every snarl and chain.  Host code.
//
// The reference (include/centrolign/snarls.hpp:130-197) finds snarls through a cactus graph (cactus.hpp:137-200): runs of nodes with one
// way in and one way out are compacted (compacted_graph.hpp:72-96), source and sink are joined into a cycle (chain_cycle_graph.hpp), every
// compacted node becomes an EDGE between the adjacency components of its two sides (adjacency_graph.hpp:71-112), and the
// three-edge-connected components of that graph are the cactus nodes; the cycles of the cactus, found by a depth-first walk from the
// component that holds the source-sink adjacency (cactus.hpp:317-470), are the chains, consecutive edges of a cycle bound a snarl.
// The ids of snarls and chains — which fix the order in which inconsistencies are found and therefore the node numbering of the polished
// graph — follow from the ORDER of those walks, which depends on the order of edge lists only; that order is reproduced here step by
// step.  The three-edge-connected components enter as a partition only (their numbering is never looked at), so they are computed
// differently: the reference ports Tsin's absorb-eject algorithm; here every cut pair of edges is found by cycle-space hashing (a random
// 64-bit word per non-tree edge of a DFS tree, a tree edge's label the XOR over the non-tree edges that span it: two edges are a cut pair
// when their labels agree), the pieces a class of cut pairs leaves get random words of their own, and two adjacency components are
// three-edge-connected when the sums of their pieces' words over all classes agree.
#ifndef CL_SNARL_TREE_HPP
#define CL_SNARL_TREE_HPP

#include <algorithm>
#include <cstdint>
#include <functional>
#include <limits>
#include <queue>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

namespace clsnarl {

constexpr uint64_t kNone = ~(uint64_t)0;

struct GraphView {   // adjacency of a cl_base_graph / cl_owned_base_graph
    uint64_t n = 0;
    const uint64_t* next_off = nullptr;
    const uint32_t* next_idx = nullptr;
    const uint64_t* prev_off = nullptr;
    const uint32_t* prev_idx = nullptr;
    uint64_t src = 0, snk = 0;
    uint64_t outdeg(uint64_t v) const { return next_off[v + 1] - next_off[v]; }
    uint64_t indeg(uint64_t v) const { return prev_off[v + 1] - prev_off[v]; }
    uint64_t next(uint64_t v, uint64_t i) const { return next_idx[next_off[v] + i]; }
    uint64_t prev(uint64_t v, uint64_t i) const { return prev_idx[prev_off[v] + i]; }
};

// ---- three-edge-connected components of an undirected multigraph (edge e joins eu[e] and ev[e]); the graph is two-edge-connected ----
inline std::vector<uint64_t> three_edge_connected_classes(uint64_t n, const std::vector<uint64_t>& eu, const std::vector<uint64_t>& ev) {
    const uint64_t m = eu.size();
    std::vector<uint64_t> off(n + 1, 0), adj(2 * m);
    for (uint64_t e = 0; e < m; ++e) { ++off[eu[e] + 1]; ++off[ev[e] + 1]; }
    for (uint64_t v = 0; v < n; ++v) off[v + 1] += off[v];
    {
        std::vector<uint64_t> fill(off.begin(), off.end() - 1);
        for (uint64_t e = 0; e < m; ++e) { adj[fill[eu[e]]++] = e; adj[fill[ev[e]]++] = e; }
    }
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng * 0x2545F4914F6CDD1Dull; };
    std::vector<uint64_t> label(m, 0), parent_edge(n, kNone), pre(n, kNone), order, acc(n, 0);
    std::vector<char> is_tree(m, 0);
    order.reserve(n);
    for (uint64_t root = 0; root < n; ++root) {
        if (pre[root] != kNone) continue;
        std::vector<std::pair<uint64_t, uint64_t>> stack(1, std::make_pair(root, off[root]));
        pre[root] = order.size();
        order.push_back(root);
        while (!stack.empty()) {
            const uint64_t v = stack.back().first;
            if (stack.back().second == off[v + 1]) { stack.pop_back(); continue; }
            const uint64_t e = adj[stack.back().second++];
            if (e == parent_edge[v] || eu[e] == ev[e]) continue;
            const uint64_t w = eu[e] == v ? ev[e] : eu[e];
            if (pre[w] == kNone) {
                is_tree[e] = 1;
                parent_edge[w] = e;
                pre[w] = order.size();
                order.push_back(w);
                stack.emplace_back(w, off[w]);
            } else if (label[e] == 0 && !is_tree[e]) {   // a non-tree edge, met from its first end
                label[e] = rnd() | 1;
                acc[v] ^= label[e];
                acc[w] ^= label[e];
            }
        }
    }
    // a tree edge's label: XOR of the non-tree edges with exactly one end below it
    for (uint64_t i = order.size(); i-- > 0;) {
        const uint64_t v = order[i], e = parent_edge[v];
        if (e == kNone) continue;
        label[e] = acc[v];
        acc[eu[e] == v ? ev[e] : eu[e]] ^= acc[v];
    }
    // classes of equal labels that hold a cut pair: two or more tree edges, or one tree edge and the single non-tree edge that spans it
    std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> count;   // label -> (tree edges, non-tree edges)
    for (uint64_t e = 0; e < m; ++e) {
        if (eu[e] == ev[e]) continue;
        auto& c = count[label[e]];
        if (is_tree[e]) ++c.first; else ++c.second;
    }
    // walking down the tree, crossing the i-th edge of a class (counted from the root) moves from piece i - 1 to piece i; without a non-tree
    // edge in the class the last piece and piece 0 are one and the same
    struct ClassState { uint64_t seen = 0, word0 = 0, prev_word = 0; };
    std::unordered_map<uint64_t, ClassState> state;
    std::vector<uint64_t> sig(n, 0), delta(n, 0);
    for (uint64_t i = 0; i < order.size(); ++i) {
        const uint64_t v = order[i], e = parent_edge[v];
        if (e == kNone) continue;
        const uint64_t p = eu[e] == v ? ev[e] : eu[e];
        uint64_t d = 0;
        const auto c = count[label[e]];
        if (c.first + c.second >= 2) {
            // the tree edges of a class lie on one root path and are met in that order by the preorder walk
            ClassState& st = state[label[e]];
            if (st.seen == 0) { st.word0 = rnd(); st.prev_word = st.word0; }
            ++st.seen;
            const uint64_t word = (st.seen == c.first && c.second == 0) ? st.word0 : rnd();
            d = word - st.prev_word;
            st.prev_word = word;
        }
        sig[v] = sig[p] + d;
        (void)delta;
    }
    // (sig of a vertex = sum over classes of the word of the piece it lies in, up to a constant per class)
    std::vector<uint64_t> cls(n, kNone);
    std::unordered_map<uint64_t, uint64_t> id_of;
    for (uint64_t v = 0; v < n; ++v) {
        auto it = id_of.find(sig[v]);
        if (it == id_of.end()) it = id_of.emplace(sig[v], id_of.size()).first;
        cls[v] = it->second;
    }
    return cls;
}

struct SnarlTree {
    // snarls ("structures") and chains, numbered as the reference numbers them
    std::vector<std::pair<uint64_t, uint64_t>> boundaries;   // per snarl
    std::vector<uint64_t> snarl_chain;                       // chain_containing
    std::vector<std::vector<uint64_t>> snarl_children;       // chains_inside
    std::vector<std::vector<uint64_t>> chain_snarls;         // structures_inside
    std::vector<uint64_t> chain_parent;                      // structure_containing, kNone for a top-level chain
    std::vector<uint64_t> begins_at, ends_at;                // per node: structure_beginning_at / structure_ending_at
    std::vector<char> net_acyclic, snarl_acyclic, chain_acyclic;
    std::vector<std::pair<uint64_t, uint64_t>> snarl_dist, chain_dist;   // (min, max) walk lengths, max = kNone when unbounded
    std::vector<char> nontrivial_left_boundary;              // the last node of every compacted run

    uint64_t n_snarls() const { return boundaries.size(); }
    uint64_t n_chains() const { return chain_snarls.size(); }

    // TwoDisconnectedStructureTree::postorder (src/structure_tree.cpp:9-49)
    std::vector<std::pair<uint64_t, bool>> postorder() const {
        std::vector<std::pair<uint64_t, bool>> result;
        for (uint64_t c = 0; c < n_chains(); ++c) {
            if (chain_parent[c] != kNone) continue;
            std::vector<std::tuple<uint64_t, bool, bool>> stack;
            stack.emplace_back(c, true, false);
            while (!stack.empty()) {
                if (std::get<2>(stack.back())) {
                    result.emplace_back(std::get<0>(stack.back()), std::get<1>(stack.back()));
                    stack.pop_back();
                } else {
                    std::get<2>(stack.back()) = true;
                    const uint64_t id = std::get<0>(stack.back());
                    const bool is_chain = std::get<1>(stack.back());
                    const auto& kids = is_chain ? chain_snarls[id] : snarl_children[id];
                    for (uint64_t k : kids) stack.emplace_back(k, !is_chain, false);
                }
            }
        }
        return result;
    }

    // NetGraph(graph, structures, struct_id) (structure_tree.hpp:318-377): node 0 = the snarl's first boundary
    struct Net {
        std::vector<std::pair<uint64_t, bool>> label;   // (node id or chain id, is chain)
        std::vector<std::vector<uint64_t>> next;
        std::vector<uint64_t> indeg;
        uint64_t add(uint64_t f, bool c) { label.emplace_back(f, c); next.emplace_back(); indeg.push_back(0); return label.size() - 1; }
        void edge(uint64_t a, uint64_t b) { next[a].push_back(b); ++indeg[b]; }
        bool acyclic() const {   // is_acyclic.hpp
            std::vector<uint64_t> deg(indeg), stack;
            for (uint64_t v = 0; v < deg.size(); ++v) if (deg[v] == 0) stack.push_back(v);
            while (!stack.empty()) {
                const uint64_t v = stack.back();
                stack.pop_back();
                for (uint64_t w : next[v]) if (--deg[w] == 0) stack.push_back(w);
            }
            for (uint64_t d : deg) if (d) return false;
            return true;
        }
    };
    Net net_graph(const GraphView& g, uint64_t snarl) const {
        Net net;
        const uint64_t start = boundaries[snarl].first, end = boundaries[snarl].second;
        std::unordered_map<uint64_t, uint64_t> fwd;
        fwd[start] = net.add(start, false);
        std::vector<uint64_t> stack(1, start);
        while (!stack.empty()) {
            const uint64_t v = stack.back();
            stack.pop_back();
            if (v == end) continue;
            for (uint64_t i = 0; i < g.outdeg(v); ++i) {
                const uint64_t w = g.next(v, i);
                auto it = fwd.find(w);
                if (it != fwd.end()) { net.edge(fwd[v], it->second); continue; }
                const uint64_t s = begins_at[w];
                if (s != kNone && w != end) {
                    const uint64_t chain = snarl_chain[s], id = net.add(chain, true);
                    const uint64_t last = boundaries[chain_snarls[chain].back()].second;
                    fwd[w] = id;
                    fwd[last] = id;
                    net.edge(fwd[v], id);
                    stack.push_back(last);
                } else {
                    const uint64_t id = net.add(w, false);
                    fwd[w] = id;
                    net.edge(fwd[v], id);
                    stack.push_back(w);
                }
            }
        }
        return net;
    }

    bool build(const GraphView& g);
};

inline bool SnarlTree::build(const GraphView& g) {
    const uint64_t n = g.n;
    // ---- CompactedGraph (compacted_graph.hpp:72-96)
    struct CNode { uint64_t front, back; std::vector<uint64_t> next, prev; };
    std::vector<CNode> cn;
    std::vector<uint64_t> front_of(n, kNone);
    for (uint64_t v = 0; v < n; ++v) {
        if (g.indeg(v) != 1 || g.outdeg(g.prev(v, 0)) != 1) {
            uint64_t back = v;
            while (g.outdeg(back) == 1 && g.indeg(g.next(back, 0)) == 1) back = g.next(back, 0);
            front_of[v] = cn.size();
            cn.push_back({v, back, {}, {}});
        }
    }
    for (uint64_t c = 0; c < cn.size(); ++c)
        for (uint64_t i = 0; i < g.outdeg(cn[c].back); ++i) {
            const uint64_t t = front_of[g.next(cn[c].back, i)];
            if (t == kNone) return false;
            cn[c].next.push_back(t);
            cn[t].prev.push_back(c);
        }
    nontrivial_left_boundary.assign(n, 0);
    for (const CNode& c : cn) nontrivial_left_boundary[c.back] = 1;
    uint64_t csrc = kNone, csnk = kNone;
    for (uint64_t c = 0; c < cn.size(); ++c) {
        if (cn[c].front == g.src) csrc = c;
        if (cn[c].back == g.snk) csnk = c;
    }
    if (csrc == kNone || csnk == kNone) return false;
    // ---- ChainCycleGraph: the sink leads to the source
    const uint64_t nc = cn.size();
    auto cyc_next = [&](uint64_t c) -> std::vector<uint64_t> { return c == csnk ? std::vector<uint64_t>(1, csrc) : cn[c].next; };
    auto cyc_prev = [&](uint64_t c) -> std::vector<uint64_t> { return c == csrc ? std::vector<uint64_t>(1, csnk) : cn[c].prev; };
    // ---- AdjacencyGraph (adjacency_graph.hpp:71-112): side 2c = the right side of compacted node c, 2c + 1 its left side
    std::vector<uint64_t> side_comp(2 * nc, kNone);
    uint64_t n_adj = 0;
    for (uint64_t i = 0; i < 2 * nc; ++i) {
        if (side_comp[i] != kNone) continue;
        side_comp[i] = n_adj;
        std::vector<std::pair<uint64_t, bool>> stack(1, std::make_pair(i / 2, (bool)(i % 2)));
        while (!stack.empty()) {
            const uint64_t c = stack.back().first;
            const bool left = stack.back().second;
            stack.pop_back();
            for (uint64_t w : (left ? cyc_prev(c) : cyc_next(c))) {
                const uint64_t j = 2 * w + (left ? 0 : 1);
                if (side_comp[j] == kNone) { side_comp[j] = n_adj; stack.emplace_back(w, !left); }
            }
        }
        ++n_adj;
    }
    struct AEdge { uint64_t target, label; };
    std::vector<std::vector<AEdge>> a_next(n_adj), a_prev(n_adj);
    std::vector<uint64_t> eu(nc), ev(nc);
    for (uint64_t c = 0; c < nc; ++c) {
        const uint64_t right = side_comp[2 * c], left = side_comp[2 * c + 1];
        a_next[left].push_back({right, c});
        a_prev[right].push_back({left, c});
        eu[c] = left; ev[c] = right;
    }
    // ---- cactus nodes = three-edge-connected components of the adjacency graph (cactus.hpp:166-197)
    const std::vector<uint64_t> comp = three_edge_connected_classes(n_adj, eu, ev);
    uint64_t n_cactus = 0;
    for (uint64_t c : comp) n_cactus = std::max(n_cactus, c + 1);
    struct KNode {
        std::vector<uint64_t> next, prev;
        std::vector<std::tuple<uint64_t, uint64_t, uint64_t>> next_origin;   // (adjacency node, edge index there, index in the target's prev list)
        std::vector<uint64_t> prev_origin;                                    // index in the source's next list
    };
    std::vector<KNode> kn(n_cactus);
    uint64_t origin = kNone;
    for (uint64_t a = 0; a < n_adj; ++a) {
        const uint64_t c1 = comp[a];
        for (uint64_t i = 0; i < a_next[a].size(); ++i) {
            const uint64_t c2 = comp[a_next[a][i].target];
            kn[c1].next.push_back(c2);
            kn[c1].next_origin.emplace_back(a, i, kn[c2].prev.size());
            kn[c2].prev.push_back(c1);
            kn[c2].prev_origin.push_back(kn[c1].next.size() - 1);
            if (a_next[a][i].label == csrc) origin = c1;   // the edge of the source run leaves the component of the source-sink adjacency
        }
    }
    if (origin == kNone) return false;
    auto edge_compacted = [&](uint64_t node, bool is_next, uint64_t idx) {
        if (!is_next) { const uint64_t src_node = kn[node].prev[idx]; idx = kn[node].prev_origin[idx]; node = src_node; }
        const auto& o = kn[node].next_origin[idx];
        return a_next[std::get<0>(o)][std::get<1>(o)].label;
    };
    // ---- CactusTree (cactus.hpp:317-600): the cycles, by a depth-first walk over the edge lists (previous edges first) from the origin
    using CEdge = std::tuple<uint64_t, bool, uint64_t>;   // (node, is next, index)
    std::vector<std::vector<CEdge>> cycles;
    {
        std::vector<char> stacked(n_cactus, 0);
        std::vector<std::vector<char>> traversed(n_cactus);
        for (uint64_t k = 0; k < n_cactus; ++k) traversed[k].assign(kn[k].next.size(), 0);
        std::vector<std::pair<std::vector<CEdge>, uint64_t>> stack;
        stack.emplace_back();
        stack[0].first.emplace_back(origin, false, kNone);
        stack[0].second = 0;
        while (!stack.empty()) {
            if (stack.back().second == stack.back().first.size()) { stack.pop_back(); continue; }
            const CEdge e = stack.back().first[stack.back().second++];
            const uint64_t to = std::get<0>(e);
            if (stack.size() != 1) {
                const auto& below = stack[stack.size() - 2];
                const uint64_t from = std::get<0>(below.first[below.second - 1]);
                uint64_t e_src, e_idx;
                if (std::get<1>(e)) { e_src = from; e_idx = std::get<2>(e); }
                else { e_src = to; e_idx = kn[from].prev_origin[std::get<2>(e)]; }
                if (traversed[e_src][e_idx]) continue;
                traversed[e_src][e_idx] = 1;
            }
            if (!stacked[to]) {
                stack.emplace_back();
                auto& rec = stack.back().first;
                for (uint64_t i = 0; i < kn[to].prev.size(); ++i) rec.emplace_back(kn[to].prev[i], false, i);
                for (uint64_t i = 0; i < kn[to].next.size(); ++i) rec.emplace_back(kn[to].next[i], true, i);
                stack.back().second = 0;
                stacked[to] = 1;
            } else {
                cycles.emplace_back();
                auto& cycle = cycles.back();
                uint64_t i = stack.size() - 1;
                while (true) {
                    const CEdge& cur = stack[i].first[stack[i].second - 1];
                    const CEdge& prv = stack[i - 1].first[stack[i - 1].second - 1];
                    cycle.emplace_back(std::get<0>(prv), std::get<1>(cur), std::get<2>(cur));
                    if (std::get<0>(prv) == to) break;
                    --i;
                }
                std::reverse(cycle.begin(), cycle.end());
            }
        }
    }
    if (cycles.empty()) return false;
    std::vector<std::vector<uint64_t>> assigned(n_cactus);
    for (uint64_t k = 0; k < n_cactus; ++k) assigned[k].assign(kn[k].next.size(), kNone);
    uint64_t root_cycle = kNone;
    for (uint64_t i = 0; i < cycles.size(); ++i) {
        if (std::get<0>(cycles[i].front()) == origin) root_cycle = i;
        for (const CEdge& e : cycles[i]) {
            uint64_t node, idx;
            if (std::get<1>(e)) { node = std::get<0>(e); idx = std::get<2>(e); }
            else { node = kn[std::get<0>(e)].prev[std::get<2>(e)]; idx = kn[std::get<0>(e)].prev_origin[std::get<2>(e)]; }
            assigned[node][idx] = i;
        }
    }
    if (root_cycle == kNone) return false;
    // the tree: cactus nodes 0 .. n_cactus - 1, cycle (chain) nodes behind them
    struct TNode { std::vector<CEdge> cycle; uint64_t parent = kNone; std::vector<uint64_t> children; };
    std::vector<TNode> tn(n_cactus + cycles.size());
    for (uint64_t i = 0; i < cycles.size(); ++i) tn[n_cactus + i].cycle = std::move(cycles[i]);
    const uint64_t root = n_cactus + root_cycle;
    {
        std::vector<char> stacked(tn.size(), 0);
        std::vector<uint64_t> stack(1, root);
        stacked[root] = 1;
        while (!stack.empty()) {
            const uint64_t v = stack.back();
            stack.pop_back();
            auto visit = [&](uint64_t w) {
                if (stacked[w]) return;
                tn[v].children.push_back(w);
                tn[w].parent = v;
                stack.push_back(w);
                stacked[w] = 1;
            };
            if (v >= n_cactus) {
                for (const CEdge& e : tn[v].cycle) visit(std::get<0>(e));
            } else {
                for (uint64_t i = 0; i < kn[v].next.size(); ++i) {
                    const uint64_t cyc = assigned[v][i];
                    visit(cyc == kNone ? kn[v].next[i] : n_cactus + cyc);
                }
                for (uint64_t i = 0; i < kn[v].prev.size(); ++i) {
                    const uint64_t cyc = assigned[kn[v].prev[i]][kn[v].prev_origin[i]];
                    visit(cyc == kNone ? kn[v].prev[i] : n_cactus + cyc);
                }
            }
        }
    }
    for (uint64_t v = n_cactus; v < tn.size(); ++v) {
        const uint64_t first = v == root ? origin : tn[v].parent;
        uint64_t i = 0;
        while (i < tn[v].cycle.size() && std::get<0>(tn[v].cycle[i]) != first) ++i;
        if (i == tn[v].cycle.size()) return false;
        std::rotate(tn[v].cycle.begin(), tn[v].cycle.begin() + i, tn[v].cycle.end());
    }
    // ---- SnarlTree::find_2_disc_structures_impl (snarls.hpp:130-197): the snarls in the order of a stack walk down the cactus tree
    std::vector<std::pair<uint64_t, uint64_t>> found;
    auto edge_walk = [&](const CEdge& e) {
        const uint64_t c = edge_compacted(std::get<0>(e), std::get<1>(e), std::get<2>(e));
        std::vector<uint64_t> walk(1, cn[c].front);
        while (walk.back() != cn[c].back) walk.push_back(g.next(walk.back(), 0));
        for (size_t i = 1; i < walk.size(); ++i) found.emplace_back(walk[i - 1], walk[i]);
        return walk;
    };
    {
        std::vector<uint64_t> stack(1, root);
        while (!stack.empty()) {
            const uint64_t v = stack.back();
            stack.pop_back();
            if (v >= n_cactus) {
                const auto& chain = tn[v].cycle;
                auto prev_walk = edge_walk(chain.front());
                for (size_t i = 1; i < chain.size(); ++i) {
                    auto walk = edge_walk(chain[i]);
                    if (std::get<1>(chain[i - 1]) == std::get<1>(chain[i])) {
                        if (std::get<1>(chain[i])) found.emplace_back(prev_walk.back(), walk.front());
                        else found.emplace_back(walk.back(), prev_walk.front());
                    }
                    prev_walk = std::move(walk);
                }
            }
            for (uint64_t w : tn[v].children) stack.push_back(w);
        }
    }
    // ---- TwoDisconnectedStructureTree::initialize (structure_tree.hpp:155-283)
    begins_at.assign(n, kNone);
    ends_at.assign(n, kNone);
    for (const auto& s : found) {
        if (s.first == g.src || s.second == g.snk || s.first == g.snk || s.second == g.src) continue;
        begins_at[s.first] = boundaries.size();
        ends_at[s.second] = boundaries.size();
        boundaries.push_back(s);
    }
    const uint64_t ns = boundaries.size();
    snarl_chain.assign(ns, kNone);
    snarl_children.assign(ns, {});
    for (uint64_t s = 0; s < ns; ++s) {
        if (snarl_chain[s] != kNone) continue;
        const uint64_t chain = chain_snarls.size();
        chain_snarls.emplace_back();
        chain_parent.push_back(kNone);
        std::vector<uint64_t> ids(1, s);
        snarl_chain[s] = chain;
        for (uint64_t at = ends_at[boundaries[s].first]; at != kNone; at = ends_at[boundaries[at].first]) { ids.push_back(at); snarl_chain[at] = chain; }
        std::reverse(ids.begin(), ids.end());
        for (uint64_t at = begins_at[boundaries[s].second]; at != kNone; at = begins_at[boundaries[at].second]) { ids.push_back(at); snarl_chain[at] = chain; }
        chain_snarls[chain] = std::move(ids);
    }
    {
        std::vector<char> traversed(n, 0);
        for (uint64_t s = 0; s < ns; ++s) {
            std::vector<uint64_t> stack(1, boundaries[s].first);
            while (!stack.empty()) {
                const uint64_t v = stack.back();
                stack.pop_back();
                for (uint64_t i = 0; i < g.outdeg(v); ++i) {
                    const uint64_t w = g.next(v, i);
                    if (w == boundaries[s].second || traversed[w]) continue;
                    traversed[w] = 1;
                    const uint64_t inner = begins_at[w];
                    if (inner != kNone) {
                        const uint64_t chain = snarl_chain[inner];
                        chain_parent[chain] = s;
                        snarl_children[s].push_back(chain);
                        const uint64_t last = boundaries[chain_snarls[chain].back()].second;
                        traversed[last] = 1;
                        stack.push_back(last);
                    } else {
                        stack.push_back(w);
                    }
                }
            }
        }
    }
    // ---- acyclicity (snarls.hpp:58-118) and distances (structure_distances.hpp:60-205, StructureDistances<SnarlTree, false>; every node one base)
    net_acyclic.assign(ns, 0);
    snarl_acyclic.assign(ns, 0);
    chain_acyclic.assign(n_chains(), 0);
    snarl_dist.assign(ns, std::make_pair((uint64_t)0, (uint64_t)0));
    chain_dist.assign(n_chains(), std::make_pair((uint64_t)0, (uint64_t)0));
    for (const auto& f : postorder()) {
        if (f.second) {
            bool ok = true;
            for (uint64_t s : chain_snarls[f.first]) ok = ok && snarl_acyclic[s];
            chain_acyclic[f.first] = ok;
            auto& d = chain_dist[f.first];
            const auto& links = chain_snarls[f.first];
            for (size_t i = 0; i < links.size(); ++i) {
                const auto& sd = snarl_dist[links[i]];
                d.first += sd.first;
                if (d.second == kNone || sd.second == kNone) d.second = kNone;
                else d.second += sd.second;
                if (i != 0) { d.first -= 1; if (d.second != kNone) d.second -= 1; }
            }
        } else {
            const Net net = net_graph(g, f.first);
            net_acyclic[f.first] = net.acyclic();
            bool ok = net_acyclic[f.first];
            if (ok) for (uint64_t c : snarl_children[f.first]) ok = ok && chain_acyclic[c];
            snarl_acyclic[f.first] = ok;
            bool finite = true;
            for (uint64_t c : snarl_children[f.first]) finite = finite && chain_dist[c].second != kNone;
            finite = finite && net_acyclic[f.first];
            auto& sd = snarl_dist[f.first];
            const uint64_t nn = net.label.size();
            if (finite) {
                std::vector<std::pair<int64_t, int64_t>> dp(nn, std::make_pair(std::numeric_limits<int64_t>::max(), (int64_t)-1));
                std::vector<uint64_t> order, stack, deg(net.indeg);
                for (uint64_t v = 0; v < nn; ++v) if (deg[v] == 0) stack.push_back(v);
                while (!stack.empty()) {
                    const uint64_t v = stack.back();
                    stack.pop_back();
                    order.push_back(v);
                    for (uint64_t w : net.next[v]) if (--deg[w] == 0) stack.push_back(w);
                }
                dp[order.front()] = std::make_pair((int64_t)1, (int64_t)1);
                for (uint64_t v : order)
                    for (uint64_t w : net.next[v]) {
                        int64_t lo, hi;
                        if (net.label[w].second) { lo = dp[v].first + (int64_t)chain_dist[net.label[w].first].first; hi = dp[v].second + (int64_t)chain_dist[net.label[w].first].second; }
                        else { lo = dp[v].first + 1; hi = dp[v].second + 1; }
                        if (lo < dp[w].first) dp[w].first = lo;
                        if (hi > dp[w].second) dp[w].second = hi;
                    }
                sd.first = (uint64_t)dp[order.back()].first;
                sd.second = (uint64_t)dp[order.back()].second;
            } else {
                std::vector<char> popped(nn, 0);
                std::priority_queue<std::pair<uint64_t, uint64_t>, std::vector<std::pair<uint64_t, uint64_t>>, std::greater<std::pair<uint64_t, uint64_t>>> queue;
                for (uint64_t v = 0; v < nn; ++v) if (net.indeg[v] == 0) { queue.emplace(1, v); break; }
                std::vector<uint64_t> dist(nn, 0);
                while (!queue.empty()) {
                    const auto top = queue.top();
                    queue.pop();
                    if (popped[top.second]) continue;
                    popped[top.second] = 1;
                    dist[top.second] = top.first;
                    for (uint64_t w : net.next[top.second]) queue.emplace(top.first + (net.label[w].second ? chain_dist[net.label[w].first].first : 1), w);
                }
                for (uint64_t v = 0; v < nn; ++v) if (net.next[v].empty()) { sd.first = dist[v]; break; }
                sd.second = kNone;
            }
        }
    }
    return true;
}

}  // namespace clsnarl

#endif
