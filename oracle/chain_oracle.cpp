/*
 * chain_oracle.cpp — TEST INFRASTRUCTURE ONLY.  CPU restatement of the Anchorer's chaining DPs
 * (SURVEY.md §8 rows a15-a22), single thread, written from the behaviour of the reference:
 *
 *   sparse_affine_chain_dp   include/centrolign/anchorer.hpp:1812-2471   (3-piece affine gap cost on the diagonal shift)
 *   sparse_chain_dp          include/centrolign/anchorer.hpp:1511-1750   (no gap cost)
 *   traceback_sparse_dp      include/centrolign/anchorer.hpp:2473-2547
 *   MatchBank                include/centrolign/match_bank.hpp:16-290    (iteration order, strict '>' update)
 *   MaxSearchTree            include/centrolign/max_search_tree.hpp      (implicit-heap BST + subtree-max pointers;
 *                                                                         winner among equal maxima = first met in
 *                                                                         its range_max traversal, :361-444)
 *   OrthogonalMaxSearchTree  include/centrolign/orthogonal_max_search_tree.hpp (outer BST on (key1,key2), cross trees on key2
 *                                                                         holding (value, outer index) pairs, :343-544)
 *   ForwardEdges / masks     include/centrolign/forward_edges.hpp:34-69, anchorer.hpp:1752-1810
 *   PostSwitchDistances      include/centrolign/post_switch_distances.hpp:44-81
 *   PathMerge                include/centrolign/path_merge.hpp:96-277
 *   ScoreFunction            include/centrolign/score_function.hpp:47-75
 *
 * Arithmetic follows the source's types: DP values are float, the gap terms are evaluated in double and rounded
 * to float on assignment, shifts are int32 computed modulo 2^32.  Parity status: PINNED — compared against the
 * compiled reference (oracle/_ref: ref_chain_dp in ref_driver.cpp) in tests/test_chain_oracle.py and against
 * the golden chains under tests/golden/chain_*.npz.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include "cl_oracle.h"

namespace {

constexpr uint32_t kNone = std::numeric_limits<uint32_t>::max();
constexpr float kMinInf = std::numeric_limits<float>::lowest();

struct Graph {
    const cl_base_graph* g;
    uint64_t n() const { return g->n_nodes; }
};

std::vector<uint32_t> topological_order(const cl_base_graph& g) {
    std::vector<uint32_t> order, stack, indeg(g.n_nodes);
    order.reserve(g.n_nodes);
    for (uint64_t v = 0; v < g.n_nodes; ++v) {
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
    return order;
}

// path_merge.hpp:96-277
struct PathMerge {
    const cl_base_graph* g = nullptr;
    uint64_t n = 0, chains = 0;
    std::vector<uint32_t> head, index, link, table;
    explicit PathMerge(const cl_base_graph& gr) : g(&gr), n(gr.n_nodes), chains(gr.n_paths + 1) {
        head.assign(n, kNone);
        index.assign(chains * n, kNone);
        link.assign(chains * n, kNone);
        table.assign(n * chains, kNone);
        for (uint64_t p = 0; p < gr.n_paths; ++p) {
            uint32_t idx = 0;
            for (uint64_t i = gr.path_off[p]; i < gr.path_off[p + 1]; ++i, ++idx) {
                uint32_t v = gr.path_nodes[i];
                for (uint64_t e = gr.next_off[v]; e < gr.next_off[v + 1]; ++e) table[(uint64_t)gr.next_idx[e] * chains + p] = idx;
                index[p * n + v] = idx;
                link[p * n + v] = head[v];
                head[v] = (uint32_t)p;
            }
        }
        for (uint32_t v : topological_order(gr)) {
            for (uint64_t e = gr.prev_off[v]; e < gr.prev_off[v + 1]; ++e) {
                uint32_t u = gr.prev_idx[e];
                for (uint64_t p = 0; p < gr.n_paths; ++p) {
                    uint32_t& mine = table[(uint64_t)v * chains + p];
                    uint32_t theirs = table[(uint64_t)u * chains + p];
                    if (mine == kNone) mine = theirs;
                    else if (theirs != kNone) mine = std::max(mine, theirs);
                }
            }
        }
        uint64_t pp = gr.n_paths;
        index[pp * n + gr.src_id] = 0;
        index[pp * n + gr.snk_id] = 1;
        head[gr.src_id] = (uint32_t)pp;
        head[gr.snk_id] = (uint32_t)pp;
        for (uint64_t v = 0; v < n; ++v)
            if (v != gr.src_id) table[v * chains + pp] = 0;
    }
    uint32_t pred(uint64_t v, uint64_t c) const { return table[v * chains + c]; }
    uint32_t index_on(uint64_t v, uint64_t c) const { return index[c * n + v]; }
    std::vector<uint32_t> chains_on(uint64_t v) const {
        std::vector<uint32_t> out;
        for (uint32_t p = head[v]; p != kNone; p = link[(uint64_t)p * n + v]) out.push_back(p);
        return out;
    }
    std::pair<uint32_t, uint32_t> chain(uint64_t v) const {  // path_merge.hpp:221-227
        uint32_t p = head[v];
        if (p == kNone) return {kNone, kNone};
        return {p, index[(uint64_t)p * n + v]};
    }
    uint64_t node_at(uint64_t c, uint64_t i) const {
        if (c == g->n_paths) return i ? g->snk_id : g->src_id;
        return g->path_nodes[g->path_off[c] + i];
    }
};

// post_switch_distances.hpp:44-81; stored value 0 = none, otherwise distance + 1; distance() returns -1 for none
struct PostSwitch {
    uint64_t n;
    std::vector<uint32_t> d;  // [chain][node]
    PostSwitch(const cl_base_graph& g, const PathMerge& pm) : n(g.n_nodes), d(pm.chains * g.n_nodes, 0) {
        for (uint32_t v : topological_order(g)) {
            for (uint64_t p = 0; p < pm.chains; ++p) {
                uint32_t* row = &d[p * n];
                for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) {
                    uint32_t u = g.prev_idx[e];
                    uint32_t pr = pm.pred(v, p);
                    if (pm.index_on(u, p) == pr) {
                        row[v] = 1;
                        break;
                    } else if (pm.pred(u, p) == pr) {
                        uint64_t thru = (uint64_t)row[u] + 1;
                        if (row[v] == 0 || row[v] > thru) row[v] = (uint32_t)thru;
                    }
                }
            }
        }
    }
    // as the reference's size_t arithmetic modulo 2^32: "none" behaves as -1
    uint32_t distance(uint64_t v, uint64_t p) const {
        uint32_t x = d[p * n + v];
        return x == 0 ? 0xFFFFFFFFu : x;
    }
};

struct MatchId {
    uint32_t set;
    uint16_t i1, i2;
    bool operator==(const MatchId& o) const { return set == o.set && i1 == o.i1 && i2 == o.i2; }
    bool operator!=(const MatchId& o) const { return !(*this == o); }
    bool operator<(const MatchId& o) const { return std::tie(set, i1, i2) < std::tie(o.set, o.i1, o.i2); }
    bool operator>(const MatchId& o) const { return o < *this; }
};
const MatchId kMaxId{0xFFFFFFFFu, 0xFFFF, 0xFFFF};  // MatchBank::max(), match_bank.hpp:226-228
const MatchId kMinId{0, 0, 0};

struct Matches {
    const clo_match_sets* m;
    uint64_t n_sets;  // num_match_sets (the leading sets that take part)
    uint64_t n1(uint64_t s) const { return m->set_off1[s + 1] - m->set_off1[s]; }
    uint64_t n2(uint64_t s) const { return m->set_off2[s + 1] - m->set_off2[s]; }
    uint32_t front1(const MatchId& id) const { return m->nodes1[m->walk_off1[m->set_off1[id.set] + id.i1]]; }
    uint32_t back1(const MatchId& id) const { return m->nodes1[m->walk_off1[m->set_off1[id.set] + id.i1 + 1] - 1]; }
    uint32_t front2(const MatchId& id) const { return m->nodes2[m->walk_off2[m->set_off2[id.set] + id.i2]]; }
    uint32_t back2(const MatchId& id) const { return m->nodes2[m->walk_off2[m->set_off2[id.set] + id.i2 + 1] - 1]; }
    uint64_t walk_len(uint64_t s) const { return m->walk_off1[m->set_off1[s] + 1] - m->walk_off1[m->set_off1[s]]; }  // walks1.front().size()
};

// score_function.hpp:51-75
double anchor_weight(const clo_chain_params& cp, uint64_t count1, uint64_t count2, uint64_t length, uint64_t full_length) {
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

// match_bank.hpp: dp value + backpointer per (set, i1, i2); starts_on / ends_on in construction order
struct MatchBank {
    const Matches& M;
    std::vector<uint64_t> base;  // per set, offset into dp
    std::vector<float> val;
    std::vector<MatchId> back;
    std::vector<std::vector<std::pair<uint32_t, uint16_t>>> starts, ends;
    MatchBank(const Matches& m, uint64_t n_nodes1) : M(m), starts(n_nodes1), ends(n_nodes1) {
        base.resize(M.n_sets + 1, 0);
        for (uint64_t s = 0; s < M.n_sets; ++s) base[s + 1] = base[s] + M.n1(s) * M.n2(s);
        val.assign(base[M.n_sets], kMinInf);
        back.assign(base[M.n_sets], kMaxId);
        for (uint32_t s = 0; s < M.n_sets; ++s)
            for (uint64_t j = 0; j < M.n1(s); ++j) {
                MatchId id{s, (uint16_t)j, 0};
                starts[M.front1(id)].emplace_back(s, (uint16_t)j);
                ends[M.back1(id)].emplace_back(s, (uint16_t)j);
            }
    }
    uint64_t slot(const MatchId& id) const { return base[id.set] + (uint64_t)id.i1 * M.n2(id.set) + id.i2; }
    float dp(const MatchId& id) const { return val[slot(id)]; }
    void update(const MatchId& id, float v, const MatchId& from) {  // match_bank.hpp:171-183: strict '>'
        uint64_t s = slot(id);
        if (v > val[s]) { val[s] = v; back[s] = from; }
    }
    template <class F>
    void for_each(F f) const {  // iterator order: set, walk1 index, walk2 index
        for (uint32_t s = 0; s < M.n_sets; ++s)
            for (uint64_t j = 0; j < M.n1(s); ++j)
                for (uint64_t k = 0; k < M.n2(s); ++k) f(MatchId{s, (uint16_t)j, (uint16_t)k});
    }
    template <class F>
    void for_each_at(const std::vector<std::pair<uint32_t, uint16_t>>& lst, F f) const {
        for (const auto& e : lst)
            for (uint64_t k = 0; k < M.n2(e.first); ++k) f(MatchId{e.first, e.second, (uint16_t)k});
    }
};

// ---- MaxSearchTree: static implicit-heap BST over sorted keys, each node points at the max of its subtree ----------
template <class K, class V>
struct MaxTree {
    std::vector<K> key;
    std::vector<V> val;
    std::vector<uint32_t> smax;
    static size_t L(size_t x) { return 2 * x + 1; }
    static size_t R(size_t x) { return 2 * x + 2; }
    static size_t P(size_t x) { return (x - 1) / 2; }
    size_t size() const { return key.size(); }
    size_t end() const { return key.size(); }

    MaxTree() = default;
    // data is stable-sorted by key unless already sorted (max_search_tree.hpp:104-111), then laid out in-order
    explicit MaxTree(std::vector<std::pair<K, V>>& data) : key(data.size()), val(data.size()), smax(data.size()) {
        if (data.empty()) return;
        auto cmp = [](const std::pair<K, V>& a, const std::pair<K, V>& b) { return a.first < b.first; };
        if (!std::is_sorted(data.begin(), data.end(), cmp)) std::stable_sort(data.begin(), data.end(), cmp);
        size_t next = 0;
        fill_in_order(0, data, next);
        for (size_t i = 0; i < size(); ++i) smax[i] = (uint32_t)i;
        for (size_t i = size() - 1; i > 0; --i)
            if (val[smax[i]] > val[smax[P(i)]]) smax[P(i)] = smax[i];
    }
    void fill_in_order(size_t root, const std::vector<std::pair<K, V>>& data, size_t& next) {
        // iterative in-order walk of the implicit heap
        std::vector<std::pair<size_t, bool>> st{{root, false}};
        while (!st.empty()) {
            auto& top = st.back();
            if (!top.second) {
                top.second = true;
                if (L(top.first) < size()) st.emplace_back(L(top.first), false);
            } else {
                size_t x = top.first;
                key[x] = data[next].first;
                val[x] = data[next].second;
                ++next;
                st.pop_back();
                if (R(x) < size()) st.emplace_back(R(x), false);
            }
        }
    }
    size_t find(const K& k) const {
        size_t c = 0;
        while (c < size()) {
            if (key[c] == k) return c;
            c = key[c] > k ? L(c) : R(c);
        }
        return end();
    }
    // lowest-position node with this key, as the first iterator of equal_range (max_search_tree.hpp:245-262)
    size_t lower_equal(const K& k) const {
        size_t lower = end(), c = 0;
        while (c < size()) {
            if (key[c] == k) { lower = c; c = L(c); }
            else c = key[c] > k ? L(c) : R(c);
        }
        return lower;
    }
    size_t successor(size_t i) const {  // iterator::operator++, max_search_tree.hpp:467-499
        if (R(i) < size()) {
            i = R(i);
            while (L(i) < size()) i = L(i);
            return i;
        }
        if (i == 0) return end();
        while (true) {
            size_t p = P(i);
            if (i == L(p)) return p;
            if (p == 0) return end();
            i = p;
        }
    }
    void refresh(size_t x) {
        size_t best = x;
        if (L(x) < size() && val[smax[L(x)]] > val[best]) best = smax[L(x)];
        if (R(x) < size() && val[smax[R(x)]] > val[best]) best = smax[R(x)];
        smax[x] = (uint32_t)best;
    }
    void update(size_t i, const V& v) {  // max_search_tree.hpp:318-358
        if (v > val[smax[i]]) {
            smax[i] = (uint32_t)i;
            size_t here = i;
            while (here != 0) {
                here = P(here);
                if (v > val[smax[here]]) smax[here] = (uint32_t)i;
                else break;
            }
            val[i] = v;
        } else {
            val[i] = v;
            if (smax[i] == i) {
                refresh(i);
                size_t here = i;
                while (here != 0) {
                    here = P(here);
                    if (smax[here] != i) break;
                    refresh(here);
                }
            }
        }
    }
    // max over keys in [lo, hi); among equal maxima the first met in this traversal (max_search_tree.hpp:361-444)
    size_t range_max(const K& lo, const K& hi) const {
        size_t c = 0;
        while (c < size() && (key[c] < lo || !(key[c] < hi))) c = !(key[c] < lo) ? L(c) : R(c);
        if (c >= size()) return end();
        size_t best = c, lc = L(c), rc = R(c);
        while (lc < size()) {
            if (!(key[lc] < lo)) {
                if (val[lc] > val[best]) best = lc;
                size_t r = R(lc);
                if (r < size() && val[smax[r]] > val[best]) best = smax[r];
                lc = L(lc);
            } else lc = R(lc);
        }
        while (rc < size()) {
            if (key[rc] < hi) {
                if (val[rc] > val[best]) best = rc;
                size_t l = L(rc);
                if (l < size() && val[smax[l]] > val[best]) best = smax[l];
                rc = R(rc);
            } else rc = L(rc);
        }
        return best;
    }
};

// ---- OrthogonalMaxSearchTree -----------------------------------------------------------------------------------------
template <class K1, class K2, class V>
struct OrthoTree {
    using Cross = MaxTree<K2, std::pair<V, uint32_t>>;
    std::vector<K1> key1;
    std::vector<K2> key2;
    std::vector<V> val;
    std::vector<Cross> cross;
    static size_t L(size_t x) { return 2 * x + 1; }
    static size_t R(size_t x) { return 2 * x + 2; }
    size_t size() const { return key1.size(); }
    size_t end() const { return key1.size(); }

    OrthoTree() = default;
    explicit OrthoTree(std::vector<std::tuple<K1, K2, V>> data) : key1(data.size()), key2(data.size()), val(data.size()), cross(data.size()) {
        if (data.empty()) return;
        auto cmp = [](const std::tuple<K1, K2, V>& a, const std::tuple<K1, K2, V>& b) {
            return std::get<0>(a) < std::get<0>(b) || (std::get<0>(a) == std::get<0>(b) && std::get<1>(a) < std::get<1>(b));
        };
        if (!std::is_sorted(data.begin(), data.end(), cmp)) std::stable_sort(data.begin(), data.end(), cmp);
        // in-order layout; pos[i] = heap node of the i-th smallest record
        std::vector<uint32_t> pos(data.size());
        {
            size_t next = 0;
            std::vector<std::pair<size_t, bool>> st{{0, false}};
            while (!st.empty()) {
                auto& top = st.back();
                if (!top.second) {
                    top.second = true;
                    if (L(top.first) < size()) st.emplace_back(L(top.first), false);
                } else {
                    size_t x = top.first;
                    pos[next] = (uint32_t)x;
                    key1[x] = std::get<0>(data[next]);
                    key2[x] = std::get<1>(data[next]);
                    val[x] = std::get<2>(data[next]);
                    ++next;
                    st.pop_back();
                    if (R(x) < size()) st.emplace_back(R(x), false);
                }
            }
        }
        // the two outer spines are never queried: no cross trees there (orthogonal_max_search_tree.hpp:170-177)
        std::vector<char> make(size(), 1);
        for (size_t c = 0; c < size(); c = L(c)) make[c] = 0;
        for (size_t c = R(0); c < size(); c = R(c)) make[c] = 0;
        // every outer node's cross tree holds the records of its whole subtree, in the order they reach it:
        // the subtree of heap node x is a contiguous rank interval; records keep their sorted order
        struct Job { size_t node, lo, hi; };
        std::vector<Job> jobs{{0, 0, data.size()}};
        std::vector<uint32_t> rank_of(size());
        for (size_t i = 0; i < data.size(); ++i) rank_of[pos[i]] = (uint32_t)i;
        while (!jobs.empty()) {
            Job j = jobs.back();
            jobs.pop_back();
            if (make[j.node]) {
                std::vector<std::pair<K2, std::pair<V, uint32_t>>> recs;
                recs.reserve(j.hi - j.lo);
                for (size_t i = j.lo; i < j.hi; ++i) recs.emplace_back(std::get<1>(data[i]), std::make_pair(std::get<2>(data[i]), pos[i]));
                cross[j.node] = Cross(recs);
            }
            size_t mid = rank_of[j.node];
            if (L(j.node) < size()) jobs.push_back({L(j.node), j.lo, mid});
            if (R(j.node) < size()) jobs.push_back({R(j.node), mid + 1, j.hi});
        }
    }
    size_t find(const K1& k1, const K2& k2) const {
        size_t c = 0;
        while (c < size()) {
            if (key1[c] == k1 && key2[c] == k2) return c;
            c = (std::make_pair(key1[c], key2[c]) > std::make_pair(k1, k2)) ? L(c) : R(c);
        }
        return end();
    }
    void update(size_t i, const V& v) {  // orthogonal_max_search_tree.hpp:318-340
        val[i] = v;
        for (size_t c = i; c < size(); c = (c == 0 ? size() : (c - 1) / 2)) {
            Cross& ct = cross[c];
            if (ct.size() == 0) break;
            size_t it = ct.lower_equal(key2[i]);
            while (ct.val[it].second != i) it = ct.successor(it);
            ct.update(it, std::make_pair(v, (uint32_t)i));
        }
    }
    // max over [lo1,hi1) x [lo2,hi2) (orthogonal_max_search_tree.hpp:343-544); returns outer node or end()
    size_t range_max(const K1& lo1, const K1& hi1, const K2& lo2, const K2& hi2) const {
        size_t c = 0;
        while (c < size() && (key1[c] < lo1 || !(key1[c] < hi1))) c = !(key1[c] < hi1) ? L(c) : R(c);
        if (c >= size()) return end();
        bool have = false;
        V bestv{};
        size_t best = end();
        auto in2 = [&](size_t x) { return !(key2[x] < lo2) && key2[x] < hi2; };
        auto consider = [&](const V& v, size_t node) {
            if (!have || v > bestv) { have = true; bestv = v; best = node; }
        };
        if (in2(c)) consider(val[c], c);
        size_t lc = L(c), rc = R(c);
        while (lc < size()) {
            if (!(key1[lc] < lo1)) {
                if (in2(lc)) consider(val[lc], lc);
                size_t r = R(lc);
                if (r < size()) {
                    size_t it = cross[r].range_max(lo2, hi2);
                    if (it != cross[r].end()) consider(cross[r].val[it].first, cross[r].val[it].second);
                }
                lc = L(lc);
            } else lc = R(lc);
        }
        while (rc < size()) {
            if (key1[rc] < hi1) {
                if (in2(rc)) consider(val[rc], rc);
                size_t l = L(rc);
                if (l < size()) {
                    size_t it = cross[l].range_max(lo2, hi2);
                    if (it != cross[l].end()) consider(cross[l].val[it].first, cross[l].val[it].second);
                }
                rc = R(rc);
            } else rc = L(rc);
        }
        return have ? best : end();
    }
};

// forward edges restricted by the match start / "after a match end" masks (forward_edges.hpp:40-53, anchorer.hpp:1752-1810)
struct ForwardEdges {
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> edges;
    ForwardEdges(const cl_base_graph& g, const PathMerge& pm, const Matches& M) : edges(g.n_nodes) {
        std::vector<char> has_start(g.n_nodes, 0), after_end(g.n_nodes, 0);
        for (uint32_t s = 0; s < M.n_sets; ++s)
            for (uint64_t j = 0; j < M.n1(s); ++j) {
                MatchId id{s, (uint16_t)j, 0};
                has_start[M.front1(id)] = 1;
                after_end[M.back1(id)] = 1;
            }
        std::vector<uint32_t> st;
        for (uint64_t v = 0; v < g.n_nodes; ++v)
            if (after_end[v]) {
                st.push_back((uint32_t)v);
                while (!st.empty()) {
                    uint32_t h = st.back();
                    st.pop_back();
                    for (uint64_t e = g.next_off[h]; e < g.next_off[h + 1]; ++e)
                        if (!after_end[g.next_idx[e]]) { after_end[g.next_idx[e]] = 1; st.push_back(g.next_idx[e]); }
                }
            }
        for (uint64_t v = 0; v < g.n_nodes; ++v) {
            if (!has_start[v]) continue;
            for (uint64_t p = 0; p < pm.chains; ++p) {
                uint32_t idx = pm.pred(v, p);
                if (idx == kNone) continue;
                uint64_t from = pm.node_at(p, idx);
                if (after_end[from]) edges[from].emplace_back((uint32_t)v, (uint32_t)p);
            }
        }
    }
};

// sources / sinks of global anchoring (anchorer.hpp:1071-1076: next(src sentinel), previous(snk sentinel))
struct Ends {
    std::vector<uint32_t> src1, src2, snk1, snk2;
    bool on = false;
    Ends(const cl_base_graph& g1, const cl_base_graph& g2, bool global) : on(global) {
        if (!global) return;
        for (uint64_t e = g1.next_off[g1.src_id]; e < g1.next_off[g1.src_id + 1]; ++e) src1.push_back(g1.next_idx[e]);
        for (uint64_t e = g2.next_off[g2.src_id]; e < g2.next_off[g2.src_id + 1]; ++e) src2.push_back(g2.next_idx[e]);
        for (uint64_t e = g1.prev_off[g1.snk_id]; e < g1.prev_off[g1.snk_id + 1]; ++e) snk1.push_back(g1.prev_idx[e]);
        for (uint64_t e = g2.prev_off[g2.snk_id]; e < g2.prev_off[g2.snk_id + 1]; ++e) snk2.push_back(g2.prev_idx[e]);
    }
};

bool reachable(const PathMerge& x, uint64_t from, uint64_t to) {  // path_merge.hpp:235-248
    auto ch = x.chain(from);
    if (ch.first == kNone) return false;
    uint32_t last = x.pred(to, ch.first);
    return last != kNone && ch.second <= last;
}

using ShiftKey = std::pair<int32_t, MatchId>;
using OffKey = std::pair<uint32_t, MatchId>;

// traceback_sparse_dp (anchorer.hpp:2473-2547): first strictly better (dp value + final term) in iteration order
template <class FinalF>
std::vector<MatchId> traceback(const MatchBank& bank, float min_score, FinalF final_term) {
    float opt = kMinInf;
    MatchId best = kMaxId;
    bank.for_each([&](const MatchId& id) {
        float v = bank.dp(id);
        const float f = final_term(id);
        if (f == kMinInf) v = f;
        else v += f;
        if (v > opt && v > min_score) { opt = v; best = id; }
    });
    std::vector<MatchId> chain;
    for (MatchId here = best; here != kMaxId; here = bank.back[bank.slot(here)]) chain.push_back(here);
    std::reverse(chain.begin(), chain.end());
    return chain;
}

}  // namespace

extern "C" {

// sparse_affine_chain_dp without sources/sinks/masks (local anchoring, the CLI default), NumPW = 3
int clo_sparse_affine_chain_ex(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                               const clo_chain_params* cp, double local_scale, int global_anchoring, uint32_t* chain_out,
                               uint64_t* chain_len, float* dp_out, clo_chain_ends* ends_out) {
    constexpr int NPW = 3;
    const Ends E(*g1, *g2, global_anchoring != 0);
    Matches M{ms, num_match_sets};
    PathMerge x1(*g1), x2(*g2);
    MatchBank bank(M, g1->n_nodes);
    PostSwitch sw1(*g1, x1), sw2(*g2, x2);
    const uint64_t C1 = x1.chains, C2 = x2.chains;

    auto source_shift = [&](const MatchId& id, uint64_t p1, uint64_t p2) -> int32_t {
        return (int32_t)(x1.index_on(M.back1(id), p1) - x2.index_on(M.back2(id), p2));
    };
    auto query_shift = [&](const MatchId& id, uint64_t p1, uint64_t p2) -> int32_t {
        uint32_t q1 = M.front1(id), q2 = M.front2(id);
        return (int32_t)(x1.pred(q1, p1) - x2.pred(q2, p2) + sw1.distance(q1, p1) - sw2.distance(q2, p2));
    };
    auto key_offset = [&](const MatchId& id, uint64_t p2) -> uint32_t { return x2.index_on(M.back2(id), p2); };
    auto query_offset = [&](const MatchId& id, uint64_t p2) -> uint32_t { return x2.pred(M.front2(id), p2) + 1u; };
    auto weight_of = [&](const MatchId& id) -> float {
        return (float)anchor_weight(*cp, ms->count1[id.set], ms->count2[id.set], M.walk_len(id.set), ms->full_length[id.set]);
    };

    // gap measurement between node pairs and sets (anchorer.hpp:1906-2000)
    auto basic_source_shift = [&](uint32_t a, uint32_t b, uint64_t p1, uint64_t p2) -> int32_t { return (int32_t)(x1.index_on(a, p1) - x2.index_on(b, p2)); };
    auto basic_query_shift = [&](uint32_t a, uint32_t b, uint64_t p1, uint64_t p2) -> int32_t {
        return (int32_t)(x1.pred(a, p1) - x2.pred(b, p2) + sw1.distance(a, p1) - sw2.distance(b, p2));
    };
    auto score_gap = [&](int32_t gap) -> float {
        float score = kMinInf;
        if (gap == 0) score = 0.0f;
        else if (gap != std::numeric_limits<int32_t>::max())
            for (int pw = 0; pw < NPW; ++pw) score = std::max<float>(score, (float)(-local_scale * (cp->gap_open[pw] + cp->gap_extend[pw] * std::abs(gap))));
        return score;
    };
    auto measure_gap = [&](uint32_t p1n, uint32_t p2n, uint32_t c1n, uint32_t c2n) -> int32_t {
        int32_t gap = std::numeric_limits<int32_t>::max();
        if ((p1n == c1n || reachable(x1, p1n, c1n)) && (p2n == c2n || reachable(x2, p2n, c2n)))
            for (uint32_t p1 : x1.chains_on(p1n))
                for (uint32_t p2 : x2.chains_on(p2n)) {
                    int32_t here = (int32_t)((uint32_t)basic_source_shift(p1n, p2n, p1, p2) - (uint32_t)basic_query_shift(c1n, c2n, p1, p2));
                    if (std::abs(here) < std::abs(gap)) gap = here;
                }
        return gap;
    };
    // note the asymmetric comparison of the reference: |gap_here| against the signed current value (anchorer.hpp:1954,1971,1991)
    auto measure_gap_sn = [&](const std::vector<uint32_t>& pv1, const std::vector<uint32_t>& pv2, uint32_t c1n, uint32_t c2n) {
        std::pair<int32_t, float> r(std::numeric_limits<int32_t>::max(), kMinInf);
        for (uint32_t a : pv1) for (uint32_t b : pv2) { int32_t h = measure_gap(a, b, c1n, c2n); if (std::abs(h) < r.first) r.first = h; }
        r.second = score_gap(r.first);
        return r;
    };
    auto measure_gap_ns = [&](uint32_t p1n, uint32_t p2n, const std::vector<uint32_t>& cv1, const std::vector<uint32_t>& cv2) {
        std::pair<int32_t, float> r(std::numeric_limits<int32_t>::max(), kMinInf);
        for (uint32_t a : cv1) for (uint32_t b : cv2) { int32_t h = measure_gap(p1n, p2n, a, b); if (std::abs(h) < r.first) r.first = h; }
        r.second = score_gap(r.first);
        return r;
    };
    auto measure_gap_ss = [&]() {
        std::pair<int32_t, float> r(std::numeric_limits<int32_t>::max(), kMinInf);
        for (uint32_t c1n : E.snk1) for (uint32_t c2n : E.snk2) for (uint32_t a : E.src1) for (uint32_t b : E.src2) {
            int32_t h = measure_gap(a, b, c1n, c2n);
            if (std::abs(h) < r.first) r.first = h;
        }
        r.second = score_gap(r.first);
        return r;
    };

    // bookkeeping (anchorer.hpp:2002-2049)
    std::vector<std::vector<std::tuple<ShiftKey, uint32_t, float>>> ortho_data(C1 * C2);
    bank.for_each([&](const MatchId& id) {
        float init_w = weight_of(id);
        if (E.on) {
            float lead = measure_gap_sn(E.src1, E.src2, M.front1(id), M.front2(id)).second;
            if (lead == kMinInf) init_w = kMinInf;
            else init_w += lead;
        }
        bank.update(id, init_w, kMaxId);
        for (uint32_t p1 : x1.chains_on(M.back1(id)))
            for (uint32_t p2 : x2.chains_on(M.back2(id)))
                ortho_data[p1 * C2 + p2].emplace_back(ShiftKey(source_shift(id, p1, p2), id), key_offset(id, p2), kMinInf);
    });
    using Ortho = OrthoTree<ShiftKey, uint32_t, float>;
    std::vector<Ortho> ortho(2 * NPW * C1 * C2);  // [pw][p1][p2]
    for (uint64_t p1 = 0; p1 < C1; ++p1)
        for (uint64_t p2 = 0; p2 < C2; ++p2) {
            for (int pw = 0; pw < 2 * NPW; ++pw) ortho[(pw * C1 + p1) * C2 + p2] = Ortho(ortho_data[p1 * C2 + p2]);
            std::vector<std::tuple<ShiftKey, uint32_t, float>>().swap(ortho_data[p1 * C2 + p2]);
        }
    // gap-free trees per (p1, p2, diagonal) (anchorer.hpp:2136-2237); the per-diagonal lists are built by
    // emplace_front, i.e. they reach the tree constructor in REVERSE iteration order before its stable sort
    using GapFree = MaxTree<OffKey, float>;
    std::vector<int32_t> min_shift(C1 * C2, 0);
    std::vector<std::vector<std::vector<std::pair<OffKey, float>>>> gf_data(C1 * C2);
    {
        std::vector<std::vector<std::tuple<int32_t, OffKey>>> flat(C1 * C2);
        bank.for_each([&](const MatchId& id) {
            for (uint32_t p1 : x1.chains_on(M.back1(id)))
                for (uint32_t p2 : x2.chains_on(M.back2(id)))
                    flat[p1 * C2 + p2].emplace_back(source_shift(id, p1, p2), OffKey(key_offset(id, p2), id));
        });
        for (uint64_t c = 0; c < C1 * C2; ++c) {
            if (flat[c].empty()) continue;
            int32_t lo = std::get<0>(flat[c][0]), hi = lo;
            for (auto& t : flat[c]) { lo = std::min(lo, std::get<0>(t)); hi = std::max(hi, std::get<0>(t)); }
            min_shift[c] = lo;
            gf_data[c].resize((size_t)((int64_t)hi - lo + 1));
            for (auto it = flat[c].rbegin(); it != flat[c].rend(); ++it)
                gf_data[c][(size_t)((int64_t)std::get<0>(*it) - lo)].emplace_back(std::get<1>(*it), kMinInf);
        }
    }
    std::vector<std::vector<std::unique_ptr<GapFree>>> gf(C1 * C2);
    for (uint64_t c = 0; c < C1 * C2; ++c) {
        gf[c].resize(gf_data[c].size());
        for (size_t i = 0; i < gf_data[c].size(); ++i)
            if (!gf_data[c][i].empty()) gf[c][i].reset(new GapFree(gf_data[c][i]));
        std::vector<std::vector<std::pair<OffKey, float>>>().swap(gf_data[c]);
    }
    ForwardEdges fwd(*g1, x1, M);

    // main sweep (anchorer.hpp:2290-2417)
    for (uint32_t node : topological_order(*g1)) {
        bank.for_each_at(bank.ends[node], [&](const MatchId& id) {
            const float dpv = bank.dp(id);
            for (uint32_t p1 : x1.chains_on(M.back1(id)))
                for (uint32_t p2 : x2.chains_on(M.back2(id))) {
                    const int32_t shift = source_shift(id, p1, p2);
                    const uint32_t koff = key_offset(id, p2);
                    const uint64_t c = p1 * C2 + p2;
                    {
                        GapFree& t = *gf[c][(size_t)((int64_t)shift - min_shift[c])];
                        t.update(t.find(OffKey(koff, id)), dpv);
                    }
                    for (int pw = 0; pw < 2 * NPW; ++pw) {
                        float value;
                        if (pw % 2 == 1) value = (float)(dpv + local_scale * cp->gap_extend[pw / 2] * shift);
                        else value = (float)(dpv - local_scale * cp->gap_extend[pw / 2] * shift);
                        Ortho& t = ortho[(pw * C1 + p1) * C2 + p2];
                        size_t it = t.find(ShiftKey(shift, id), koff);
                        if (value > t.val[it]) t.update(it, value);
                    }
                }
        });
        for (const auto& edge : fwd.edges[node]) {
            const uint32_t fwd_id = edge.first, chain1 = edge.second;
            bank.for_each_at(bank.starts[fwd_id], [&](const MatchId& id) {
                const float weight = weight_of(id);
                for (uint64_t chain2 = 0; chain2 < C2; ++chain2) {
                    const int32_t query = query_shift(id, chain1, chain2);
                    const uint32_t offset = query_offset(id, chain2);
                    const uint64_t c = chain1 * C2 + chain2;
                    if (query >= min_shift[c] && (uint64_t)((int64_t)query - min_shift[c]) < gf[c].size()) {
                        const auto& tp = gf[c][(size_t)((int64_t)query - min_shift[c])];
                        if (tp) {
                            size_t it = tp->range_max(OffKey(0, kMinId), OffKey(offset, kMinId));
                            if (it != tp->end()) bank.update(id, tp->val[it] + weight, tp->key[it].second);
                        }
                    }
                    for (int pw = 0; pw < 2 * NPW; ++pw) {
                        const Ortho& t = ortho[(pw * C1 + chain1) * C2 + chain2];
                        if (pw % 2 == 1) {
                            size_t it = t.range_max(ShiftKey(std::numeric_limits<int32_t>::min(), kMinId), ShiftKey(query, kMinId), 0u, offset);
                            if (it != t.end()) {
                                float value = (float)((t.val[it] + weight) - local_scale * (cp->gap_open[pw / 2] + cp->gap_extend[pw / 2] * query));
                                bank.update(id, value, t.key1[it].second);
                            }
                        } else {
                            size_t it = t.range_max(ShiftKey(query + 1, kMinId), ShiftKey(std::numeric_limits<int32_t>::max(), kMaxId), 0u, offset);
                            if (it != t.end()) {
                                float value = (float)((t.val[it] + weight) - local_scale * (cp->gap_open[pw / 2] - cp->gap_extend[pw / 2] * query));
                                bank.update(id, value, t.key1[it].second);
                            }
                        }
                    }
                }
            });
        }
    }
    const float min_score = E.on ? measure_gap_ss().second : 0.0f;   // anchorer.hpp:2419-2424
    auto chain = traceback(bank, min_score, [&](const MatchId& id) -> float {
        return E.on ? measure_gap_ns(M.back1(id), M.back2(id), E.snk1, E.snk2).second : 0.0f;
    });
    *chain_len = chain.size();
    for (size_t i = 0; i < chain.size(); ++i) {
        chain_out[3 * i] = chain[i].set;
        chain_out[3 * i + 1] = chain[i].i1;
        chain_out[3 * i + 2] = chain[i].i2;
    }
    if (dp_out) bank.for_each([&](const MatchId& id) { dp_out[bank.slot(id)] = bank.dp(id); });
    if (ends_out) {   // gap annotation of the chain ends (anchorer.hpp:2445-2451, 2461-2467)
        memset(ends_out, 0, sizeof(*ends_out));
        if (E.on && !chain.empty()) {
            auto lead = measure_gap_sn(E.src1, E.src2, M.front1(chain.front()), M.front2(chain.front()));
            auto tail = measure_gap_ns(M.back1(chain.back()), M.back2(chain.back()), E.snk1, E.snk2);
            ends_out->gap_before_first = lead.first; ends_out->gap_score_before_first = lead.second;
            ends_out->gap_after_last = tail.first; ends_out->gap_score_after_last = tail.second;
        }
    }
    return 0;
}

int clo_sparse_affine_chain(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                            const clo_chain_params* cp, double local_scale, uint32_t* chain_out, uint64_t* chain_len,
                            float* dp_out) {
    return clo_sparse_affine_chain_ex(g1, g2, ms, num_match_sets, cp, local_scale, 0, chain_out, chain_len, dp_out, nullptr);
}

// sparse_chain_dp without sources/sinks/masks (anchorer.hpp:1511-1750)
int clo_sparse_chain_ex(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                        const clo_chain_params* cp, int global_anchoring, uint32_t* chain_out, uint64_t* chain_len, float* dp_out) {
    const Ends E(*g1, *g2, global_anchoring != 0);
    Matches M{ms, num_match_sets};
    PathMerge x1(*g1), x2(*g2);
    MatchBank bank(M, g1->n_nodes);
    const uint64_t C1 = x1.chains, C2 = x2.chains;
    auto weight_of = [&](const MatchId& id) -> float {
        return (float)anchor_weight(*cp, ms->count1[id.set], ms->count2[id.set], M.walk_len(id.set), ms->full_length[id.set]);
    };
    using Tree = MaxTree<OffKey, float>;
    std::vector<std::vector<std::pair<OffKey, float>>> data(C2);
    bank.for_each([&](const MatchId& id) {
        auto ch = x2.chain(M.back2(id));
        data[ch.first].emplace_back(OffKey(ch.second, id), kMinInf);
        float init_w = weight_of(id);
        if (E.on) {   // anchorer.hpp:1562-1582: a chain can only start on a match that the sources reach
            bool f1 = false, f2 = false;
            for (uint32_t a : E.src1) if (a == M.front1(id) || reachable(x1, a, M.front1(id))) { f1 = true; break; }
            for (uint32_t b : E.src2) if (b == M.front2(id) || reachable(x2, b, M.front2(id))) { f2 = true; break; }
            if (!f1 || !f2) init_w = kMinInf;
        }
        bank.update(id, init_w, kMaxId);
    });
    for (auto& d : data) std::stable_sort(d.begin(), d.end());
    std::vector<Tree> trees(C1 * C2);
    for (uint64_t i = 0; i < C1; ++i)
        for (uint64_t j = 0; j < C2; ++j) trees[i * C2 + j] = Tree(data[j]);
    ForwardEdges fwd(*g1, x1, M);
    for (uint32_t node : topological_order(*g1)) {
        const uint32_t chain1 = x1.chain(node).first;
        bank.for_each_at(bank.ends[node], [&](const MatchId& id) {
            auto ch = x2.chain(M.back2(id));
            Tree& t = trees[(uint64_t)chain1 * C2 + ch.first];
            size_t it = t.find(OffKey(ch.second, id));
            float dpv = bank.dp(id);
            if (t.val[it] < dpv) t.update(it, dpv);
        });
        for (const auto& edge : fwd.edges[node]) {
            const uint32_t fwd_id = edge.first, c1 = edge.second;
            bank.for_each_at(bank.starts[fwd_id], [&](const MatchId& id) {
                const float weight = weight_of(id);
                for (uint64_t c2 = 0; c2 < C2; ++c2) {
                    uint32_t pred2 = x2.pred(M.front2(id), c2);
                    if (pred2 == kNone) continue;
                    const Tree& t = trees[(uint64_t)c1 * C2 + c2];
                    size_t it = t.range_max(OffKey(0, kMinId), OffKey(pred2 + 1, kMinId));
                    if (it == t.end()) continue;
                    bank.update(id, t.val[it] + weight, t.key[it].second);
                }
            });
        }
    }
    auto chain = traceback(bank, 0.0f, [&](const MatchId& id) -> float {   // anchorer.hpp:1724-1741
        if (!E.on) return 0.0f;
        for (uint32_t a : E.snk1)
            for (uint32_t b : E.snk2)
                if ((a == M.back1(id) || reachable(x1, M.back1(id), a)) && (b == M.back2(id) || reachable(x2, M.back2(id), b))) return 0.0f;
        return kMinInf;
    });
    *chain_len = chain.size();
    for (size_t i = 0; i < chain.size(); ++i) {
        chain_out[3 * i] = chain[i].set;
        chain_out[3 * i + 1] = chain[i].i1;
        chain_out[3 * i + 2] = chain[i].i2;
    }
    if (dp_out) bank.for_each([&](const MatchId& id) { dp_out[bank.slot(id)] = bank.dp(id); });
    return 0;
}

int clo_sparse_chain(const cl_base_graph* g1, const cl_base_graph* g2, const clo_match_sets* ms, uint64_t num_match_sets,
                     const clo_chain_params* cp, uint32_t* chain_out, uint64_t* chain_len, float* dp_out) {
    return clo_sparse_chain_ex(g1, g2, ms, num_match_sets, cp, 0, chain_out, chain_len, dp_out);
}

}  // extern "C"
