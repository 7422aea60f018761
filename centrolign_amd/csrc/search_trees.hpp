// search_trees.hpp — static-topology search trees with the reference's observable behaviour
// (include/centrolign/max_search_tree.hpp, include/centrolign/orthogonal_max_search_tree.hpp): implicit complete binary
// tree over the sorted keys, every node pointing at the maximum of its subtree (first one to reach the value keeps
// the pointer), range_max inspecting the split node, then the left search path with its right off-path subtrees,
// then the right search path with its left off-path subtrees, first strictly greater value winning.  Host code of
// the product (used by cl_despecify_indel_breakpoints); the oracle has its own copy.
#ifndef CL_SEARCH_TREES_HPP
#define CL_SEARCH_TREES_HPP

#include <algorithm>
#include <cstdint>
#include <tuple>
#include <utility>
#include <vector>

namespace clhost {

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


}  // namespace clhost

#endif
