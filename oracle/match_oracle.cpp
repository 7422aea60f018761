/*
 * match_oracle.cpp — TEST INFRASTRUCTURE ONLY (see cl_oracle.h).
 *
 * CPU restatement of centrolign's path match finding: PathMatchFinder::find_matches
 * (include/centrolign/match_finder.hpp:120-212) over PathESA (include/centrolign/path_esa.hpp:81-200) and the
 * minimal-rare-match query of ESA (include/centrolign/esa.hpp:196-494, 610-665).
 *
 * Written for clarity and independence from the product, not speed:
 *   - suffix array: sorted suffixes by prefix doubling with std::sort (the reference uses SA-IS, path_esa.hpp:205+; any
 *     correct suffix sort gives the same array);
 *   - LCP array: Kasai's algorithm exactly as path_esa.hpp:174-200;
 *   - the bottom-up stack traversal with per-node child lists and too-frequent flags exactly as esa.hpp:436-494;
 *   - counts: the number of DISTINCT (component, node id) starts in an interval — what both of the reference's counting
 *     structures (Hui's colour set size, src/esa.cpp:149-300, and the range-unique query, esa.hpp:233-277) return — by
 *     direct scans of the interval with early exits;
 *   - the suffix-link sibling of esa.hpp:352-362 by widening the suffix-array interval around ISA[SA[begin] + 1] while
 *     the LCP stays >= the parent's depth (the child-table / suffix-link machinery is not restated).
 *
 * Parity status: PINNED against the compiled reference (oracle/_ref: ref_find_matches) in tests/test_match_finder.py, and
 * against the committed golden vectors (the match sets the reference's own MSA runs handed to Core::align).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <tuple>
#include <vector>

#include "cl_oracle.h"

namespace {

/* ScoreFunction::anchor_weight (score_function.hpp:47-75), in the operation order of the reference as built (-ffast-math:
 * see chain_oracle.cpp) */
double anchor_weight(const clo_chain_params& cp, uint64_t count1, uint64_t count2, uint64_t length, uint64_t full_length) {
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

struct Index {
    std::vector<uint8_t> text;
    std::vector<uint32_t> id;
    std::vector<uint8_t> comp;
    std::vector<uint32_t> sa, isa, lcp;
    uint64_t n_ids[2];
    std::vector<uint32_t> leaves[2];   /* leaves of component c at suffix-array positions < p */
};

/* path_esa.hpp:92-118 */
void join(const cl_base_graph* gs[2], Index& X) {
    for (int c = 0; c < 2; ++c) {
        const cl_base_graph& g = *gs[c];
        for (uint64_t p = 0; p < g.n_paths; ++p) {
            X.text.push_back(g.label[g.src_id] + 1); X.id.push_back((uint32_t)g.src_id); X.comp.push_back(c);
            for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i) {
                X.text.push_back(g.label[g.path_nodes[i]] + 1); X.id.push_back(g.path_nodes[i]); X.comp.push_back(c);
            }
            X.text.push_back(g.label[g.snk_id] + 1); X.id.push_back((uint32_t)g.snk_id); X.comp.push_back(c);
        }
        X.n_ids[c] = g.n_nodes + 1;
    }
    X.text.push_back(0); X.id.push_back((uint32_t)gs[1]->n_nodes); X.comp.push_back(1);
}

void suffix_sort(const uint8_t* text, uint32_t n, std::vector<uint32_t>& sa, std::vector<uint32_t>& isa) {
    sa.resize(n); isa.resize(n);
    std::iota(sa.begin(), sa.end(), 0u);
    std::vector<uint32_t> rank(n), tmp(n);
    for (uint32_t i = 0; i < n; ++i) rank[i] = text[i];
    for (uint32_t h = 1;; h *= 2) {
        auto key = [&](uint32_t i) { return std::make_pair(rank[i], (uint64_t)i + h < n ? rank[i + h] + 1 : 0u); };
        std::sort(sa.begin(), sa.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
        tmp[sa[0]] = 0;
        for (uint32_t j = 1; j < n; ++j) tmp[sa[j]] = tmp[sa[j - 1]] + (key(sa[j - 1]) < key(sa[j]) ? 1 : 0);
        rank = tmp;
        if (n == 0 || rank[sa[n - 1]] == n - 1 || h >= n) break;
    }
    for (uint32_t j = 0; j < n; ++j) isa[sa[j]] = j;
}

/* path_esa.hpp:174-200 */
void kasai(const uint8_t* text, uint32_t n, const std::vector<uint32_t>& sa, const std::vector<uint32_t>& isa, std::vector<uint32_t>& lcp) {
    lcp.assign(n, 0);
    uint32_t matched = 0;
    for (uint32_t i = 0; i + 1 < n; ++i) {   /* the end sentinel sorts first: it has no previous suffix */
        const uint32_t pos = isa[i], j = sa[pos - 1];
        while (text[i + matched] == text[j + matched]) ++matched;
        lcp[pos] = matched;
        if (matched) --matched;
    }
}

struct Interval { uint32_t begin, end; };

struct Query {
    const Index& X;
    uint64_t max_count;
    std::vector<uint32_t> stamp[2];
    uint32_t tick = 0;
    explicit Query(const Index& x, uint64_t mc) : X(x), max_count(mc) {
        for (int c = 0; c < 2; ++c) stamp[c].assign(X.n_ids[c], 0);
    }
    uint64_t leaves(const Interval& v, int c) const { return X.leaves[c][v.end + 1] - X.leaves[c][v.begin]; }
    /* distinct starts per component; stops early (returning what it has) once the product exceeds max_count */
    void counts(const Interval& v, uint64_t out[2]) {
        ++tick;
        out[0] = out[1] = 0;
        for (uint32_t i = v.begin; i <= v.end; ++i) {
            const uint32_t pos = X.sa[i];
            uint32_t& s = stamp[X.comp[pos]][X.id[pos]];
            if (s != tick) { s = tick; ++out[X.comp[pos]]; if (out[0] * out[1] > max_count) return; }
        }
    }
    /* does some component have more distinct starts in v than c[] ? */
    bool more_frequent(const Interval& v, const uint64_t c[2]) {
        ++tick;
        uint64_t k[2] = {0, 0};
        for (uint32_t i = v.begin; i <= v.end; ++i) {
            const uint32_t pos = X.sa[i];
            uint32_t& s = stamp[X.comp[pos]][X.id[pos]];
            if (s != tick) { s = tick; if (++k[X.comp[pos]] > c[X.comp[pos]]) return true; }
        }
        return false;
    }
};

struct Match { Interval node; uint32_t length; uint64_t count[2]; };

/* esa.hpp:284-494 */
std::vector<Match> minimal_rare_matches(const Index& X, uint64_t max_count) {
    const uint32_t n = (uint32_t)X.text.size();
    Query Q(X, max_count);
    std::vector<Match> matches;
    /* add_matches (:290-431): returns whether any child is too frequent */
    auto add_matches = [&](const Interval& parent, uint32_t parent_depth, const std::vector<Interval>& children, const std::vector<bool>& too_frequent) {
        bool any_too_frequent = false;
        const uint32_t unique_length = parent_depth + 1;
        for (size_t k = 0; k < children.size(); ++k) {
            if (too_frequent[k]) { any_too_frequent = true; continue; }
            const Interval& child = children[k];
            uint64_t c[2] = {0, 0};
            bool over = false;
            if (Q.leaves(child, 0) != 0 && Q.leaves(child, 1) != 0) {   /* :389-394: a zero count leaves every count 0 */
                Q.counts(child, c);
                over = c[0] * c[1] > max_count;
            }
            const uint64_t total = c[0] * c[1];
            if (unique_length == 1) {   /* children of the root (:302-350) */
                if (total > 0 && !over) matches.push_back(Match{child, unique_length, {c[0], c[1]}});
                else any_too_frequent = true;
                continue;
            }
            if (over) { any_too_frequent = true; continue; }   /* :417-419 */
            if (total == 0) continue;
            /* the sibling below the parent's suffix link (:352-362): the interval of the child's string minus its first character */
            const uint32_t q = X.isa[X.sa[child.begin] + 1];
            Interval link{q, q};
            while (link.begin > 0 && X.lcp[link.begin] >= parent_depth) --link.begin;
            while (link.end + 1 < n && X.lcp[link.end + 1] >= parent_depth) ++link.end;
            if (Q.more_frequent(link, c) && Q.more_frequent(parent, c)) matches.push_back(Match{child, unique_length, {c[0], c[1]}});
        }
        return any_too_frequent;
    };
    /* the traversal (:436-494): records of (lcp, left, children, children too frequent) */
    struct Rec { uint32_t lcp, left; std::vector<Interval> children; std::vector<bool> too_frequent; };
    std::vector<Rec> stack;
    stack.push_back(Rec{0, 0, {}, {}});
    const Interval none{0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t i = 1; i < n; ++i) {
        Interval last = none;
        bool has_too_frequent = false;
        uint32_t left = i - 1;
        while (stack.back().lcp > X.lcp[i]) {
            Rec top = std::move(stack.back());
            stack.pop_back();
            last = Interval{top.left, i - 1};
            has_too_frequent = add_matches(last, top.lcp, top.children, top.too_frequent);
            left = top.left;
            if (stack.back().lcp >= X.lcp[i]) {
                stack.back().children.push_back(last);
                stack.back().too_frequent.push_back(has_too_frequent);
                last = none;
                has_too_frequent = false;
            }
        }
        if (stack.back().lcp < X.lcp[i]) {
            stack.push_back(Rec{X.lcp[i], left, {}, {}});
            if (last.begin != none.begin) {
                stack.back().children.push_back(last);
                stack.back().too_frequent.push_back(has_too_frequent);
            }
        }
    }
    while (!stack.empty()) {
        Rec top = std::move(stack.back());
        stack.pop_back();
        const Interval node{top.left, n - 1};
        const bool f = add_matches(node, top.lcp, top.children, top.too_frequent);
        if (!stack.empty()) { stack.back().children.push_back(node); stack.back().too_frequent.push_back(f); }
    }
    return matches;
}

}  // namespace

extern "C" {

int clo_suffix_array_lcp(const uint8_t* text, uint64_t n, uint32_t* sa_out, uint32_t* lcp_out) {
    std::vector<uint32_t> sa, isa, lcp;
    suffix_sort(text, (uint32_t)n, sa, isa);
    kasai(text, (uint32_t)n, sa, isa, lcp);
    std::memcpy(sa_out, sa.data(), n * 4);
    std::memcpy(lcp_out, lcp.data(), n * 4);
    return 0;
}

int clo_find_matches(const cl_base_graph* g1, const cl_base_graph* g2, const clo_chain_params* cp, uint64_t max_count,
                     uint64_t* n_sets_out, uint64_t** rows_out, uint32_t** nodes_out, uint64_t* n_nodes_out) {
    const cl_base_graph* gs[2] = {g1, g2};
    Index X;
    join(gs, X);
    const uint32_t n = (uint32_t)X.text.size();
    suffix_sort(X.text.data(), n, X.sa, X.isa);
    kasai(X.text.data(), n, X.sa, X.isa, X.lcp);
    for (int c = 0; c < 2; ++c) {
        X.leaves[c].assign((size_t)n + 1, 0);
        for (uint32_t p = 0; p < n; ++p) X.leaves[c][p + 1] = X.leaves[c][p] + (X.comp[X.sa[p]] == c);
    }
    std::vector<Match> matches = minimal_rare_matches(X, max_count);
    /* query_index (match_finder.hpp:148-205): positive weight only; walks in suffix-array order, one per distinct start */
    std::vector<uint64_t> rows;
    std::vector<uint32_t> nodes;
    std::vector<uint32_t> stamp[2];
    for (int c = 0; c < 2; ++c) stamp[c].assign(X.n_ids[c], 0);
    uint32_t tick = 0;
    for (const Match& m : matches) {
        if (!(anchor_weight(*cp, m.count[0], m.count[1], m.length, m.length) > 0.0)) continue;
        ++tick;
        std::vector<uint32_t> walks[2];
        uint64_t k[2] = {0, 0};
        for (uint32_t i = m.node.begin; i <= m.node.end; ++i) {   /* esa.hpp:621-663 */
            const uint32_t pos = X.sa[i];
            const int c = X.comp[pos];
            uint32_t& s = stamp[c][X.id[pos]];
            if (s == tick) continue;
            s = tick;
            ++k[c];
            for (uint32_t j = 0; j < m.length; ++j) walks[c].push_back(X.id[pos + j]);
        }
        rows.insert(rows.end(), {k[0], k[1], (uint64_t)m.length, k[0], k[1], (uint64_t)m.length});
        nodes.insert(nodes.end(), walks[0].begin(), walks[0].end());
        nodes.insert(nodes.end(), walks[1].begin(), walks[1].end());
    }
    *n_sets_out = rows.size() / 6;
    *n_nodes_out = nodes.size();
    *rows_out = (uint64_t*)malloc((rows.size() ? rows.size() : 1) * sizeof(uint64_t));
    *nodes_out = (uint32_t*)malloc((nodes.size() ? nodes.size() : 1) * sizeof(uint32_t));
    if (!*rows_out || !*nodes_out) return -1;
    std::memcpy(*rows_out, rows.data(), rows.size() * sizeof(uint64_t));
    std::memcpy(*nodes_out, nodes.data(), nodes.size() * sizeof(uint32_t));
    return 0;
}

void clo_free(void* p) { free(p); }

}  // extern "C"
