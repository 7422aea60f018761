// wfa_host.hpp — the two wavefront heuristics of Stitcher::do_alignment, host algorithms in the
// reference as well:
//   pwfa_po_poa           include/centrolign/alignment.hpp:2299-2338  (pruned graph-graph WFA; route "w", stitcher.hpp:326-339)
//   deletion_wfa_po_poa   include/centrolign/alignment.hpp:2036-2282  (two-sided WFA around one long deletion; routes "ad1/ad2")
// over
//   to_wfa_params :1613-1655, wfa_iteration :1712-1875, wfa_traceback / wfa_traceback_rev :1892-1957,
//   pwfa_po_poa_internal :1959-2034, minmax_distance (minmax_distance.hpp:16-72), target_reachability
//   (target_reachability.hpp:16-32), shortest_path (shortest_path.hpp:32-100).
// Bucket-queue Dijkstra over (node1, node2, component) with FIFO buckets; every choice among equal scores follows from
// the enqueue order, which is reproduced.  SuperbubbleDistanceOracle::min_distance(a, b) is the number of edges on a
// shortest a->b path or -1 (src/test/test_superbubble_distance_oracle.cpp:30-36): computed here by one DP per queried
// source node, memoised.
#ifndef CL_WFA_HOST_HPP
#define CL_WFA_HOST_HPP

#include <algorithm>
#include <cstdint>
#include <deque>
#include <limits>
#include <queue>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace clwfa {

constexpr uint64_t kGap = ~0ull;   // AlignedPair::gap

// one side: nodes 0..n-1, next / previous lists in the reference's order, sources, sinks, labels
struct Side {
    uint64_t n = 0;
    const uint8_t* label = nullptr;
    std::vector<std::vector<uint64_t>> next, prev;
    std::vector<uint64_t> sources, sinks, order;   // order: topological_order.hpp:12-60
};

typedef std::vector<std::pair<uint64_t, uint64_t>> Alignment;

struct WfaParams {
    uint32_t mismatch = 0, gap_open[3] = {0, 0, 0}, gap_extend[3] = {0, 0, 0};
    int npw = 1;
};

inline uint32_t wfa_gcd(uint32_t a, uint32_t b) {   // alignment.hpp:1617-1628
    if (a < b) std::swap(a, b);
    const uint32_t r = a % b;
    return r == 0 ? b : wfa_gcd(b, r);
}

// alignment.hpp:1613-1655
inline WfaParams to_wfa_params(uint32_t match, uint32_t mismatch, const uint32_t* gap_open, const uint32_t* gap_extend, int npw) {
    WfaParams w;
    w.npw = npw;
    w.mismatch = 2 * (match + mismatch);
    uint32_t factor = w.mismatch;
    for (int i = 0; i < npw; ++i) {
        w.gap_open[i] = 2 * gap_open[i];
        w.gap_extend[i] = 2 * gap_extend[i] + match;
        factor = wfa_gcd(factor, w.gap_open[i]);
        factor = wfa_gcd(factor, w.gap_extend[i]);
    }
    if (factor != 1) {
        w.mismatch /= factor;
        for (int i = 0; i < npw; ++i) { w.gap_open[i] /= factor; w.gap_extend[i] /= factor; }
    }
    return w;
}

struct Pos { uint64_t a, b; int comp; };
struct Item { Pos from, to; };
typedef std::deque<std::queue<Item>> BucketQueue;

// HashBackedMap (alignment.hpp:1691-1706): only membership and lookup are used, never iteration
struct BackMap {
    uint64_t n2, comps;
    std::unordered_map<uint64_t, Pos> table;
    BackMap(uint64_t size1, uint64_t size2, int npw) : n2(size2 + 1), comps((uint64_t)(2 * npw + 1)) { (void)size1; }
    uint64_t key(const Pos& p) const { return (p.a * n2 + p.b) * comps + (uint64_t)(p.comp + (int64_t)(comps / 2)); }
    bool count(const Pos& p) const { return table.count(key(p)) != 0; }
    Pos& operator[](const Pos& p) {
        auto it = table.find(key(p));
        if (it == table.end()) it = table.emplace(key(p), Pos{0, 0, 0}).first;   // value-initialised tuple, as operator[] of the reference's map
        return it->second;
    }
};

// alignment.hpp:1712-1875; returns (-1,-1) unless the stop condition was met at the popped position
template <bool Forward, class PruneF, class UpdateF, class Next1F, class Next2F, class StopF, class GreedyF>
inline std::pair<uint64_t, uint64_t> wfa_iteration(BucketQueue& queue, int64_t& queue_min_score, BackMap& backpointer, const Side& g1,
                                                   const Side& g2, const WfaParams& wp, const PruneF& prune, const UpdateF& update,
                                                   const Next1F& next1, const Next2F& next2, const StopF& stop, const GreedyF& greedy) {
    const std::pair<uint64_t, uint64_t> null(~0ull, ~0ull);
    auto enqueue = [&](const Pos& from, uint64_t to1, uint64_t to2, int to_comp, uint64_t penalty) {
        while (queue.size() <= penalty) queue.emplace_back();
        queue[penalty].push(Item{from, Pos{to1, to2, to_comp}});
    };
    while (queue.front().empty()) {
        queue.pop_front();
        ++queue_min_score;
    }
    const Item it = queue.front().front();
    queue.front().pop();
    const Pos here = it.to;
    if (prune(here, queue_min_score) || backpointer.count(here)) return null;
    update(here, queue_min_score);
    backpointer[here] = it.from;
    if (stop(here.a, here.b, here.comp)) return std::make_pair(here.a, here.b);
    if (Forward) {
        if (here.comp == 0) {
            if (greedy(here.a, here.b)) {
                enqueue(here, next1(here.a).front(), next2(here.b).front(), 0, 0);
            } else {
                for (uint64_t n1 : next1(here.a)) {
                    for (uint64_t n2 : next2(here.b)) enqueue(here, n1, n2, 0, g1.label[n1] == g2.label[n2] ? 0 : wp.mismatch);
                    for (int i = 0; i < wp.npw; ++i) enqueue(here, n1, here.b, i + 1, wp.gap_open[i] + wp.gap_extend[i]);
                }
                for (uint64_t n2 : next2(here.b))
                    for (int i = 0; i < wp.npw; ++i) enqueue(here, here.a, n2, -i - 1, wp.gap_open[i] + wp.gap_extend[i]);
            }
        } else {
            enqueue(here, here.a, here.b, 0, 0);
            if (here.comp > 0) {
                for (uint64_t n1 : next1(here.a)) enqueue(here, n1, here.b, here.comp, wp.gap_extend[here.comp - 1]);
            } else {
                for (uint64_t n2 : next2(here.b)) enqueue(here, here.a, n2, here.comp, wp.gap_extend[-here.comp - 1]);
            }
        }
    } else {
        if (here.comp == 0) {
            if (here.a < g1.n && here.b < g2.n) {
                const uint64_t penalty = g1.label[here.a] == g2.label[here.b] ? 0 : wp.mismatch;
                for (uint64_t n1 : next1(here.a))
                    for (uint64_t n2 : next2(here.b)) enqueue(here, n1, n2, 0, penalty);
            }
            for (int i = 0; i < wp.npw; ++i) {
                enqueue(here, here.a, here.b, i + 1, 0);
                enqueue(here, here.a, here.b, -i - 1, 0);
            }
        } else if (here.comp > 0) {
            if (here.a < g1.n)
                for (uint64_t n1 : next1(here.a)) {
                    enqueue(here, n1, here.b, here.comp, wp.gap_extend[here.comp - 1]);
                    enqueue(here, n1, here.b, 0, wp.gap_open[here.comp - 1] + wp.gap_extend[here.comp - 1]);
                }
        } else {
            if (here.b < g2.n)
                for (uint64_t n2 : next2(here.b)) {
                    enqueue(here, here.a, n2, here.comp, wp.gap_extend[-here.comp - 1]);
                    enqueue(here, here.a, n2, 0, wp.gap_open[-here.comp - 1] + wp.gap_extend[-here.comp - 1]);
                }
        }
    }
    return null;
}

// alignment.hpp:1892-1923
inline Alignment wfa_traceback(BackMap& backpointer, uint64_t tb1, uint64_t tb2, const Side& g1, const Side& g2) {
    Alignment aln;
    int comp = 0;
    while (tb1 != g1.n || tb2 != g2.n) {
        const Pos nxt = backpointer[Pos{tb1, tb2, comp}];
        if (nxt.a != tb1 && nxt.b != tb2) aln.emplace_back(tb1, tb2);
        else if (nxt.a != tb1) aln.emplace_back(tb1, kGap);
        else if (nxt.b != tb2) aln.emplace_back(kGap, tb2);
        tb1 = nxt.a; tb2 = nxt.b; comp = nxt.comp;
    }
    std::reverse(aln.begin(), aln.end());
    return aln;
}

// alignment.hpp:1925-1957
inline Alignment wfa_traceback_rev(BackMap& backpointer, uint64_t tb1, uint64_t tb2) {
    Alignment aln;
    int comp = 0;
    Pos nxt = backpointer[Pos{tb1, tb2, comp}];
    while (nxt.a != ~0ull && nxt.b != ~0ull) {
        if (nxt.a != tb1 && nxt.b != tb2) aln.emplace_back(nxt.a, nxt.b);
        else if (nxt.a != tb1) aln.emplace_back(nxt.a, kGap);
        else if (nxt.b != tb2) aln.emplace_back(kGap, nxt.b);
        tb1 = nxt.a; tb2 = nxt.b; comp = nxt.comp;
        nxt = backpointer[Pos{tb1, tb2, comp}];
    }
    return aln;
}

// alignment.hpp:2299-2338 over pwfa_po_poa_internal :1959-2034
inline Alignment pwfa_po_poa(const Side& g1, const Side& g2, const WfaParams& wp, int64_t prune_limit) {
    // minmax_distance from the sources (minmax_distance.hpp:16-72), target_reachability of the sinks (target_reachability.hpp:16-32)
    auto minmax = [](const Side& g) {
        std::vector<std::pair<int64_t, int64_t>> dp(g.n, std::make_pair(std::numeric_limits<int64_t>::max(), (int64_t)-1));
        for (uint64_t v : g.sources) dp[v] = std::make_pair((int64_t)0, (int64_t)0);
        for (uint64_t v : g.order)
            if (dp[v].first != std::numeric_limits<int64_t>::max())
                for (uint64_t w : g.next[v]) {
                    dp[w].first = std::min(dp[w].first, dp[v].first + 1);
                    dp[w].second = std::max(dp[w].second, dp[v].second + 1);
                }
        return dp;
    };
    auto reach = [](const Side& g) {
        std::vector<char> r(g.n, 0);
        for (uint64_t v : g.sinks) r[v] = 1;
        for (size_t i = g.order.size(); i-- > 0;)
            for (uint64_t w : g.next[g.order[i]]) r[g.order[i]] = r[g.order[i]] || r[w];
        return r;
    };
    const auto dists1 = minmax(g1), dists2 = minmax(g2);
    const auto reachable1 = reach(g1), reachable2 = reach(g2);
    int64_t furthest = std::numeric_limits<int64_t>::min() + prune_limit;
    auto prune = [&](const Pos& p, int64_t) {
        if ((p.a < g1.n && !reachable1[p.a]) || (p.b < g2.n && !reachable2[p.b])) return true;
        const int64_t d1 = p.a != g1.n ? dists1[p.a].second : -1, d2 = p.b != g2.n ? dists2[p.b].second : -1;
        return d1 + d2 < furthest - prune_limit;
    };
    auto update = [&](const Pos& p, int64_t) {
        if ((p.a == g1.n || reachable1[p.a]) && (p.b == g2.n || reachable2[p.b])) {
            const int64_t d1 = p.a != g1.n ? dists1[p.a].first : -1, d2 = p.b != g2.n ? dists2[p.b].first : -1;
            furthest = std::max<int64_t>(furthest, d1 + d2);
        }
    };
    BackMap backpointer(g1.n, g2.n, wp.npw);
    int64_t queue_min_score = 0;
    BucketQueue queue;
    queue.emplace_back();
    queue.back().push(Item{Pos{~0ull, ~0ull, 0}, Pos{g1.n, g2.n, 0}});
    auto next1 = [&](uint64_t v) -> const std::vector<uint64_t>& { return v == g1.n ? g1.sources : g1.next[v]; };
    auto next2 = [&](uint64_t v) -> const std::vector<uint64_t>& { return v == g2.n ? g2.sources : g2.next[v]; };
    const std::unordered_set<uint64_t> sink1(g1.sinks.begin(), g1.sinks.end()), sink2(g2.sinks.begin(), g2.sinks.end());
    auto stop = [&](uint64_t a, uint64_t b, int comp) { return (sink1.empty() || sink1.count(a)) && (sink2.empty() || sink2.count(b)) && comp == 0; };
    auto greedy = [&](uint64_t a, uint64_t b) -> bool {
        if (next1(a).size() == 1 && next2(b).size() == 1 && !sink1.count(a) && !sink2.count(b)) return g1.label[next1(a).front()] == g2.label[next2(b).front()];
        return false;
    };
    std::pair<uint64_t, uint64_t> end(~0ull, ~0ull);
    while (end == std::pair<uint64_t, uint64_t>(~0ull, ~0ull))
        end = wfa_iteration<true>(queue, queue_min_score, backpointer, g1, g2, wp, prune, update, next1, next2, stop, greedy);
    return wfa_traceback(backpointer, end.first, end.second, g1, g2);
}

// edges on a shortest from->to path, or -1: SuperbubbleDistanceOracle::min_distance as its test defines it
struct MinDistance {
    const Side& g;
    std::unordered_map<uint64_t, std::vector<int64_t>> memo;
    explicit MinDistance(const Side& side) : g(side) {}
    int64_t operator()(uint64_t from, uint64_t to) {
        auto it = memo.find(from);
        if (it == memo.end()) {
            std::vector<int64_t> d(g.n, -1);
            d[from] = 0;
            for (uint64_t v : g.order)
                if (d[v] >= 0)
                    for (uint64_t w : g.next[v])
                        if (d[w] < 0 || d[v] + 1 < d[w]) d[w] = d[v] + 1;
            it = memo.emplace(from, std::move(d)).first;
        }
        return it->second[to];
    }
};

// shortest_path(graph, from, to), shortest_path.hpp:32-100
inline std::vector<uint64_t> shortest_path(const Side& g, uint64_t from, uint64_t to) {
    const uint64_t INF = (uint64_t)std::numeric_limits<int64_t>::max();
    std::vector<uint64_t> dp(g.n, INF), path;
    dp[from] = 0;
    for (uint64_t v : g.order) {
        const uint64_t thru = dp[v] + 1;
        for (uint64_t w : g.next[v]) dp[w] = std::min(dp[w], thru);
    }
    if (dp[to] == INF) return path;
    path.push_back(to);
    while (dp[path.back()] != 0) {
        bool moved = false;
        for (uint64_t p : g.prev[path.back()])
            if (dp[p] + 1 == dp[path.back()]) { path.push_back(p); moved = true; break; }
        if (!moved) break;
    }
    std::reverse(path.begin(), path.end());
    return path;
}

// ---- two-sided search around one long deletion (alignment.hpp:2036-2282) -----------------------------------------------------------------------------
// `sh` is the short graph (node_id1 of the result), `lg` the long one.  Two Dijkstra fronts, one from the sources and one from the sinks, advance in turn
// (the one with the lower floor first) until a position of one front in match state can be joined to one of the other by a pure deletion in the long graph;
// the search then runs `scope` score units longer — no single step costs more — and the cheapest junction among everything the fronts landed on is taken.

// one front of the search, with what it has landed on: short node -> (long node, score) in match state.  `landed` is ITERATED when the junction is chosen:
// the reference keeps the same std::unordered_map<uint64_t, ...> with the same insertion sequence, which is what makes the iteration order (and with it the
// choice among equally good junctions) the same
struct DeletionFront {
    BackMap back;
    BucketQueue queue;
    int64_t floor = 0;     // the lowest score still queued
    std::unordered_map<uint64_t, std::vector<std::pair<uint64_t, int64_t>>> landed;
    DeletionFront(const Side& sh, const Side& lg, const WfaParams& wp) : back(sh.n, lg.n, wp.npw) { queue.emplace_back(); }
    void seed(uint64_t a, uint64_t b) { queue.back().push(Item{Pos{~0ull, ~0ull, 0}, Pos{a, b, 0}}); }
};

// where the two halves are joined: the short node both fronts reached, the long nodes either side of the deletion
struct DeletionJunction {
    int64_t score = std::numeric_limits<int64_t>::max();
    uint64_t short_node = ~0ull, long_before = ~0ull, long_after = ~0ull;
};

inline Alignment deletion_wfa_po_poa(const Side& sh, const Side& lg, const WfaParams& wp) {
    const int64_t never = std::numeric_limits<int64_t>::max();
    const std::pair<uint64_t, uint64_t> nowhere(~0ull, ~0ull);
    int64_t scope = wp.mismatch;
    for (int i = 0; i < wp.npw; ++i) scope = std::max<int64_t>(scope, (int64_t)wp.gap_open[i] + wp.gap_extend[i]);
    MinDistance long_hops(lg);
    DeletionFront from_sources(sh, lg, wp), from_sinks(sh, lg, wp);
    from_sources.seed(sh.n, lg.n);
    for (uint64_t a : sh.sinks)
        for (uint64_t b : lg.sinks) from_sinks.seed(a, b);
    int64_t search_until = never;      // set once, when the fronts first become joinable

    // a long-graph deletion from `before` to `after` exists (or none is needed)
    auto joinable = [&](uint64_t before, uint64_t after) { return before == after || (before != lg.n && after != lg.n && long_hops(before, after) != -1); };
    // what a front does with a position it settles: record match-state landings, and look for a first junction with the other front
    auto settle = [&](DeletionFront& mine, const DeletionFront& other, bool mine_is_before) {
        return [&mine, &other, mine_is_before, &search_until, &joinable, scope](const Pos& p, int64_t score) {
            if (p.comp == 0) mine.landed[p.a].emplace_back(p.b, score);
            if (search_until != never) return;
            const auto met = other.landed.find(p.a);
            if (met == other.landed.end()) return;
            for (const auto& there : met->second)
                if (mine_is_before ? joinable(p.b, there.first) : joinable(there.first, p.b)) search_until = score + scope;
        };
    };
    const auto settle_fwd = settle(from_sources, from_sinks, true), settle_rev = settle(from_sinks, from_sources, false);
    auto finished = [&](uint64_t, uint64_t, int) { return from_sources.floor >= search_until && from_sinks.floor >= search_until; };
    auto keep_all = [](const Pos&, int64_t) { return false; };
    auto never_greedy = [](uint64_t, uint64_t) { return false; };
    // neighbours with the dummy start (node id n) in front of the sources
    const std::unordered_set<uint64_t> short_sources(sh.sources.begin(), sh.sources.end()), long_sources(lg.sources.begin(), lg.sources.end());
    auto after_short = [&](uint64_t v) -> const std::vector<uint64_t>& { return v == sh.n ? sh.sources : sh.next[v]; };
    auto after_long = [&](uint64_t v) -> const std::vector<uint64_t>& { return v == lg.n ? lg.sources : lg.next[v]; };
    auto before_short = [&](uint64_t v) { std::vector<uint64_t> p = sh.prev[v]; if (short_sources.count(v)) p.push_back(sh.n); return p; };
    auto before_long = [&](uint64_t v) { std::vector<uint64_t> p = lg.prev[v]; if (long_sources.count(v)) p.push_back(lg.n); return p; };

    for (bool done = false; !done;) {
        if (from_sources.floor <= from_sinks.floor)
            done = wfa_iteration<true>(from_sources.queue, from_sources.floor, from_sources.back, sh, lg, wp, keep_all, settle_fwd, after_short, after_long, finished, never_greedy) != nowhere;
        else
            done = wfa_iteration<false>(from_sinks.queue, from_sinks.floor, from_sinks.back, sh, lg, wp, keep_all, settle_rev, before_short, before_long, finished, never_greedy) != nowhere;
    }

    // the cheapest junction: landing scores of both sides plus the cheapest gap component for the hops between them
    auto deletion_cost = [&](int64_t hops) {
        int64_t cost = (int64_t)(wp.gap_open[0] + (uint64_t)wp.gap_extend[0] * (uint64_t)hops);
        for (int i = 1; i < wp.npw; ++i) cost = std::min<int64_t>(cost, (int64_t)(wp.gap_open[i] + (uint64_t)wp.gap_extend[i] * (uint64_t)hops));
        return cost;
    };
    DeletionJunction best;
    for (const auto& at_short : from_sources.landed) {
        const auto met = from_sinks.landed.find(at_short.first);
        if (met == from_sinks.landed.end()) continue;
        for (const auto& before : at_short.second) {
            if (before.first == lg.n) continue;
            for (const auto& after : met->second) {
                if (after.first == lg.n) continue;
                const int64_t hops = long_hops(before.first, after.first);
                if (hops == -1) continue;
                const int64_t total = deletion_cost(hops) + (before.second + after.second);
                if (total < best.score) {
                    best.score = total;
                    best.short_node = at_short.first; best.long_before = before.first; best.long_after = after.first;
                }
            }
        }
    }
    // left half, the deleted stretch of the long graph, right half
    Alignment aln = wfa_traceback(from_sources.back, best.short_node, best.long_before, sh, lg);
    const std::vector<uint64_t> deleted = shortest_path(lg, best.long_before, best.long_after);
    for (size_t i = 1; i < deleted.size(); ++i) aln.emplace_back(kGap, deleted[i]);
    const Alignment right = wfa_traceback_rev(from_sinks.back, best.short_node, best.long_after);
    aln.insert(aln.end(), right.begin(), right.end());
    return aln;
}

}  // namespace clwfa

#endif
