// cl_cyclize_api.cpp — cyclisation (the CLI's -c): what lies between the hot path's calls in
// Core::calibrate_anchor_scores_and_identify_bonds (src/core.cpp:196-297) and Core::apply_bonds (:594-648).
//
//   cl_identify_bonds            Bonder::identify_bonds (include/centrolign/bonder.hpp:116-522) with the CLI's bond algorithm
//                                (LongestNearOptDevConstrained, bonder.hpp:63; no CLI switch selects another), trim_partition_ends
//                                (src/bonder.cpp:595-799) and deduplicate_self_bonds (:473-551), on a LEAF against itself — the only
//                                place the reference calls it (src/core.cpp:229-234)
//   cl_leaf_calibrate            the per-leaf step of the calibration that also keeps what the bond rounds need (src/core.cpp:122-175)
//   cl_leaf_bond_alignments      the tandem-duplication rounds of one leaf (:199-296): masked anchor chain -> bonds -> internal_stitch
//   cl_simplify_bubbles          simplify_bubbles (src/modify_graph.cpp:165-382) + purge_uncovered_nodes (:89-163)
//   cl_apply_bonds               Core::apply_bonds up to the polishing step (:594-645): path positions -> node ids, internal_fuse, simplify_bubbles
//
// Host code, like the reference's; the device work is inside the calls it makes (cl_anchor_chain_masked, cl_internal_stitch).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "cl_internal.hpp"

namespace {

constexpr uint64_t kGap = ~(uint64_t)0;

struct Seg { double len = 0, opt = 0, sec = 0; };   // (length, opt segment score, secondary segment score)

// PartitionClient::traceback (include/centrolign/partition_client.hpp:33-55)
std::vector<std::pair<size_t, size_t>> partition_traceback(const std::vector<std::pair<double, double>>& dp, const std::vector<size_t>& back, size_t at) {
    std::vector<std::pair<size_t, size_t>> part;
    bool inside = true;
    while (at > 0) {
        if (inside) {
            const size_t prev = back[at];
            part.emplace_back(prev, at);
            at = prev;
            inside = false;
        } else {
            inside = dp[at].first == dp[at - 1].second;
            --at;
        }
    }
    std::reverse(part.begin(), part.end());
    return part;
}

// Bonder::longest_deviation_constrained_partition (src/bonder.cpp:801-923): intervals of shared segments whose secondary score keeps up
// with the optimal chain's, whose indel drift stays within deviation_drift_factor * sqrt(length) and whose distance from the main
// diagonal is at least their length minus separation_drift_factor * sqrt(length); total length above min_length per interval maximised.
// sep[i]: distance between the optimal and the secondary copy of shared segment i on the leaf (SuperbubbleDistanceOracle::min_distance on
// a chain: the difference of the positions, whichever way round it is positive)
std::vector<std::pair<size_t, size_t>> deviation_constrained_partition(const std::vector<Seg>& shared, const std::vector<Seg>& between,
                                                                       const std::vector<std::pair<int64_t, int64_t>>& deviation,
                                                                       const std::vector<double>& sep, const cl_bond_params& bp) {
    const double mininf = std::numeric_limits<double>::lowest();
    const size_t n = shared.size();
    std::vector<std::pair<double, double>> dp(n + 1, std::make_pair(mininf, mininf));   // (excluded, included)
    dp[0] = std::make_pair(0.0, 0.0);
    std::vector<size_t> back(n + 1, (size_t)-1);
    size_t best = 0;
    for (size_t i = 1; i <= n; ++i) {
        dp[i].first = std::max(dp[i - 1].first, dp[i - 1].second);
        const double separation = sep[i - 1];
        double run_len = 0.0, run_opt = 0.0, run_sec = 0.0;
        int64_t dev_opt = 0, dev_sec = 0, lo = 0, hi = 0;
        for (size_t j = i; j-- > 0;) {
            run_len += shared[j].len; run_opt += shared[j].opt; run_sec += shared[j].sec;
            if (j + 1 != i) {
                run_len += between[j].len; run_opt += between[j].opt; run_sec += between[j].sec;
                dev_opt += deviation[j].first; dev_sec += deviation[j].second;
            }
            lo = std::min(lo, dev_opt - dev_sec);
            hi = std::max(hi, dev_opt - dev_sec);
            const double root = sqrt(run_len);
            if (run_sec >= bp.min_opt_proportion * run_opt && (double)(hi - lo) <= root * bp.deviation_drift_factor &&
                separation >= run_len - root * bp.separation_drift_factor) {
                const double score = dp[j].first + run_len - bp.min_length;
                if (score > dp[i].second) { dp[i].second = score; back[i] = j; }
            }
        }
        if (dp[i].second > dp[best].second) best = i;
    }
    return partition_traceback(dp, back, best);
}

// Bonder::trim_partition_ends (src/bonder.cpp:595-799): the ends of an interval are cut back until a window of trim_window_proportion *
// min_length at each end meets the score proportion by itself
void trim_partition_ends(std::vector<std::pair<size_t, size_t>>& part, const std::vector<Seg>& shared, const std::vector<Seg>& between, const cl_bond_params& bp) {
    const double window = bp.trim_window_proportion * bp.min_length;
    for (auto& iv : part) {
        double len = shared[iv.first].len, opt = shared[iv.first].opt, sec = shared[iv.first].sec, part_opt = 0.0, part_sec = 0.0;
        size_t wend = iv.first + 1;
        // grows the window to the right while whole (intervening, shared) steps fit; the step that does not fit contributes a fraction
        auto grow_right = [&]() {
            while (wend < iv.second) {
                const double added = between[wend - 1].len + shared[wend].len;
                if (len + added > window) {
                    if (len + between[wend - 1].len < window) {
                        const double frac = (window - len - between[wend - 1].len) / shared[wend].len;
                        part_opt = between[wend - 1].opt + frac * shared[wend].opt;
                        part_sec = between[wend - 1].sec + frac * shared[wend].sec;
                    } else {
                        const double frac = (window - len) / between[wend - 1].len;
                        part_opt = frac * between[wend - 1].opt;
                        part_sec = frac * between[wend - 1].sec;
                    }
                    break;
                }
                len += added;
                opt += between[wend - 1].opt + shared[wend].opt;
                sec += between[wend - 1].sec + shared[wend].sec;
                ++wend;
            }
        };
        grow_right();
        while (iv.first < iv.second && (sec + part_sec) < bp.min_opt_proportion * (opt + part_opt)) {
            len -= shared[iv.first].len; opt -= shared[iv.first].opt; sec -= shared[iv.first].sec;
            if (iv.first + 1 != wend) { len -= between[iv.first].len; opt -= between[iv.first].opt; sec -= between[iv.first].sec; }
            ++iv.first;
            part_opt = 0.0; part_sec = 0.0;
            grow_right();
        }
        if (iv.first == iv.second) continue;
        len = shared[iv.second - 1].len; opt = shared[iv.second - 1].opt; sec = shared[iv.second - 1].sec;
        size_t wbeg = iv.second - 1;
        part_opt = 0.0; part_sec = 0.0;
        auto grow_left = [&]() {
            while (wbeg > iv.first) {
                const double added = between[wbeg - 1].len + shared[wbeg - 1].len;
                if (len + added > window) {
                    if (len + between[wbeg - 1].len < window) {
                        const double frac = (window - len - between[wbeg - 1].len) / shared[wbeg - 1].len;
                        part_opt = between[wbeg - 1].opt + frac * shared[wbeg - 1].opt;
                        part_sec = between[wbeg - 1].sec + frac * shared[wbeg - 1].sec;
                    } else {
                        const double frac = (window - len) / between[wbeg - 1].len;
                        part_opt = frac * between[wbeg - 1].opt;
                        part_sec = frac * between[wbeg - 1].sec;
                    }
                    break;
                }
                len += added;
                --wbeg;
                opt += between[wbeg].opt + shared[wbeg].opt;
                sec += between[wbeg].sec + shared[wbeg].sec;
            }
        };
        grow_left();
        while (iv.first < iv.second && (sec + part_sec) < bp.min_opt_proportion * (opt + part_opt)) {
            --iv.second;
            len -= shared[iv.second].len; opt -= shared[iv.second].opt; sec -= shared[iv.second].sec;
            // (the reference takes off intervening_segments[interval.second], the segment BEHIND the entry it removes, not the one in front of it,
            // src/bonder.cpp:753-757; for the last shared segment that index is one past the end of its vector: what it reads there is not defined —
            // mostly allocator bookkeeping, denormal or zero as a double, sometimes not: the reference's -c output changes with MALLOC_PERTURB_ and, on some inputs,
            // from run to run (scripts/fuzz_msa.py, profiles/r06_fuzz_msa.json) — and counts as nothing here)
            if (wbeg != iv.second && iv.second < between.size()) { len -= between[iv.second].len; opt -= between[iv.second].opt; sec -= between[iv.second].sec; }
            else if (wbeg != iv.second) {
                ++cl_fallbacks.bond_trims_past_the_end;   // (cl_fallback_counters: a run that stood here has no defined counterpart in the reference)
                if (getenv("CL_BOND_DEBUG")) fprintf(stderr, "[identify_bonds] end trim of the LAST shared segment (%zu of %zu): the reference reads intervening_segments[%zu] of %zu here\n", iv.second, shared.size(), iv.second, between.size());
            }
            part_opt = 0.0; part_sec = 0.0;
            grow_left();
        }
    }
    part.erase(std::remove_if(part.begin(), part.end(), [](const std::pair<size_t, size_t>& p) { return p.first == p.second; }), part.end());
}

struct Bond { uint64_t offset1, offset2, length; double score; };
using BondInterval = std::vector<Bond>;

struct ChainView {
    const cl_chain_anchors* c;
    uint64_t size() const { return c->n; }
    uint64_t len(uint64_t a) const { return c->walk_off[a + 1] - c->walk_off[a]; }
    uint32_t w1(uint64_t a, uint64_t i) const { return c->walk1[c->walk_off[a] + i]; }
    uint32_t w2(uint64_t a, uint64_t i) const { return c->walk2[c->walk_off[a] + i]; }
};

// Bonder::identify_bonds (bonder.hpp:116-452) for a leaf against itself: graph 1 and graph 2 are the same single-path graph, a node's
// step on its path is its id (make_base_graph numbers the bases in order, src/modify_graph.cpp:30-45)
std::vector<BondInterval> identify_bonds(uint64_t n_nodes, const ChainView& opt, const ChainView& sec, const cl_bond_params& bp) {
    std::vector<BondInterval> bonds;
    for (int on_graph1 = 1; on_graph1 >= 0; --on_graph1) {
        auto proj = [&](const ChainView& ch, uint64_t a, uint64_t i) { return on_graph1 ? ch.w1(a, i) : ch.w2(a, i); };
        auto bond = [&](const ChainView& ch, uint64_t a, uint64_t i) { return on_graph1 ? ch.w2(a, i) : ch.w1(a, i); };
        // where the optimal chain's anchors lie on the projecting graph
        std::vector<std::pair<size_t, size_t>> at(n_nodes, std::make_pair((size_t)-1, (size_t)-1));
        for (uint64_t i = 0; i < opt.size(); ++i)
            for (uint64_t j = 0; j < opt.len(i); ++j) at[proj(opt, i, j)] = std::make_pair((size_t)i, (size_t)j);
        // maximal runs the two chains share there: (secondary anchor, index on it, optimal anchor, index on it, length)
        struct Shared { size_t i, j, k, l, len; };
        std::vector<Shared> sh;
        for (uint64_t i = 0; i < sec.size(); ++i) {
            size_t pk = (size_t)-1, pl = (size_t)-1;
            for (uint64_t j = 0; j < sec.len(i); ++j) {
                const size_t k = at[proj(sec, i, j)].first, l = at[proj(sec, i, j)].second;
                if (k != (size_t)-1) {
                    if (pk == k && pl == l - 1) ++sh.back().len;
                    else sh.push_back({(size_t)i, (size_t)j, k, l, 1});
                }
                pk = k; pl = l;
            }
        }
        { std::vector<std::pair<size_t, size_t>>().swap(at); }
        if (sh.empty()) continue;
        // between consecutive anchors of the optimal chain on the projecting graph: extract_graphs_between + source_sink_minmax on a chain,
        // i.e. the EDGES of the stretch in between (minmax_distance starts its sources at 0, minmax_distance.hpp:24-27), 0 for an empty one
        std::vector<double> dist_between(opt.size() ? opt.size() - 1 : 0);
        for (uint64_t i = 0; i + 1 < opt.size(); ++i) {
            const int64_t a = proj(opt, i, opt.len(i) - 1), b = proj(opt, i + 1, 0);
            dist_between[i] = b > a + 1 ? (double)(b - a - 2) : 0.0;
        }
        std::vector<Seg> shared(sh.size()), between(sh.size() - 1);
        std::vector<std::pair<int64_t, int64_t>> deviation(between.size(), std::make_pair((int64_t)0, (int64_t)0));
        std::vector<double> sep(sh.size());
        for (size_t x = 0; x < sh.size(); ++x) {
            // SuperbubbleDistanceOracle::min_distance(optimal copy, secondary copy), either way round (src/bonder.cpp:838-850)
            const int64_t a = bond(opt, sh[x].k, sh[x].l), b = bond(sec, sh[x].i, sh[x].j);
            sep[x] = (double)(b >= a ? b - a : a - b);
        }
        for (size_t x = 0; x < sh.size(); ++x) {
            const Shared& s = sh[x];
            shared[x].len = (double)s.len;
            shared[x].opt = ((double)s.len * opt.c->score[s.k]) / (double)opt.len(s.k);
            shared[x].sec = ((double)s.len * sec.c->score[s.i]) / (double)sec.len(s.i);
            if (x == 0) continue;
            Seg& bt = between[x - 1];
            const Shared& p = sh[x - 1];
            if (p.k == s.k) {
                bt.len = (double)(s.l - p.l - p.len);
                bt.opt = (bt.len * opt.c->score[s.k]) / (double)opt.len(s.k);
            } else {
                size_t offset = p.l + p.len;
                for (size_t a = p.k; a <= s.k; ++a) {
                    const size_t sub = a == s.k ? s.l : opt.len(a) - offset;
                    bt.len += (double)sub;
                    bt.opt += ((double)sub * opt.c->score[a]) / (double)opt.len(a);
                    if (a != s.k) {
                        bt.len += dist_between[a];
                        if (bp.include_gap_scores) bt.opt += opt.c->gap_score_after[a];
                        deviation[x - 1].first += opt.c->gap_after[a];
                    }
                    offset = 0;
                }
            }
            if (p.i == s.i) {
                bt.sec = ((double)(s.j - p.j - p.len) * sec.c->score[s.i]) / (double)sec.len(s.i);
            } else {
                size_t offset = p.j + p.len;
                for (size_t a = p.i; a <= s.i; ++a) {
                    const size_t sub = a == s.i ? s.j : sec.len(a) - offset;
                    bt.sec += ((double)sub * sec.c->score[a]) / (double)sec.len(a);
                    if (a != s.i) {
                        if (bp.include_gap_scores) bt.sec += sec.c->gap_score_after[a];
                        deviation[x - 1].second += sec.c->gap_after[a];
                    }
                    offset = 0;
                }
            }
        }
        auto part = deviation_constrained_partition(shared, between, deviation, sep, bp);
        trim_partition_ends(part, shared, between, bp);
        for (const auto& iv : part) {
            bonds.emplace_back();
            BondInterval& bi = bonds.back();
            for (size_t x = iv.first; x < iv.second; ++x) {
                const Shared& s = sh[x];
                for (size_t y = 0; y < s.len; ++y) {
                    const uint64_t o1 = bond(opt, s.k, s.l + y), o2 = bond(sec, s.i, s.j + y);
                    // (a shared run always opens a bond of its own: the reference resets its current path ids per run, bonder.hpp:408-419)
                    if (bi.empty() || y == 0 || bi.back().offset1 + bi.back().length != o1 || bi.back().offset2 + bi.back().length != o2) {
                        if (!bi.empty()) bi.back().score = ((double)bi.back().length * sec.c->score[s.i]) / (double)sec.len(s.i);
                        bi.push_back({o1, o2, 1, 0.0});
                    } else {
                        ++bi.back().length;
                    }
                    bi.back().score = ((double)bi.back().length * sec.c->score[s.i]) / (double)sec.len(s.i);
                }
            }
        }
    }
    return bonds;
}

// Bonder::deduplicate_self_bonds (src/bonder.cpp:473-551): of two intervals that cover about the same pair of stretches (either way round),
// the one with more bonded bases stays
void deduplicate_self_bonds(std::vector<BondInterval>& bonds, const cl_bond_params& bp) {
    const int64_t slosh = (int64_t)ceil(bp.deduplication_slosh_proportion * bp.min_length);
    auto match_or_include = [&](int64_t b1, int64_t e1, int64_t b2, int64_t e2) {
        return (b1 - slosh <= b2 && e1 + slosh >= e2) || (b2 - slosh <= b1 && e2 + slosh >= e1);
    };
    std::vector<char> keep(bonds.size(), 1);
    for (size_t i = 0; i < bonds.size(); ++i) {
        if (!keep[i]) continue;
        const BondInterval& a = bonds[i];
        const int64_t b11 = a.front().offset1, b21 = a.front().offset2, e11 = a.back().offset1 + a.back().length, e21 = a.back().offset2 + a.back().length;
        for (size_t j = i + 1; j < bonds.size(); ++j) {
            const BondInterval& b = bonds[j];
            const int64_t b12 = b.front().offset1, b22 = b.front().offset2, e12 = b.back().offset1 + b.back().length, e22 = b.back().offset2 + b.back().length;
            if ((match_or_include(b11, e11, b12, e12) && match_or_include(b21, e21, b22, e22)) ||
                (match_or_include(b11, e11, b22, e22) && match_or_include(b21, e21, b12, e12))) {
                uint64_t t1 = 0, t2 = 0;
                for (const Bond& x : a) t1 += x.length;
                for (const Bond& x : b) t2 += x.length;
                if (t1 > t2) keep[j] = 0;
                else { keep[i] = 0; break; }
            }
        }
    }
    size_t w = 0;
    for (size_t i = 0; i < bonds.size(); ++i) if (keep[i]) { if (w != i) bonds[w] = std::move(bonds[i]); ++w; }
    bonds.resize(w);
}

int bonds_out(const std::vector<BondInterval>& bonds, cl_bonds* out) {
    uint64_t total = 0;
    for (const auto& b : bonds) total += b.size();
    out->n_intervals = bonds.size();
    out->interval_off = (uint64_t*)malloc((bonds.size() + 1) * sizeof(uint64_t));
    out->offset1 = (uint64_t*)malloc((total ? total : 1) * sizeof(uint64_t));
    out->offset2 = (uint64_t*)malloc((total ? total : 1) * sizeof(uint64_t));
    out->length = (uint64_t*)malloc((total ? total : 1) * sizeof(uint64_t));
    out->score = (double*)malloc((total ? total : 1) * sizeof(double));
    if (!out->interval_off || !out->offset1 || !out->offset2 || !out->length || !out->score) { cl_bonds_free(out); return CL_ERR_OUT_OF_MEMORY; }
    uint64_t at = 0;
    for (size_t i = 0; i < bonds.size(); ++i) {
        out->interval_off[i] = at;
        for (const Bond& b : bonds[i]) { out->offset1[at] = b.offset1; out->offset2[at] = b.offset2; out->length[at] = b.length; out->score[at] = b.score; ++at; }
    }
    out->interval_off[bonds.size()] = at;
    return CL_OK;
}


// the caller's std::vector<match_set_t> after an anchor_chain call: position k holds the set that was at order[k] (anchorer.hpp:1131-1168)
std::unique_ptr<cl_owned_match_sets> reordered_sets(const cl_match_sets& v, const uint64_t* order) {
    std::unique_ptr<cl_owned_match_sets> o(new cl_owned_match_sets());
    for (uint64_t k = 0; k < v.n_sets; ++k) {
        const uint64_t s = order[k];
        for (int side = 0; side < 2; ++side) {
            const uint64_t* set_off = side ? v.set_off2 : v.set_off1;
            const uint64_t* walk_off = side ? v.walk_off2 : v.walk_off1;
            const uint32_t* nodes = side ? v.nodes2 : v.nodes1;
            auto& o_set = side ? o->set_off2 : o->set_off1;
            auto& o_walk = side ? o->walk_off2 : o->walk_off1;
            auto& o_nodes = side ? o->nodes2 : o->nodes1;
            for (uint64_t w = set_off[s]; w < set_off[s + 1]; ++w) {
                o_nodes.insert(o_nodes.end(), nodes + walk_off[w], nodes + walk_off[w + 1]);
                o_walk.push_back(o_nodes.size());
            }
            o_set.push_back(o_walk.size() - 1);
        }
        o->count1.push_back(v.count1[s]); o->count2.push_back(v.count2[s]); o->full_length.push_back(v.full_length[s]);
    }
    return o;
}

// purge_uncovered_nodes (src/modify_graph.cpp:89-163): nodes no path visits (the sentinels excepted) are dropped, the others keep their order
bool purge_uncovered_impl(cl_owned_base_graph& g) {
    const uint64_t n = g.label.size();
    std::vector<char> covered(n, 0);
    covered[g.src_id] = covered[g.snk_id] = 1;
    for (uint32_t v : g.path_nodes) covered[v] = 1;
    bool all = true;
    for (uint64_t v = 0; v < n && all; ++v) all = covered[v];
    if (all) return false;
    std::vector<uint64_t> removed_before(n + 1, 0);
    for (uint64_t v = 0; v < n; ++v) removed_before[v + 1] = removed_before[v] + (covered[v] ? 0 : 1);
    auto tr = [&](uint64_t v) { return v - removed_before[v]; };
    cl_owned_base_graph p;
    const uint64_t m = n - removed_before[n];
    std::vector<std::vector<uint32_t>> next(m), prev(m);
    for (uint64_t v = 0; v < n; ++v) {
        if (!covered[v]) continue;
        p.label.push_back(g.label[v]);
        for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) {
            const uint64_t w = g.next_idx[e];
            if (covered[w]) { next[tr(v)].push_back((uint32_t)tr(w)); prev[tr(w)].push_back((uint32_t)tr(v)); }
        }
    }
    p.next_off.assign(1, 0); p.prev_off.assign(1, 0);
    for (uint64_t v = 0; v < m; ++v) {
        p.next_idx.insert(p.next_idx.end(), next[v].begin(), next[v].end()); p.next_off.push_back(p.next_idx.size());
        p.prev_idx.insert(p.prev_idx.end(), prev[v].begin(), prev[v].end()); p.prev_off.push_back(p.prev_idx.size());
    }
    p.path_off = g.path_off;
    p.path_nodes.reserve(g.path_nodes.size());
    for (uint32_t v : g.path_nodes) p.path_nodes.push_back((uint32_t)tr(v));
    p.src_id = tr(g.src_id); p.snk_id = tr(g.snk_id);
    g = std::move(p);
    return true;
}

// simplify_bubbles (src/modify_graph.cpp:165-382).  The reference walks its snarl decomposition and acts on the snarls that are acyclic, hold
// no nested snarl and whose every allele is a plain run of nodes from the snarl's source to its sink: there, alleles that spell the same
// sequence are merged by moving every path onto the first of them (in the source's next order).  Such a snarl is recognised directly: a node
// s, not a sentinel, with two or more successors, each of which starts a run of nodes with one predecessor and one successor apiece that ends
// in one common node t != s whose predecessors are exactly the ends of those runs (and s itself for a direct edge).  In the cactus graph the
// reference builds (include/centrolign/cactus.hpp:138-200) the two adjacency components at s's outgoing side and t's incoming side are joined
// by the alleles and, around the graph made circular, by the rest: three edge-disjoint connections, one cactus node, the alleles its self
// loops, and the edges that end in s and start in t consecutive on its cycle — i.e. (s, t) is reported as a snarl; structures that touch a
// sentinel are dropped (structure_tree.hpp:164-169).  Snarls are disjoint but for their boundaries and a merge only rewrites path steps, so
// the order in which they are met does not matter.
bool simplify_bubbles_impl(cl_owned_base_graph& g, std::string& error) {
    const uint64_t n = g.label.size();
    auto outdeg = [&](uint64_t v) { return g.next_off[v + 1] - g.next_off[v]; };
    auto indeg = [&](uint64_t v) { return g.prev_off[v + 1] - g.prev_off[v]; };
    // StepIndex (step_index.hpp:36-44) of the graph as it comes in
    std::vector<uint64_t> step_off(n + 1, 0);
    for (uint32_t v : g.path_nodes) ++step_off[v + 1];
    for (uint64_t v = 0; v < n; ++v) step_off[v + 1] += step_off[v];
    std::vector<uint64_t> step_at(g.path_nodes.size());   // global index into path_nodes, ascending per node: (path, step) order
    {
        std::vector<uint64_t> fill(step_off.begin(), step_off.end() - 1);
        for (uint64_t i = 0; i < g.path_nodes.size(); ++i) step_at[fill[g.path_nodes[i]]++] = i;
    }
    std::vector<uint64_t> path_of(g.path_nodes.size());
    for (uint64_t p = 0; p + 1 < g.path_off.size(); ++p) for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i) path_of[i] = p;
    bool did = false;
    std::vector<std::vector<uint32_t>> alleles;
    for (uint64_t s = 0; s < n; ++s) {
        const uint64_t k = outdeg(s);
        if (k < 2 || s == g.src_id || s == g.snk_id) continue;
        alleles.assign(k, {});
        uint64_t t = ~(uint64_t)0;
        bool ok = true;
        for (uint64_t a = 0; a < k && ok; ++a) {
            uint64_t v = g.next_idx[g.next_off[s] + a];
            while (v != s && indeg(v) == 1 && outdeg(v) == 1) { alleles[a].push_back((uint32_t)v); v = g.next_idx[g.next_off[v]]; }
            if (t == ~(uint64_t)0) t = v;
            ok = v == t;
        }
        if (!ok || t == s || t == g.src_id || t == g.snk_id || indeg(t) != k) continue;
        // every predecessor of t is the end of one of the runs (or s): with k runs ending in t and k predecessors that is the case unless two
        // successors of s are the same node (the graphs here have no parallel edges)
        for (uint64_t i = 1; i < k; ++i) {
            if (alleles[i].empty()) continue;
            uint64_t first = i;   // the first allele, in s's next order, that spells the same sequence
            for (uint64_t j = 0; j < i; ++j) {
                if (alleles[j].size() != alleles[i].size()) continue;
                bool same = true;
                for (size_t x = 0; x < alleles[i].size() && same; ++x) same = g.label[alleles[j][x]] == g.label[alleles[i][x]];
                if (same) { first = j; break; }
            }
            if (first == i) continue;
            const uint32_t head = alleles[i].front();
            for (uint64_t e = step_off[head]; e < step_off[head + 1]; ++e) {
                const uint64_t at = step_at[e], p = path_of[at];
                if (at + alleles[i].size() > g.path_off[p + 1]) { error = "cannot assign subpath past the end of a path"; return false; }
                for (size_t x = 0; x < alleles[i].size(); ++x) g.path_nodes[at + x] = alleles[first][x];
                did = true;
            }
        }
    }
    if (did) purge_uncovered_impl(g);
    return true;
}

cl_owned_base_graph owned_copy(const cl_base_graph* g) {
    cl_owned_base_graph o;
    o.label.assign(g->label, g->label + g->n_nodes);
    o.next_off.assign(g->next_off, g->next_off + g->n_nodes + 1);
    o.prev_off.assign(g->prev_off, g->prev_off + g->n_nodes + 1);
    o.next_idx.assign(g->next_idx, g->next_idx + g->next_off[g->n_nodes]);
    o.prev_idx.assign(g->prev_idx, g->prev_idx + g->prev_off[g->n_nodes]);
    o.path_off.assign(g->path_off, g->path_off + g->n_paths + 1);
    o.path_nodes.assign(g->path_nodes, g->path_nodes + g->path_off[g->n_paths]);
    o.src_id = g->src_id; o.snk_id = g->snk_id;
    return o;
}

}  // namespace

bool cl_purge_uncovered(cl_owned_base_graph& g) { return purge_uncovered_impl(g); }

extern "C" {

void cl_bond_params_default(cl_bond_params* p) {   // src/parameters.cpp:91-97
    if (!p) return;
    p->min_opt_proportion = 0.2;
    p->include_gap_scores = 1;
    p->min_length = 100000.0;
    p->deviation_drift_factor = 150.0;
    p->separation_drift_factor = 50.0;
    p->deduplication_slosh_proportion = 0.1;
    p->trim_window_proportion = 0.1;
}

void cl_bonds_free(cl_bonds* b) {
    if (!b) return;
    free(b->interval_off); free(b->offset1); free(b->offset2); free(b->length); free(b->score);
    memset(b, 0, sizeof(*b));
}

int cl_identify_bonds(const cl_base_graph* leaf, const cl_chain_anchors* opt_chain, const cl_chain_anchors* secondary_chain, const cl_bond_params* params,
                      int deduplicate, cl_bonds* out) {
    if (!leaf || !opt_chain || !secondary_chain || !params || !out) return CL_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    if (leaf->n_paths != 1) return CL_ERR_INVALID_ARGUMENT;
    for (const cl_chain_anchors* c : {opt_chain, secondary_chain}) {
        if (c->n && (!c->walk_off || !c->walk1 || !c->walk2 || !c->score || !c->gap_after || !c->gap_score_after)) return CL_ERR_INVALID_ARGUMENT;
        for (uint64_t a = 0; a < c->n; ++a) {
            if (c->walk_off[a + 1] <= c->walk_off[a]) return CL_ERR_INVALID_ARGUMENT;
            for (uint64_t i = c->walk_off[a]; i < c->walk_off[a + 1]; ++i)
                if (c->walk1[i] >= leaf->n_nodes || c->walk2[i] >= leaf->n_nodes) return CL_ERR_INVALID_ARGUMENT;
        }
    }
    auto bonds = identify_bonds(leaf->n_nodes, ChainView{opt_chain}, ChainView{secondary_chain}, *params);
    if (deduplicate) deduplicate_self_bonds(bonds, *params);
    return bonds_out(bonds, out);
}

}  // extern "C"

struct cl_leaf_calibration {
    cl_owned_match_sets* matches = nullptr;   // the leaf's matches against itself (second copy under sentinels 7 / 8)
    cl_anchor_chain_result chain{};           // the main-diagonal chain the scale was estimated on
    double scale = 0.0;
};

extern "C" int cl_estimate_score_scale_chain(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                                             const cl_anchor_params* ap, cl_anchor_chain_result* out);   // cl_chain_api.cpp

extern "C" {

void cl_leaf_calibration_free(cl_leaf_calibration* c) {
    if (!c) return;
    cl_owned_match_sets_free(c->matches);
    cl_anchor_chain_result_free(&c->chain);
    delete c;
}

// the per-leaf step of Core::calibrate_anchor_scores_and_identify_bonds (src/core.cpp:122-175), keeping the matches and the chain when the
// tandem-duplication rounds will want them (:168-172)
int cl_leaf_calibrate(cl_context* ctx, const cl_base_graph* leaf, const cl_match_params* mp, const cl_anchor_params* ap, double* scale_out,
                      cl_leaf_calibration** memo_out) {
    cl_bind_device(ctx);
    if (!ctx || !leaf || !mp || !ap || !scale_out) return CL_ERR_INVALID_ARGUMENT;
    if (memo_out) *memo_out = nullptr;
    if (leaf->n_nodes == 0 || leaf->src_id >= leaf->n_nodes || leaf->snk_id >= leaf->n_nodes) return CL_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> lab1(leaf->label, leaf->label + leaf->n_nodes), lab2(lab1);
    lab1[leaf->src_id] = 5; lab1[leaf->snk_id] = 6;
    lab2[leaf->src_id] = 7; lab2[leaf->snk_id] = 8;
    cl_base_graph a = *leaf, b = *leaf;
    a.label = lab1.data();
    b.label = lab2.data();
    std::unique_ptr<cl_leaf_calibration> memo(new cl_leaf_calibration());
    int rc = cl_find_matches(ctx, &a, &b, mp, &memo->matches, nullptr);
    if (rc) return rc;
    cl_match_sets v;
    cl_owned_match_sets_view(memo->matches, &v);
    // the main-diagonal subset (:135-148): one set per graph-1 walk, matched to itself, counts and full length retained
    const uint64_t n_walks = v.set_off1[v.n_sets];
    std::vector<uint64_t> set_off(n_walks + 1), count1(n_walks), count2(n_walks), full_length(n_walks);
    for (uint64_t w = 0; w <= n_walks; ++w) set_off[w] = w;
    for (uint64_t s = 0; s < v.n_sets; ++s)
        for (uint64_t w = v.set_off1[s]; w < v.set_off1[s + 1]; ++w) { count1[w] = v.count1[s]; count2[w] = v.count2[s]; full_length[w] = v.full_length[s]; }
    cl_match_sets diag{n_walks, set_off.data(), v.walk_off1, v.nodes1, set_off.data(), v.walk_off1, v.nodes1, count1.data(), count2.data(), full_length.data()};
    rc = cl_estimate_score_scale_chain(ctx, &a, &a, &diag, ap, &memo->chain);
    if (rc) { cl_leaf_calibration_free(memo.release()); return rc; }
    memo->scale = memo->chain.scale;
    *scale_out = memo->scale;
    if (memo_out) *memo_out = memo.release();
    else cl_leaf_calibration_free(memo.release());
    return CL_OK;
}

void cl_alignment_list_free(cl_alignment_list* l) {
    if (!l) return;
    for (uint64_t i = 0; i < l->n; ++i) free(l->alignments[i].pairs);
    free(l->alignments);
    memset(l, 0, sizeof(*l));
}

// the tandem-duplication rounds of one leaf (src/core.cpp:199-296): the next-best chain with everything found so far masked, its bonds against
// the main-diagonal chain, every bond stitched into an alignment in PATH POSITIONS, the mask extended by the chain (both ways round)
int cl_leaf_bond_alignments(cl_context* ctx, const cl_base_graph* leaf, const cl_leaf_calibration* memo, const cl_anchor_params* ap,
                            const cl_stitch_params* sp, const cl_bond_params* bp, uint64_t max_rounds, cl_alignment_list* out) {
    cl_bind_device(ctx);
    if (!ctx || !leaf || !memo || !ap || !sp || !bp || !out) return CL_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    if (leaf->n_paths != 1) { cl_set_error(ctx, "cl_leaf_bond_alignments: a leaf has one path"); return CL_ERR_INVALID_ARGUMENT; }
    std::vector<uint8_t> lab(leaf->label, leaf->label + leaf->n_nodes);
    lab[leaf->src_id] = 5; lab[leaf->snk_id] = 6;
    cl_base_graph g = *leaf;
    g.label = lab.data();
    const uint32_t* path = leaf->path_nodes + leaf->path_off[0];
    const uint64_t path_len = leaf->path_off[1] - leaf->path_off[0];
    std::vector<uint64_t> first_step(leaf->n_nodes, kGap);   // StepIndex::path_steps(node).front().second
    for (uint64_t i = path_len; i-- > 0;) first_step[path[i]] = i;
    const cl_anchor_chain_result& oc = memo->chain;
    const cl_chain_anchors opt{oc.n_anchors, oc.walk_off, oc.walk1, oc.walk2, oc.score, oc.gap_after, oc.gap_score_after};

    std::unique_ptr<cl_owned_match_sets> own;          // the sets in their current order (the first round reads the calibration's)
    cl_match_sets cur;
    cl_owned_match_sets_view(memo->matches, &cur);
    uint64_t* mask = nullptr;
    uint64_t n_mask = 0;
    int rc = cl_generate_diagonal_mask(&cur, &mask, &n_mask);
    if (rc) return rc;
    std::vector<cl_alignment> alns;
    auto fail = [&](int code) { free(mask); for (auto& a : alns) free(a.pairs); return code; };
    for (uint64_t iter = 0; iter < max_rounds; ++iter) {
        cl_anchor_chain_result sec;
        if ((rc = cl_anchor_chain_masked(ctx, &g, &g, &cur, ap, mask, n_mask, &memo->scale, &sec))) return fail(rc);
        const cl_chain_anchors secondary{sec.n_anchors, sec.walk_off, sec.walk1, sec.walk2, sec.score, sec.gap_after, sec.gap_score_after};
        cl_bonds bonds;
        rc = cl_identify_bonds(&g, &opt, &secondary, bp, 1, &bonds);
        if (rc) { cl_anchor_chain_result_free(&sec); return fail(rc); }
        // the call has reordered the caller's sets and re-indexed the mask (anchorer.hpp:1131-1168): position k now holds the old set_order[k]
        {
            std::vector<uint64_t> inv(cur.n_sets);
            for (uint64_t k = 0; k < cur.n_sets; ++k) inv[sec.set_order[k]] = k;
            for (uint64_t i = 0; i < n_mask; ++i) mask[3 * i] = inv[mask[3 * i]];
            auto next = reordered_sets(cur, sec.set_order);
            own = std::move(next);
            cl_owned_match_sets_view(own.get(), &cur);
        }
        if (bonds.n_intervals == 0) { cl_bonds_free(&bonds); cl_anchor_chain_result_free(&sec); break; }
        for (uint64_t b = 0; b < bonds.n_intervals && !rc; ++b) {
            // Core::bonds_to_chain (core.hpp:405-424): every bond an anchor of the leaf's path nodes
            std::vector<uint64_t> walk_off{0};
            std::vector<uint32_t> w1, w2;
            for (uint64_t e = bonds.interval_off[b]; e < bonds.interval_off[b + 1]; ++e) {
                for (uint64_t j = 0; j < bonds.length[e]; ++j) { w1.push_back(path[bonds.offset1[e] + j]); w2.push_back(path[bonds.offset2[e] + j]); }
                walk_off.push_back(w1.size());
            }
            cl_alignment aln{};
            rc = cl_internal_stitch(ctx, &g, walk_off.size() - 1, walk_off.data(), w1.data(), w2.data(), sp, &aln);
            if (rc) break;
            for (uint64_t i = 0; i < 2 * aln.n_pairs; ++i) if (aln.pairs[i] != kGap) aln.pairs[i] = first_step[aln.pairs[i]];
            alns.push_back(aln);
        }
        cl_bonds_free(&bonds);
        if (rc) { cl_anchor_chain_result_free(&sec); return fail(rc); }
        uint64_t* grown = nullptr;
        uint64_t n_grown = 0;
        rc = cl_update_mask(&cur, sec.walk_off[sec.n_anchors], sec.walk1, sec.walk2, 1, mask, n_mask, &grown, &n_grown);
        cl_anchor_chain_result_free(&sec);
        if (rc) return fail(rc);
        free(mask);
        mask = grown; n_mask = n_grown;
    }
    free(mask);
    out->n = alns.size();
    out->alignments = (cl_alignment*)malloc((alns.size() ? alns.size() : 1) * sizeof(cl_alignment));
    if (!out->alignments) { for (auto& a : alns) free(a.pairs); return CL_ERR_OUT_OF_MEMORY; }
    for (size_t i = 0; i < alns.size(); ++i) out->alignments[i] = alns[i];
    return CL_OK;
}

int cl_simplify_bubbles(const cl_base_graph* graph, cl_owned_base_graph** out) {
    if (!graph || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (graph->n_nodes == 0 || graph->src_id >= graph->n_nodes || graph->snk_id >= graph->n_nodes) return CL_ERR_INVALID_ARGUMENT;
    std::unique_ptr<cl_owned_base_graph> o(new cl_owned_base_graph(owned_copy(graph)));
    std::string error;
    if (!simplify_bubbles_impl(*o, error)) return CL_ERR_INVALID_ARGUMENT;
    *out = o.release();
    return CL_OK;
}

// Core::apply_bonds up to the polishing step (src/core.cpp:613-645): the bond alignments, in path positions of the named paths, become node
// pairs of the root graph, internal_fuse merges along all of them at once, simplify_bubbles tidies up what the merge left behind
int cl_apply_bonds(const cl_base_graph* root, uint64_t n_alignments, const uint64_t* path_of_alignment, const cl_alignment* alignments,
                   cl_owned_base_graph** out) {
    if (!root || !out || (n_alignments && (!path_of_alignment || !alignments))) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    std::vector<uint64_t> pairs;
    for (uint64_t a = 0; a < n_alignments; ++a) {
        const uint64_t p = path_of_alignment[a];
        if (p >= root->n_paths) return CL_ERR_INVALID_ARGUMENT;
        const uint32_t* path = root->path_nodes + root->path_off[p];
        const uint64_t len = root->path_off[p + 1] - root->path_off[p];
        for (uint64_t i = 0; i < 2 * alignments[a].n_pairs; ++i) {
            const uint64_t x = alignments[a].pairs[i];
            if (x != kGap && x >= len) return CL_ERR_INVALID_ARGUMENT;
            pairs.push_back(x == kGap ? kGap : (uint64_t)path[x]);
        }
    }
    cl_owned_base_graph* fused = nullptr;
    int rc = cl_internal_fuse(root, pairs.data(), pairs.size() / 2, &fused, nullptr);
    if (rc) return rc;
    std::string error;
    if (!simplify_bubbles_impl(*fused, error)) { cl_owned_base_graph_free(fused); return CL_ERR_INVALID_ARGUMENT; }
    *out = fused;
    return CL_OK;
}

}  // extern "C"
