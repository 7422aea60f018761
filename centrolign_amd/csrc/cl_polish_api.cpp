// cl_polish_api.cpp — the polishing step of cyclisation (Core::polish_cyclized_graph, src/core.cpp:650-767):
//
//   cl_identify_inconsistencies   InconsistencyIdentifier::identify_inconsistencies (include/centrolign/inconsistency_identifier.hpp:66-343,
//                                 src/inconsistency_identifier.cpp): tight cycles, indels placed inconsistently across a bond, merged along
//                                 their chains and padded with flanking sequence — node pairs that bound the regions to realign
//
// Host code (graph bookkeeping, as in the reference); the realignments themselves re-enter the hot path through cl_core_align.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstring>
#include <tuple>
#include <deque>
#include <list>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "cl_internal.hpp"
#include "snarl_tree.hpp"

namespace {

using clsnarl::kNone;
using clsnarl::SnarlTree;

struct Steps {   // StepIndex (include/centrolign/step_index.hpp): per node its (path, step) visits in path order
    std::vector<uint64_t> off;
    std::vector<std::pair<uint64_t, uint64_t>> at;
    void build(const cl_base_graph& g) {
        off.assign(g.n_nodes + 1, 0);
        for (uint64_t i = 0; i < g.path_off[g.n_paths]; ++i) ++off[g.path_nodes[i] + 1];
        for (uint64_t v = 0; v < g.n_nodes; ++v) off[v + 1] += off[v];
        at.resize(g.path_off[g.n_paths]);
        std::vector<uint64_t> fill(off.begin(), off.end() - 1);
        for (uint64_t p = 0; p < g.n_paths; ++p)
            for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i) at[fill[g.path_nodes[i]]++] = std::make_pair(p, i - g.path_off[p]);
    }
    const std::pair<uint64_t, uint64_t>* begin(uint64_t v) const { return at.data() + off[v]; }
    const std::pair<uint64_t, uint64_t>* end(uint64_t v) const { return at.data() + off[v + 1]; }
};

using PathPositions = std::map<uint64_t, std::vector<uint64_t>>;   // (the reference's hash maps are only looked up or reduced order-free)

PathPositions path_positions(const Steps& st, uint64_t node) {
    PathPositions pp;
    for (auto it = st.begin(node); it != st.end(node); ++it) pp[it->first].push_back(it->second);
    for (auto& kv : pp) if (!std::is_sorted(kv.second.begin(), kv.second.end())) std::sort(kv.second.begin(), kv.second.end());
    return pp;
}

// InconsistencyIdentifier::identify_tight_cycles (src/inconsistency_identifier.cpp:324-469): snarls whose net graph has a cycle and through
// which no path takes more than max_tight_cycle_size steps, outermost ones only
std::vector<std::pair<uint64_t, uint64_t>> tight_cycles(const SnarlTree& sn, const Steps& st, const cl_polish_params& pp) {
    std::vector<char> chain_blocked(sn.n_chains(), 0), snarl_blocked(sn.n_snarls(), 0);
    std::vector<std::list<uint64_t>> chain_cyc(sn.n_chains()), snarl_cyc(sn.n_snarls());
    for (const auto& f : sn.postorder()) {
        uint64_t start, end;
        if (f.second) {
            if (chain_blocked[f.first]) {
                if (sn.chain_parent[f.first] != kNone) snarl_blocked[sn.chain_parent[f.first]] = 1;
                continue;
            }
            start = sn.boundaries[sn.chain_snarls[f.first].front()].first;
            end = sn.boundaries[sn.chain_snarls[f.first].back()].second;
        } else {
            if (snarl_blocked[f.first]) { chain_blocked[sn.snarl_chain[f.first]] = 1; continue; }
            start = sn.boundaries[f.first].first;
            end = sn.boundaries[f.first].second;
            if (!sn.nontrivial_left_boundary[start]) continue;
        }
        std::map<uint64_t, std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> pos;
        for (auto it = st.begin(start); it != st.end(start); ++it) pos[it->first].first.push_back(it->second);
        for (auto it = st.begin(end); it != st.end(end); ++it) pos[it->first].second.push_back(it->second);
        uint64_t max_size = 0;
        for (auto& kv : pos) {
            std::sort(kv.second.first.begin(), kv.second.first.end());
            std::sort(kv.second.second.begin(), kv.second.second.end());
            for (size_t i = 0; i < kv.second.first.size(); ++i) max_size = std::max<uint64_t>(max_size, kv.second.second[i] - kv.second.first[i]);
        }
        if (max_size > pp.max_tight_cycle_size) {
            if (f.second) { if (sn.chain_parent[f.first] != kNone) snarl_blocked[sn.chain_parent[f.first]] = 1; }
            else chain_blocked[sn.snarl_chain[f.first]] = 1;
        } else if (!f.second && !sn.net_acyclic[f.first]) {
            snarl_cyc[f.first].clear();
            snarl_cyc[f.first].push_back(f.first);
        }
        if (f.second) {
            if (sn.chain_parent[f.first] != kNone) { auto& up = snarl_cyc[sn.chain_parent[f.first]]; up.splice(up.end(), chain_cyc[f.first]); }
        } else {
            auto& up = chain_cyc[sn.snarl_chain[f.first]];
            up.splice(up.end(), snarl_cyc[f.first]);
        }
    }
    std::vector<std::pair<uint64_t, uint64_t>> out;
    for (const auto* lists : {&chain_cyc, &snarl_cyc})
        for (const auto& l : *lists) for (uint64_t s : l) out.push_back(sn.boundaries[s]);
    return out;
}

// InconsistencyIdentifier::identify_inconsistent_bonds (src/inconsistency_identifier.cpp:17-321): along a chain that some path runs through
// more than once, windows of non-trivial snarls in which two passes of one path carry enough sequence the other does not
std::vector<std::pair<uint64_t, uint64_t>> inconsistent_bonds(const SnarlTree& sn, const Steps& st, const cl_polish_params& pp) {
    auto max_path_distance = [](const PathPositions& l, const PathPositions& r, uint64_t path) {
        uint64_t d = 0;
        const auto& a = l.find(path)->second;
        const auto& b = r.find(path)->second;
        for (size_t i = 0; i < a.size(); ++i) d = std::max<uint64_t>(d, b[i] - a[i] + 1);
        return d;
    };
    auto median_path_distance = [](const PathPositions& l, const PathPositions& r) {
        std::vector<uint64_t> d;
        for (const auto& kv : l) {
            const auto& b = r.find(kv.first)->second;
            for (size_t i = 0; i < kv.second.size(); ++i) d.push_back(b[i] - kv.second[i] + 1);
        }
        // utility.hpp median(): the middle element of the sorted values, the mean of the middle two for an even number
        std::sort(d.begin(), d.end());
        if (d.empty()) return (uint64_t)0;
        return d.size() % 2 ? d[d.size() / 2] : (d[d.size() / 2 - 1] + d[d.size() / 2]) / 2;
    };
    std::vector<std::pair<uint64_t, uint64_t>> out;
    std::deque<std::pair<uint64_t, bool>> queue;
    for (uint64_t c = 0; c < sn.n_chains(); ++c) if (sn.chain_parent[c] == kNone) queue.emplace_back(c, true);
    while (!queue.empty()) {
        const auto f = queue.front();
        queue.pop_front();
        if (!f.second) { for (uint64_t c : sn.snarl_children[f.first]) queue.emplace_back(c, true); continue; }
        const auto& chain = sn.chain_snarls[f.first];
        std::vector<size_t> nontrivial;
        for (size_t i = 0; i < chain.size(); ++i) if (sn.nontrivial_left_boundary[sn.boundaries[chain[i]].first]) nontrivial.push_back(i);
        if (nontrivial.empty()) continue;
        PathPositions multipass = path_positions(st, sn.boundaries[chain.front()].first);
        for (auto it = multipass.begin(); it != multipass.end();) { if (it->second.size() == 1) it = multipass.erase(it); else ++it; }
        std::vector<char> used(nontrivial.size(), 0);
        if (!multipass.empty()) {
            for (const auto& kv : path_positions(st, sn.boundaries[chain.back()].second)) {
                auto it = multipass.find(kv.first);
                if (it == multipass.end()) continue;
                for (uint64_t p : kv.second) it->second.push_back(p);
                std::sort(it->second.begin(), it->second.end());
            }
            auto identify_pass = [&](uint64_t path, uint64_t pos) -> uint64_t {
                auto it = multipass.find(path);
                if (it == multipass.end()) return kNone;
                return (uint64_t)(std::upper_bound(it->second.begin(), it->second.end(), pos) - it->second.begin()) / 2;
            };
            using PassSetLengths = std::map<uint64_t, std::map<std::vector<bool>, uint64_t>>;
            std::vector<PassSetLengths> per_snarl(nontrivial.size());
            for (size_t i = 0; i < nontrivial.size(); ++i) {
                for (uint64_t child : sn.snarl_children[chain[nontrivial[i]]]) {
                    const auto& cc = sn.chain_snarls[child];
                    const PathPositions left = path_positions(st, sn.boundaries[cc.front()].first), right = path_positions(st, sn.boundaries[cc.back()].second);
                    for (const auto& pass : multipass) {
                        uint64_t length;
                        std::vector<bool> which(pass.second.size() / 2, false);
                        auto it = left.find(pass.first);
                        if (it == left.end()) length = median_path_distance(left, right);
                        else {
                            length = max_path_distance(left, right, pass.first);
                            for (uint64_t pos : it->second) which[identify_pass(it->first, pos)] = true;
                        }
                        per_snarl[i][pass.first][std::move(which)] += length;
                    }
                }
            }
            std::vector<std::pair<size_t, PassSetLengths>> windows(nontrivial.size());
            for (size_t i = 0; i < windows.size(); ++i) windows[i] = std::make_pair(i, per_snarl[i]);
            size_t steps = 1;
            while (!windows.empty()) {
                decltype(windows) next_windows;
                for (size_t i = windows.size(); i-- > 0;) {
                    auto window = std::move(windows[i]);
                    bool suspicious = false;
                    for (auto it = window.second.begin(); it != window.second.end() && !suspicious; ++it) {
                        if (it->second.empty()) continue;
                        const size_t n_pass = it->second.begin()->first.size();
                        for (size_t p1 = 0; p1 < n_pass && !suspicious; ++p1)
                            for (size_t p2 = p1 + 1; p2 < n_pass; ++p2) {
                                uint64_t dj1 = 0, dj2 = 0, neither = 0;
                                for (const auto& rec : it->second) {
                                    if (rec.first[p1] && !rec.first[p2]) dj1 += rec.second;
                                    else if (!rec.first[p1] && rec.first[p2]) dj2 += rec.second;
                                    else if (!rec.first[p1] && !rec.first[p2]) neither += rec.second;
                                }
                                if (dj1 >= pp.min_inconsistency_disjoint_length && dj2 >= pp.min_inconsistency_disjoint_length &&
                                    (dj1 + dj2) / 2 + neither >= pp.min_inconsistency_total_length) { suspicious = true; break; }
                            }
                    }
                    if (suspicious) {
                        out.emplace_back(sn.boundaries[chain[nontrivial[window.first]]].first, sn.boundaries[chain[nontrivial[window.first + steps - 1]]].second);
                        for (size_t j = window.first; j < window.first + steps; ++j) used[j] = 1;
                    } else if (window.first + steps < nontrivial.size() && !used[window.first + steps] &&
                               nontrivial[window.first + steps] - nontrivial[window.first] < pp.max_bond_inconsistency_window) {
                        for (const auto& from : per_snarl[window.first + steps]) {
                            auto& into = window.second[from.first];
                            for (const auto& rec : from.second) into[rec.first] += rec.second;
                        }
                        next_windows.push_back(std::move(window));
                    }
                }
                ++steps;
                windows = std::move(next_windows);
            }
        }
        for (size_t i = 0; i < nontrivial.size(); ++i) if (!used[i]) queue.emplace_back(chain[nontrivial[i]], false);
    }
    return out;
}

// InconsistencyIdentifier::expand_inconsistencies (inconsistency_identifier.hpp:189-343): every region grows along its chain, least-grown side
// first, while the snarl beside it adds at most padding_target_min_length of shortest walk and padding_max_length_limit of longest walk and
// does not touch another region
void expand(std::vector<std::pair<uint64_t, uint64_t>>& inc, const SnarlTree& sn, const cl_polish_params& pp) {
    struct Frame {
        uint64_t left_min = 0, right_min = 0, left_max = 0, right_max = 0;
        uint64_t can_left = 1, can_right = 1;
        uint64_t left = kNone, right = kNone, origin = kNone;
        std::pair<uint64_t, bool> frontier() const {
            if ((can_left && left_min < right_min) || !can_right) return std::make_pair(left_min, true);
            return std::make_pair(right_min, false);
        }
        bool operator<(const Frame& o) const { return frontier() > o.frontier(); }
    };
    std::vector<Frame> heap;
    std::unordered_set<uint64_t> is_boundary;
    for (size_t i = 0; i < inc.size(); ++i) {
        is_boundary.insert(inc[i].first);
        is_boundary.insert(inc[i].second);
        Frame f;
        f.left = inc[i].first; f.right = inc[i].second; f.origin = i;
        heap.push_back(f);
    }
    std::make_heap(heap.begin(), heap.end());
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end());
        Frame& nx = heap.back();
        const bool go_left = nx.frontier().second;
        uint64_t& boundary = go_left ? nx.left : nx.right;
        uint64_t& can = go_left ? nx.can_left : nx.can_right;
        uint64_t& grown_min = go_left ? nx.left_min : nx.right_min;
        uint64_t& grown_max = go_left ? nx.left_max : nx.right_max;
        const uint64_t beside = go_left ? sn.ends_at[boundary] : sn.begins_at[boundary];
        if (beside == kNone) can = 0;
        else {
            const uint64_t next_boundary = go_left ? sn.boundaries[beside].first : sn.boundaries[beside].second;
            if (is_boundary.count(next_boundary)) can = 0;
            else {
                const auto d = sn.snarl_dist[beside];
                if (d.second == kNone) can = 0;
                else {
                    const uint64_t lo = grown_min + d.first - 1, hi = grown_max + d.second - 1;
                    if (lo > pp.padding_target_min_length || hi > pp.padding_max_length_limit) can = 0;
                    else {
                        grown_min = lo; grown_max = hi;
                        is_boundary.erase(boundary);
                        boundary = next_boundary;
                        is_boundary.insert(next_boundary);
                    }
                }
            }
        }
        if (!nx.can_left && !nx.can_right) {
            inc[nx.origin] = std::make_pair(nx.left, nx.right);
            heap.pop_back();
        } else {
            std::push_heap(heap.begin(), heap.end());
        }
    }
}

std::vector<std::pair<uint64_t, uint64_t>> identify_inconsistencies(const cl_base_graph& g, const cl_polish_params& pp, bool& ok) {
    clsnarl::GraphView view{g.n_nodes, g.next_off, g.next_idx, g.prev_off, g.prev_idx, g.src_id, g.snk_id};
    SnarlTree sn;
    ok = sn.build(view);
    if (!ok) return {};
    Steps st;
    st.build(g);
    const auto cycles = tight_cycles(sn, st, pp);
    const auto bonds = inconsistent_bonds(sn, st, pp);
    std::vector<uint64_t> pos_in_chain(sn.n_snarls(), 0);
    for (uint64_t c = 0; c < sn.n_chains(); ++c) for (size_t i = 0; i < sn.chain_snarls[c].size(); ++i) pos_in_chain[sn.chain_snarls[c][i]] = i;
    // per snarl the furthest snarl of its chain up to which an inconsistency that starts here reaches
    std::vector<uint64_t> reach(sn.n_snarls(), kNone);
    for (const auto& t : cycles) reach[sn.begins_at[t.first]] = sn.ends_at[t.second];
    for (const auto& b : bonds) {
        const uint64_t s = sn.begins_at[b.first], other = sn.ends_at[b.second];
        if (reach[s] == kNone || pos_in_chain[reach[s]] < pos_in_chain[other]) reach[s] = other;
    }
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    std::deque<std::pair<uint64_t, bool>> queue;
    for (uint64_t c = 0; c < sn.n_chains(); ++c) if (sn.chain_parent[c] == kNone) queue.emplace_back(c, true);
    while (!queue.empty()) {
        const auto f = queue.front();
        queue.pop_front();
        if (f.second) {
            const auto& chain = sn.chain_snarls[f.first];
            for (size_t i = 0; i < chain.size(); ++i) {
                if (reach[chain[i]] != kNone) {
                    if (!merged.empty() && merged.back().second == sn.boundaries[chain[i]].first) merged.back().second = sn.boundaries[reach[chain[i]]].second;
                    else merged.emplace_back(sn.boundaries[chain[i]].first, sn.boundaries[reach[chain[i]]].second);
                    while (chain[i] != sn.ends_at[merged.back().second]) ++i;
                } else {
                    queue.emplace_back(chain[i], false);
                }
            }
        } else {
            for (uint64_t c : sn.snarl_children[f.first]) queue.emplace_back(c, true);
        }
    }
    expand(merged, sn, pp);
    return merged;
}


// ---- InducedMatchFinder (include/centrolign/induced_match_finder.hpp, src/induced_match_finder.cpp): the full graph's matches against
//      itself, localised to the regions that are realigned: per region the match sets that touch it, per set where its walks lie on every
//      path and how often it occurs in the whole graph
struct PathHitSet {
    std::map<uint64_t, std::vector<std::pair<uint64_t, uint64_t>>> hit_locations;   // path -> (offset, index of the walk in its set), ascending
    uint64_t length = 0, deduplicated_count = 0;
};

std::vector<std::vector<PathHitSet>> induce_matches(const cl_base_graph& g, const cl_match_sets& ms, const std::vector<std::pair<uint64_t, uint64_t>>& regions,
                                                    const Steps& st) {
    std::vector<std::vector<PathHitSet>> hits(regions.size());
    std::vector<uint64_t> region_of(g.n_nodes, kNone);
    for (size_t i = 0; i < regions.size(); ++i) {
        std::vector<uint64_t> stack(1, regions[i].first);
        region_of[regions[i].first] = i;
        region_of[regions[i].second] = i;
        while (!stack.empty()) {
            const uint64_t v = stack.back();
            stack.pop_back();
            for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) {
                const uint64_t w = g.next_idx[e];
                if (region_of[w] == kNone) { region_of[w] = i; stack.push_back(w); }
            }
        }
    }
    for (uint64_t s = 0; s < ms.n_sets; ++s) {
        std::vector<uint64_t> started;   // regions whose hit set for this match set exists
        const uint64_t n_walks = ms.set_off1[s + 1] - ms.set_off1[s];
        for (uint64_t j = 0; j < n_walks; ++j) {
            const uint64_t w = ms.set_off1[s] + j;
            const uint32_t* walk = ms.nodes1 + ms.walk_off1[w];
            const uint64_t len = ms.walk_off1[w + 1] - ms.walk_off1[w];
            std::vector<uint64_t> touched;
            for (uint64_t k = 0; k < len; ++k) if (region_of[walk[k]] != kNone) touched.push_back(region_of[walk[k]]);
            std::sort(touched.begin(), touched.end());
            touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
            if (touched.empty()) continue;
            for (uint64_t r : touched)
                if (std::find(started.begin(), started.end(), r) == started.end()) {
                    started.push_back(r);
                    hits[r].emplace_back();
                    hits[r].back().length = len;
                    hits[r].back().deduplicated_count = n_walks;
                }
            // the paths that spell the whole walk: (path, step) pairs extended node by node
            std::vector<std::pair<uint64_t, uint64_t>> ext(st.begin(walk[0]), st.end(walk[0]));
            for (uint64_t k = 1; k < len && !ext.empty(); ++k) {
                std::vector<std::pair<uint64_t, uint64_t>> next;
                for (auto it = st.begin(walk[k]); it != st.end(walk[k]); ++it)
                    if (it->second > 0 && std::binary_search(ext.begin(), ext.end(), std::make_pair(it->first, it->second - 1))) next.push_back(*it);
                ext.swap(next);   // ((path, step) lists of a node are ascending, so every list stays sorted)
            }
            for (const auto& e : ext)
                for (uint64_t r : touched) hits[r].back().hit_locations[e.first].emplace_back(e.second + 1 - len, j);
        }
        for (uint64_t r : started) {
            auto& loc = hits[r].back().hit_locations;
            if (loc.empty() || (loc.size() == 1 && loc.begin()->second.size() == 1)) hits[r].pop_back();
            else for (auto& kv : loc) std::sort(kv.second.begin(), kv.second.end());
        }
    }
    return hits;
}

// a subproblem graph of a realignment: its paths are stretches [begin, end] of the full graph's paths
struct SubPaths { std::vector<std::tuple<uint64_t, uint64_t, uint64_t>> of; };   // per path (full-graph path, begin, end)

// InducedMatchFinderComponentView::find_matches (induced_match_finder.hpp:100-372) as owned match sets
// *past_the_paths: a hit that starts ONE step behind a stretch is taken for the stretch as well (the upper bound (path_end + 1, 0) lets a location of walk 0 at
// path_end + 1 through, induced_match_finder.hpp:190): its clipped end is path_end - match_begin = -1 as a size_t, and it is read from the step BEHIND the subproblem's
// path — in the reference past the end of that path's own vector (whatever the heap holds; the runs scripts/fuzz_msa.py met ended in a segmentation fault).  Nothing can
// be "the same" there: this function stops and says so instead of reading on into the next path's nodes
std::unique_ptr<cl_owned_match_sets> induced_find_matches(const cl_base_graph& full, const std::vector<PathHitSet>& path_hits, const cl_base_graph& g1,
                                                          const SubPaths& sp1, const cl_base_graph& g2, const SubPaths& sp2, bool* past_the_paths) {
    std::unique_ptr<cl_owned_match_sets> out(new cl_owned_match_sets());
    *past_the_paths = false;
    std::unordered_set<uint64_t> parent_seen;
    uint64_t len1 = 0, len2 = 0;
    for (int side = 0; side < 2; ++side) {
        const SubPaths& sp = side ? sp2 : sp1;
        uint64_t& len = side ? len2 : len1;
        for (const auto& t : sp.of)
            if (parent_seen.insert(std::get<0>(t)).second) len += full.path_off[std::get<0>(t) + 1] - full.path_off[std::get<0>(t)];
    }
    const double ratio = double(len1) / double(len2);
    // double -> size_t as the reference's build converts it (x86-64, gcc: cvttsd2si below 2^63, else of x - 2^63 with the top bit flipped): when
    // both graphs hold stretches of the SAME paths only, the second graph's path length is 0 (a parent path is counted once, for the graph that
    // shows it first), the ratio is infinite, and infinity — not representable — converts to 0; the counts then fall back to what was observed
    auto to_size = [](double x) -> uint64_t {
        const double two63 = 9223372036854775808.0;
        if (x < two63) return x >= -two63 ? (uint64_t)(int64_t)x : 0x8000000000000000ull;
        const double y = x - two63;
        return (y < two63 ? (uint64_t)(int64_t)y : 0x8000000000000000ull) ^ 0x8000000000000000ull;
    };
    auto assign_count = [&to_size](uint64_t observed1, uint64_t observed2, uint64_t target, double ratio12) {
        uint64_t count2 = to_size(round(sqrt(target / ratio12)));
        uint64_t count1 = to_size(round(sqrt(target * ratio12)));
        if (count1 >= observed1 && count2 < observed2) { count2 = observed2; count1 = to_size(round(target / double(count2))); }
        else if (count2 >= observed2 && count1 < observed1) { count1 = observed1; count2 = to_size(round(target / double(count1))); }
        return std::make_pair(std::max(count1, observed1), std::max(count2, observed2));
    };
    for (const PathHitSet& hs : path_hits) {
        std::unordered_set<uint64_t> origin_walks;
        std::vector<std::tuple<uint64_t, uint64_t, bool, uint64_t, uint64_t>> iv;   // (match begin, match end, on graph 1, path, offset on the path)
        uint64_t observed1 = 0, observed2 = 0;
        for (int side = 0; side < 2; ++side) {
            const bool do1 = side == 0;
            const cl_base_graph& g = do1 ? g1 : g2;
            const SubPaths& sp = do1 ? sp1 : sp2;
            uint64_t& observed = do1 ? observed1 : observed2;
            std::set<std::pair<uint64_t, uint64_t>> initial;
            for (uint64_t p = 0; p < g.n_paths; ++p) {
                const uint64_t parent = std::get<0>(sp.of[p]), pb = std::get<1>(sp.of[p]), pe = std::get<2>(sp.of[p]);
                auto it = hs.hit_locations.find(parent);
                if (it == hs.hit_locations.end()) continue;
                auto lo = std::lower_bound(it->second.begin(), it->second.end(), std::make_pair(pb >= hs.length ? pb - hs.length : (uint64_t)0, (uint64_t)0));
                auto hi = std::upper_bound(it->second.begin(), it->second.end(), std::make_pair(pe + 1, (uint64_t)0));
                for (auto l = lo; l != hi; ++l) {
                    const uint64_t mb = l->first, me = mb + hs.length;
                    const uint64_t offset = mb < pb ? 0 : mb - pb;
                    if (offset >= g.path_off[p + 1] - g.path_off[p]) {
                        // CL_POLISH_SKIP_STRAY_HITS=1: what the bound evidently means — a hit that begins behind the stretch is none of the stretch's — instead of the error: for
                        // inputs the reference dies on, so nothing it prints can be compared with
                        static const bool skip = [] { const char* e = getenv("CL_POLISH_SKIP_STRAY_HITS"); return e && e[0] == '1'; }();
                        if (skip) continue;
                        *past_the_paths = true;
                        return out;
                    }
                    origin_walks.insert(l->second);
                    const uint64_t begin = mb < pb ? pb - mb : 0, end = me > pe ? pe - mb : hs.length;
                    const uint64_t node = g.path_nodes[g.path_off[p] + offset];
                    if (initial.emplace(node, begin).second) iv.emplace_back(begin, end, do1, p, offset);
                    ++observed;
                }
            }
        }
        const uint64_t total = observed1 * observed2 + hs.deduplicated_count - origin_walks.size();
        const auto counts = assign_count(observed1, observed2, total, ratio);
        std::sort(iv.begin(), iv.end());
        std::vector<size_t> active;   // a heap by interval end (std::push_heap / pop_heap: the order of the walks of a set is the heap's)
        auto cmp = [&](size_t a, size_t b) { return std::get<1>(iv[a]) > std::get<1>(iv[b]); };
        uint64_t last = 0, n1 = 0, n2 = 0;
        size_t i = 0;
        while (i < iv.size() || !active.empty()) {
            bool is_start;
            uint64_t next;
            if (active.empty() || (i < iv.size() && std::get<0>(iv[i]) < std::get<1>(iv[active.front()]))) { is_start = true; next = std::get<0>(iv[i]); }
            else { is_start = false; next = std::get<1>(iv[active.front()]); }
            if (n1 != 0 && n2 != 0 && next != last) {
                for (int side = 0; side < 2; ++side) {
                    auto& set_off = side ? out->set_off2 : out->set_off1;
                    auto& walk_off = side ? out->walk_off2 : out->walk_off1;
                    auto& nodes = side ? out->nodes2 : out->nodes1;
                    const cl_base_graph& g = side ? g2 : g1;
                    for (size_t idx : active) {
                        if (std::get<2>(iv[idx]) != (side == 0)) continue;
                        const uint64_t b = std::get<4>(iv[idx]) + (last - std::get<0>(iv[idx])), e = b + (next - last);
                        const uint32_t* path = g.path_nodes + g.path_off[std::get<3>(iv[idx])];
                        if (e < b || e > g.path_off[std::get<3>(iv[idx]) + 1] - g.path_off[std::get<3>(iv[idx])]) { *past_the_paths = true; return out; }
                        nodes.insert(nodes.end(), path + b, path + e);
                        walk_off.push_back(nodes.size());
                    }
                    set_off.push_back(walk_off.size() - 1);
                }
                out->full_length.push_back(hs.length);
                out->count1.push_back(counts.first);
                out->count2.push_back(counts.second);
            }
            last = next;
            if (is_start) {
                size_t j = i + 1;
                while (j < iv.size() && std::get<0>(iv[j]) == std::get<0>(iv[i])) ++j;
                for (size_t k = i; k < j; ++k) {
                    active.push_back(k);
                    if (std::get<2>(iv[k])) ++n1; else ++n2;
                    std::push_heap(active.begin(), active.end(), cmp);
                }
                i = j;
            } else {
                auto heap_end = active.end();
                std::pop_heap(active.begin(), heap_end--, cmp);
                while (heap_end != active.begin() && std::get<1>(iv[active.front()]) == std::get<1>(iv[active.back()])) std::pop_heap(active.begin(), heap_end--, cmp);
                for (auto it = heap_end; it != active.end(); ++it) { if (std::get<2>(iv[*it])) --n1; else --n2; }
                active.resize(heap_end - active.begin());
            }
        }
    }
    return out;
}


// Core::make_copy_expanded_tree (src/core.cpp:769-976) up to the Newick text it builds (distances left out: nothing downstream reads them):
// the guide tree with every leaf replaced by its copies in this region; the highest subtrees whose observed leaves all have the same number
// of copies are repeated once per copy under a node of their own, so that corresponding copies are aligned with one another first
bool expanded_newick(const ClGuideTreeView& tree, const std::vector<std::tuple<uint64_t, uint64_t, uint64_t>>& intervals,
                     const std::vector<std::string>& subpath_names, const std::vector<std::string>& subpath_parent, std::string& out, std::string& error) {
    const uint64_t U2 = ~(uint64_t)0 - 1, U1 = ~(uint64_t)0;   // the reference's -2 "unobserved" and -1 "inconsistent"
    std::unordered_map<std::string, std::vector<std::string>> copies;
    {
        std::vector<size_t> idx(intervals.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return intervals[a] < intervals[b]; });
        for (size_t i : idx) copies[subpath_parent[i]].push_back(subpath_names[i]);
    }
    std::unordered_map<std::string, uint32_t> id_of;
    for (uint32_t v = 0; v < tree.label.size(); ++v) if (!tree.label[v].empty()) id_of[tree.label[v]] = v;
    std::vector<uint64_t> count(tree.kids.size(), 0);
    for (const auto& c : copies) {
        auto it = id_of.find(c.first);
        if (it == id_of.end()) { error = "path " + c.first + " is not in the guide tree"; return false; }
        count[it->second] = c.second.size();
    }
    for (uint32_t v : tree.postorder) {
        if (tree.kids[v].empty()) continue;
        uint64_t last = U2;
        for (uint32_t c : tree.kids[v]) {
            if (count[c] == U1 || (last != U2 && count[c] != last)) { last = U1; break; }
            if (count[c] != 0) last = count[c];
        }
        if (last != U2) count[v] = last;
    }
    if (count[tree.root] == 0) { error = "Root is not included in induced subpath tree"; return false; }
    struct Rec { uint64_t node, copy; std::vector<std::pair<uint64_t, uint64_t>> edges; size_t next = 0; };
    std::vector<Rec> stack(1);
    if (count[tree.root] == U1) {
        stack.back().node = tree.root; stack.back().copy = U1;
        for (uint32_t c : tree.kids[tree.root]) if (count[c] != 0) stack.back().edges.emplace_back(c, U1);
    } else {
        stack.back().node = U1; stack.back().copy = U1;
        for (uint64_t i = 0; i < count[tree.root]; ++i) stack.back().edges.emplace_back(tree.root, i);
    }
    while (!stack.empty()) {
        if (stack.back().next == stack.back().edges.size()) {
            const Rec& top = stack.back();
            if (!top.edges.empty()) out += ')';
            if (top.node != U1 && tree.kids[top.node].empty()) {
                if (top.copy == U1) { error = "Leaf of induced subpath tree was not marked as having consistent count"; return false; }
                out += '"' + copies.at(tree.label[top.node])[top.copy] + '"';
            }
            stack.pop_back();
            continue;
        }
        out += stack.back().next == 0 ? '(' : ',';
        const auto edge = stack.back().edges[stack.back().next++];
        Rec rec;
        if (edge.second == U1 && count[edge.first] != U1) {   // the first subtree with a consistent count on the way down: a node to house its copies
            rec.node = U1; rec.copy = U1;
            for (uint64_t i = 0; i < count[edge.first]; ++i) rec.edges.emplace_back(edge.first, i);
        } else {
            rec.node = edge.first; rec.copy = edge.second;
            for (uint32_t c : tree.kids[edge.first]) if (count[c] != 0) rec.edges.emplace_back(c, edge.second);
        }
        stack.push_back(std::move(rec));
    }
    out += ';';
    return true;
}

// CL_POLISH_DEBUG=<file>: what every realignment was given and what it made, in the container format of oracle/ref_driver.cpp's dumps
// (name, type, count, data), for comparisons against the reference's recorded flow; development aid only
struct DebugDump {
    FILE* f = nullptr;
    DebugDump() { const char* p = getenv("CL_POLISH_DEBUG"); if (p && *p) { f = fopen(p, "wb"); if (f) fwrite("CLDUMP1\n", 1, 8, f); } }
    ~DebugDump() { if (f) fclose(f); }
    void put(const std::string& name, uint8_t dtype, const void* data, uint64_t count, size_t esz) {
        if (!f) return;
        const uint32_t nl = (uint32_t)name.size();
        fwrite(&nl, 4, 1, f); fwrite(name.data(), 1, nl, f); fwrite(&dtype, 1, 1, f); fwrite(&count, 8, 1, f);
        if (count) fwrite(data, esz, count, f);
    }
    void str(const std::string& n, const std::string& v) { put(n, 0, v.data(), v.size(), 1); }
    void u64(const std::string& n, const std::vector<uint64_t>& v) { put(n, 2, v.data(), v.size(), 8); }
};

struct MutableGraph {   // BaseGraph with add_node / add_edge appending (src/graph.cpp:214-229)
    std::vector<uint8_t> label;
    std::vector<std::vector<uint32_t>> next, prev;
    std::vector<uint64_t> path_off;
    std::vector<uint32_t> path_nodes;
    uint64_t src = 0, snk = 0;
    explicit MutableGraph(const cl_base_graph& g) : label(g.label, g.label + g.n_nodes), next(g.n_nodes), prev(g.n_nodes),
        path_off(g.path_off, g.path_off + g.n_paths + 1), path_nodes(g.path_nodes, g.path_nodes + g.path_off[g.n_paths]), src(g.src_id), snk(g.snk_id) {
        for (uint64_t v = 0; v < g.n_nodes; ++v) {
            next[v].assign(g.next_idx + g.next_off[v], g.next_idx + g.next_off[v + 1]);
            prev[v].assign(g.prev_idx + g.prev_off[v], g.prev_idx + g.prev_off[v + 1]);
        }
    }
    uint32_t add_node(uint8_t l) { label.push_back(l); next.emplace_back(); prev.emplace_back(); return (uint32_t)label.size() - 1; }
    void add_edge(uint32_t a, uint32_t b) { next[a].push_back(b); prev[b].push_back(a); }
    cl_owned_base_graph* owned() const {
        std::unique_ptr<cl_owned_base_graph> o(new cl_owned_base_graph());
        o->label = label;
        o->next_off.assign(1, 0); o->prev_off.assign(1, 0);
        for (size_t v = 0; v < label.size(); ++v) {
            o->next_idx.insert(o->next_idx.end(), next[v].begin(), next[v].end()); o->next_off.push_back(o->next_idx.size());
            o->prev_idx.insert(o->prev_idx.end(), prev[v].begin(), prev[v].end()); o->prev_off.push_back(o->prev_idx.size());
        }
        o->path_off = path_off;
        o->path_nodes = path_nodes;
        o->src_id = src; o->snk_id = snk;
        return o.release();
    }
};

struct Realigned { cl_owned_base_graph* graph = nullptr; SubPaths paths; };

// Core::integrate_polished_subgraphs (src/core.cpp:978-1069) without the final purge
void integrate(MutableGraph& root, const std::vector<Realigned>& realigned) {
    for (const Realigned& r : realigned) {
        cl_base_graph g;
        cl_owned_base_graph_view(r.graph, &g);
        std::vector<uint32_t> trans(g.n_nodes, ~0u);
        for (uint64_t v = 0; v < g.n_nodes; ++v) if (v != g.src_id && v != g.snk_id) trans[v] = root.add_node(g.label[v]);
        for (uint64_t v = 0; v < g.n_nodes; ++v) {
            if (v == g.src_id || v == g.snk_id) continue;
            for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e) {
                const uint64_t w = g.next_idx[e];
                if (w != g.src_id && w != g.snk_id) root.add_edge(trans[v], trans[w]);
            }
        }
        std::set<std::pair<uint32_t, uint32_t>> adjacencies;
        for (uint64_t p = 0; p < g.n_paths; ++p) {
            const uint64_t parent = std::get<0>(r.paths.of[p]), begin = std::get<1>(r.paths.of[p]), end = std::get<2>(r.paths.of[p]);
            if (begin == end) continue;
            uint32_t* rp = root.path_nodes.data() + root.path_off[parent];
            const uint64_t len = root.path_off[parent + 1] - root.path_off[parent];
            const uint32_t before = begin == 0 ? (uint32_t)root.src : rp[begin - 1], after = end + 1 == len ? (uint32_t)root.snk : rp[end + 1];
            const uint32_t* sub = g.path_nodes + g.path_off[p];
            const uint64_t sub_len = g.path_off[p + 1] - g.path_off[p];
            if (adjacencies.emplace(before, trans[sub[0]]).second) root.add_edge(before, trans[sub[0]]);
            if (adjacencies.emplace(trans[sub[sub_len - 1]], after).second) root.add_edge(trans[sub[sub_len - 1]], after);
            for (uint64_t i = 0; i < sub_len; ++i) rp[begin + i] = trans[sub[i]];
        }
    }
}

}  // namespace

extern "C" {

void cl_polish_params_default(cl_polish_params* p) {   // src/parameters.cpp:98-103
    if (!p) return;
    p->max_tight_cycle_size = 10000;
    p->max_bond_inconsistency_window = 100;
    p->min_inconsistency_disjoint_length = 8;
    p->min_inconsistency_total_length = 50;
    p->padding_target_min_length = 1000;
    p->padding_max_length_limit = 10000;
}

int cl_identify_inconsistencies(const cl_base_graph* graph, const cl_polish_params* params, uint64_t** bounds_out, uint64_t* n_out) {
    if (!graph || !params || !bounds_out || !n_out) return CL_ERR_INVALID_ARGUMENT;
    *bounds_out = nullptr;
    *n_out = 0;
    if (graph->n_nodes == 0 || graph->src_id >= graph->n_nodes || graph->snk_id >= graph->n_nodes) return CL_ERR_INVALID_ARGUMENT;
    bool ok = true;
    const auto inc = identify_inconsistencies(*graph, *params, ok);
    if (!ok) return CL_ERR_INVALID_ARGUMENT;
    *bounds_out = (uint64_t*)malloc((inc.size() ? inc.size() : 1) * 2 * sizeof(uint64_t));
    if (!*bounds_out) return CL_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < inc.size(); ++i) { (*bounds_out)[2 * i] = inc[i].first; (*bounds_out)[2 * i + 1] = inc[i].second; }
    *n_out = inc.size();
    return CL_OK;
}

}  // extern "C"

extern "C" {

// Core::polish_cyclized_graph (src/core.cpp:650-767): the regions InconsistencyIdentifier reports are cut out path by path, every stretch a
// sequence of its own, realigned from scratch over the guide tree expanded by their copies — match sets induced from the whole graph's
// matches against itself, Core::align with score_boundaries (core.hpp:235-238) and fuse per tree node: the hot path again — and the
// realigned subgraphs put back in place of the old nodes (integrate_polished_subgraphs + purge_uncovered_nodes)
int cl_polish_cyclized_graph(cl_context* ctx, const cl_base_graph* graph, const char* const* path_names, const char* newick, const char* const* sequence_names,
                             uint64_t n_sequences, const cl_merge_params* mp, const cl_polish_params* pp, cl_owned_base_graph** out, uint64_t* n_regions_out) {
    cl_bind_device(ctx);
    return cl_polish_cyclized_graph_workers(&ctx, 1, graph, path_names, newick, sequence_names, n_sequences, mp, pp, out, n_regions_out);
}

}  // extern "C"

// cl_polish_cyclized_graph with worker contexts (cl_msa hands over the contexts its merges ran on): the regions are realigned independently of
// one another — the reference takes them one after the other, src/core.cpp:650-767 — so they are handed out to one thread per context; the
// realigned subgraphs go back into the graph in region order afterwards, as in the reference
int cl_polish_cyclized_graph_workers(cl_context* const* ctxs, unsigned n_ctx, const cl_base_graph* graph, const char* const* path_names, const char* newick,
                                     const char* const* sequence_names, uint64_t n_sequences, const cl_merge_params* mp, const cl_polish_params* pp,
                                     cl_owned_base_graph** out, uint64_t* n_regions_out) {
    cl_context* ctx = ctxs && n_ctx ? ctxs[0] : nullptr;
    if (!ctx || !graph || !path_names || !sequence_names || !mp || !pp || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (n_regions_out) *n_regions_out = 0;
    bool ok = true;
    const auto regions = identify_inconsistencies(*graph, *pp, ok);
    if (!ok) { cl_set_error(ctx, "cl_polish_cyclized_graph: no snarl decomposition of the graph"); return CL_ERR_INVALID_ARGUMENT; }
    if (n_regions_out) *n_regions_out = regions.size();
    MutableGraph root(*graph);
    if (regions.empty()) { *out = root.owned(); return CL_OK; }
    Steps st;
    st.build(*graph);
    // the whole graph's matches against itself give the counts (reassign_sentinels 5, 6; second copy under 7, 8: src/core.cpp:683-696)
    std::vector<uint8_t> lab1(graph->label, graph->label + graph->n_nodes), lab2(lab1);
    lab1[graph->src_id] = 5; lab1[graph->snk_id] = 6;
    lab2[graph->src_id] = 7; lab2[graph->snk_id] = 8;
    cl_base_graph a = *graph, b = *graph;
    a.label = lab1.data();
    b.label = lab2.data();
    root.label[graph->src_id] = 5; root.label[graph->snk_id] = 6;
    cl_owned_match_sets* full = nullptr;
    int rc = cl_find_matches(ctx, &a, &b, &mp->match, &full, nullptr);
    if (rc) return rc;
    cl_match_sets fv;
    cl_owned_match_sets_view(full, &fv);
    const auto hits = induce_matches(a, fv, regions, st);
    cl_owned_match_sets_free(full);
    ClGuideTreeView tree;
    std::string error;
    if ((rc = cl_processed_guide_tree(newick, sequence_names, n_sequences, tree, error))) { cl_set_error(ctx, "%s", error.c_str()); return rc; }
    std::vector<Realigned> realigned(regions.size());
    auto fail = [&](int code) { for (auto& r : realigned) cl_owned_base_graph_free(r.graph); return code; };
    cl_core_align_params ap = mp->align;
    ap.partition.score_boundaries = 1;
    static const char kDecode[] = "ACGTN";
    DebugDump dbg;
    const bool timing = getenv("CL_POLISH_TIMING") != nullptr;   // (single worker: the sums are not atomic)
    double t_induce = 0, t_align = 0, t_chain = 0, t_part = 0, t_stitch = 0;
    uint64_t n_small_merges = 0;
    auto realign_region = [&](cl_context* ctx, size_t ri) -> int {
        int rc = CL_OK;
        std::string error;
        const std::string rp = "region" + std::to_string(ri) + ".";
        std::map<uint64_t, std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> loc;
        for (auto it = st.begin(regions[ri].first); it != st.end(regions[ri].first); ++it) loc[it->first].first.push_back(it->second);
        for (auto it = st.begin(regions[ri].second); it != st.end(regions[ri].second); ++it) loc[it->first].second.push_back(it->second);
        std::vector<std::tuple<uint64_t, uint64_t, uint64_t>> intervals;
        std::vector<std::string> names, parents, seqs;
        for (auto& kv : loc) {
            if (kv.second.first.size() != kv.second.second.size()) { cl_set_error(ctx, "Path starts or ends in the middle of a cycle realignment interval"); return CL_ERR_INVALID_ARGUMENT; }
            for (size_t k = 0; k < kv.second.first.size(); ++k) {
                const uint64_t bg = kv.second.first[k], en = kv.second.second[k];
                if (getenv("CL_POLISH_DEBUG")) fprintf(stderr, "[polish] region %zu (nodes %llu .. %llu): path %llu occurrence %zu = steps %llu .. %llu\n", ri, (unsigned long long)regions[ri].first, (unsigned long long)regions[ri].second, (unsigned long long)kv.first, k, (unsigned long long)bg, (unsigned long long)en);
                intervals.emplace_back(kv.first, bg, en);
                parents.push_back(path_names[kv.first]);
                names.push_back(parents.back() + ":" + std::to_string(bg) + "-" + std::to_string(en));
                std::string sq;
                for (uint64_t j = bg; j <= en; ++j) {
                    const uint8_t l = graph->label[graph->path_nodes[graph->path_off[kv.first] + j]];
                    sq.push_back(l < 5 ? kDecode[l] : 'N');
                }
                seqs.push_back(std::move(sq));
            }
        }
        std::string text;
        if (!expanded_newick(tree, intervals, names, parents, text, error)) { cl_set_error(ctx, "%s", error.c_str()); return CL_ERR_INVALID_ARGUMENT; }
        if (dbg.f) { std::string all; for (const auto& nm : names) all += nm + "\n"; dbg.str(rp + "names", all); dbg.str(rp + "tree", text); }
        std::vector<const char*> name_ptr;
        for (const auto& nm : names) name_ptr.push_back(nm.c_str());
        cl_msa_plan plan;
        if ((rc = cl_msa_plan_create(ctx, text.c_str(), name_ptr.data(), name_ptr.size(), &plan))) return rc;
        const uint64_t n_slots = plan.n_leaves + plan.n_merges;
        std::vector<cl_owned_base_graph*> slot_graph(n_slots, nullptr);
        std::vector<SubPaths> slot_paths(n_slots);
        auto drop = [&]() { for (auto* g : slot_graph) cl_owned_base_graph_free(g); cl_msa_plan_free(&plan); };
        for (uint64_t i = 0; i < plan.n_leaves && !rc; ++i) {
            const uint64_t sq = plan.leaf_sequence[i];
            rc = cl_leaf_graph(seqs[sq].c_str(), seqs[sq].size(), &slot_graph[i]);
            slot_paths[i].of.assign(1, intervals[sq]);
        }
        for (uint64_t k = 0; k < plan.n_merges && !rc; ++k) {
            const uint64_t ia = plan.merge_children[2 * k], ib = plan.merge_children[2 * k + 1];
            cl_base_graph g1, g2;
            cl_owned_base_graph_view(slot_graph[ia], &g1);
            cl_owned_base_graph_view(slot_graph[ib], &g2);
            std::vector<uint8_t> l1(g1.label, g1.label + g1.n_nodes), l2(g2.label, g2.label + g2.n_nodes);
            l1[g1.src_id] = 5; l1[g1.snk_id] = 6;
            l2[g2.src_id] = 7; l2[g2.snk_id] = 8;
            g1.label = l1.data();
            g2.label = l2.data();
            const auto t_m0 = std::chrono::steady_clock::now();
            bool past_the_paths = false;
            auto ms = induced_find_matches(a, hits[ri], g1, slot_paths[ia], g2, slot_paths[ib], &past_the_paths);
            if (past_the_paths) {
                cl_set_error(ctx, "polishing region %zu: a match that starts one step behind a realigned stretch is read past the end of the subproblem's paths "
                                  "(include/centrolign/induced_match_finder.hpp:190-205 reads past its path's vector there; the reference ends in a segmentation fault on such an input)", ri);
                rc = CL_ERR_INVALID_ARGUMENT;
                break;
            }
            const auto t_m1 = std::chrono::steady_clock::now();
            cl_match_sets view;
            cl_owned_match_sets_view(ms.get(), &view);
            if (getenv("CL_POLISH_DEBUG")) {
                for (int side = 0; side < 2; ++side) {
                    const cl_base_graph& g = side ? g2 : g1;
                    const SubPaths& sp = side ? slot_paths[ib] : slot_paths[ia];
                    bool shown = false;
                    for (uint64_t p = 0; p < g.n_paths; ++p) {
                        const uint64_t plen = g.path_off[p + 1] - g.path_off[p], want = std::get<2>(sp.of[p]) - std::get<1>(sp.of[p]) + 1;
                        if (plen != want && !shown) { shown = true; fprintf(stderr, "[polish] region %zu merge %llu side %d: path %llu of the slot graph has %llu steps, its stretch %llu .. %llu has %llu (graph: %llu paths, stretches: %zu)\n", ri, (unsigned long long)k, side, (unsigned long long)p, (unsigned long long)plen, (unsigned long long)std::get<1>(sp.of[p]), (unsigned long long)std::get<2>(sp.of[p]), (unsigned long long)want, (unsigned long long)g.n_paths, sp.of.size()); }
                    }
                }
            }
            if (dbg.f) {
                std::vector<uint64_t> flat;
                for (uint64_t st_ = 0; st_ < view.n_sets; ++st_) {
                    const uint64_t w1 = view.set_off1[st_ + 1] - view.set_off1[st_], w2 = view.set_off2[st_ + 1] - view.set_off2[st_];
                    flat.push_back(w1); flat.push_back(w2); flat.push_back(w1 ? view.walk_off1[view.set_off1[st_] + 1] - view.walk_off1[view.set_off1[st_]] : 0);
                    flat.push_back(view.count1[st_]); flat.push_back(view.count2[st_]); flat.push_back(view.full_length[st_]);
                    for (uint64_t w = view.set_off1[st_]; w < view.set_off1[st_ + 1]; ++w) flat.push_back(view.nodes1[view.walk_off1[w]]);
                    for (uint64_t w = view.set_off2[st_]; w < view.set_off2[st_ + 1]; ++w) flat.push_back(view.nodes2[view.walk_off2[w]]);
                }
                dbg.u64(rp + "m" + std::to_string(k) + ".matches", flat);
            }
            cl_core_align_result al;
            if ((rc = cl_core_align(ctx, &g1, &g2, &view, &ap, &al))) break;
            if (timing) {
                const auto t_m2 = std::chrono::steady_clock::now();
                t_induce += std::chrono::duration<double, std::milli>(t_m1 - t_m0).count();
                t_align += std::chrono::duration<double, std::milli>(t_m2 - t_m1).count();
                t_chain += al.chain_ms; t_part += al.partition_ms; t_stitch += al.stitch_ms; ++n_small_merges;
            }
            const uint64_t slot = plan.n_leaves + k;
            if (dbg.f) dbg.put(rp + "m" + std::to_string(k) + ".alignment", 2, al.alignment.pairs, 2 * al.alignment.n_pairs, 8);
            rc = cl_fuse(&g1, &g2, al.alignment.pairs, al.alignment.n_pairs, &slot_graph[slot]);
            cl_core_align_result_free(&al);
            if (rc) break;
            slot_paths[slot].of = slot_paths[ia].of;
            slot_paths[slot].of.insert(slot_paths[slot].of.end(), slot_paths[ib].of.begin(), slot_paths[ib].of.end());
            cl_owned_base_graph_free(slot_graph[ia]); slot_graph[ia] = nullptr;
            cl_owned_base_graph_free(slot_graph[ib]); slot_graph[ib] = nullptr;
        }
        if (rc) { drop(); return rc; }
        realigned[ri].graph = slot_graph[n_slots - 1];
        slot_graph[n_slots - 1] = nullptr;
        realigned[ri].paths = slot_paths[n_slots - 1];
        drop();
        return CL_OK;
    };
    const unsigned n_threads = dbg.f ? 1u : (unsigned)std::min<size_t>(std::max(1u, n_ctx), regions.size());   // (the debug dump is written in region order)
    if (n_threads <= 1) {
        for (size_t ri = 0; ri < regions.size(); ++ri)
            if ((rc = realign_region(ctx, ri))) return fail(rc);
    } else {
        std::atomic<size_t> next{0};
        std::atomic<int> first_rc{CL_OK};
        std::vector<std::thread> threads;
        for (unsigned t = 0; t < n_threads; ++t)
            threads.emplace_back([&, t] {
                (void)hipSetDevice(ctxs[t]->device);
                for (size_t ri; first_rc.load() == CL_OK && (ri = next.fetch_add(1)) < regions.size();) {
                    const int r = realign_region(ctxs[t], ri);
                    if (r) {
                        int expected = CL_OK;
                        if (first_rc.compare_exchange_strong(expected, r) && ctxs[t] != ctx) cl_set_error(ctx, "%s", cl_last_error(ctxs[t]));
                    }
                }
            });
        for (auto& th : threads) th.join();
        if ((rc = first_rc.load())) return fail(rc);
    }
    if (timing) fprintf(stderr, "[cl_polish] %zu regions, %llu merges: induced matches %.0f ms, core_align %.0f ms (chain %.0f, partition %.0f, stitch %.0f)\n", regions.size(),
                        (unsigned long long)n_small_merges, t_induce, t_align, t_chain, t_part, t_stitch);
    integrate(root, realigned);
    for (auto& r : realigned) cl_owned_base_graph_free(r.graph);
    cl_owned_base_graph* result = root.owned();
    cl_purge_uncovered(*result);
    *out = result;
    return CL_OK;
}
