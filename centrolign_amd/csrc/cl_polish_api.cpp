// cl_polish_api.cpp — the polishing step of cyclisation (Core::polish_cyclized_graph, src/core.cpp:650-767):
//
//   cl_identify_inconsistencies   InconsistencyIdentifier::identify_inconsistencies (include/centrolign/inconsistency_identifier.hpp:66-343,
//                                 src/inconsistency_identifier.cpp): tight cycles, indels placed inconsistently across a bond, merged along
//                                 their chains and padded with flanking sequence — node pairs that bound the regions to realign
//
// Host code (graph bookkeeping, as in the reference); the realignments themselves re-enter the hot path through cl_core_align.
#include <algorithm>
#include <cstring>
#include <deque>
#include <list>
#include <map>
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
