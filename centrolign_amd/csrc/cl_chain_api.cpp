// cl_chain_api.cpp — host side of the chaining seams (include/centrolign_amd.h): cl_chain_sparse_affine / cl_chain_sparse
// (the Anchorer's chaining DPs, include/centrolign/anchorer.hpp:1511-2547, with the DP itself on the GPU,
// chain_kernels.hip) and cl_anchor_chain (Anchorer::anchor_chain, :958-1329, with fill-in re-anchoring :619-956).
//
// chain_dp_batch runs K independent chaining instances in one device pass; its host work is O(M) or O(M log M):
//   * the coordinate system: PathMerge tables (path_merge.hpp:96-277), post-switch distances
//     (post_switch_distances.hpp:44-81), source/query shifts and offsets per (chain1, chain2) (anchorer.hpp:1875-1904),
//     forward-edge existence (forward_edges.hpp:40-53 with the masks of anchorer.hpp:1752-1810), anchor weights
//     (score_function.hpp:51-75);
//   * ordering the match pairs by the topological position of their first graph-1 node, so that every possible
//     predecessor of a pair comes before it;
//   * after the device has produced every DP value: the optimum (anchorer.hpp:2483-2499), and for each pair on the
//     chain the predecessor the reference would have recorded.  The DP VALUE is a maximum and is order-free, the
//     BACKPOINTER is whatever the reference's search trees return among equal maxima: first candidate query in its
//     loop order whose value reaches the maximum (strict '>' in MatchBank::update_dp, match_bank.hpp:177), and inside
//     that query the tree's traversal rule — first unit met by range_max; inside a cross tree the larger outer index
//     (values there are (score, index) pairs, orthogonal_max_search_tree.hpp:76); inside a gap-free subtree the
//     earliest inserted (strict '>' in MaxSearchTree::update, max_search_tree.hpp:318-358).  replay_units evaluates that
//     rule directly on the sorted keys; it only runs when a maximum is attained more than once.
#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <unistd.h>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "chain_device.h"
#include "cl_internal.hpp"
#include "stitch_host.hpp"

hipError_t cl_chain_launch_inter(const ClChainDevice& D, uint32_t block_first, uint32_t block_count, uint32_t src_block_lo,
                                 uint32_t src_block_hi, uint32_t max_recs, uint32_t tile_recs, hipStream_t stream);
hipError_t cl_chain_launch_intra(const ClChainDevice& D, uint32_t block_first, uint32_t block_count, uint32_t near_blocks, hipStream_t stream);
hipError_t cl_chain_launch_group(const ClChainDevice& D, uint32_t s0, uint32_t s1, uint32_t n_group_recs, int* dp_enc, hipStream_t stream);   // chain_kernels.hip: many combinations
hipError_t cl_chain_launch_walk(const ClChainDevice& D, uint32_t first, uint32_t count, hipStream_t stream, hipEvent_t done);
hipError_t cl_chain_launch_walk_fold(const ClChainDevice& D, uint32_t fold, uint32_t first, uint32_t count, hipStream_t stream, hipEvent_t done);
hipError_t cl_chain_launch_walk2(const ClChainDevice& D, uint32_t first, uint32_t count, uint32_t qpt, uint32_t n_help, hipStream_t stream, hipEvent_t done);   // chain_walk2.hip
uint32_t cl_chain_walk2_helpers(uint32_t qpt);
hipError_t cl_chain_launch_own_rec(const ClChainDevice& D, uint32_t max_recs, hipStream_t stream);
hipError_t cl_chain_acc_row(const ClChainDevice& D, uint32_t s, int* out, hipStream_t stream);
hipError_t cl_chain_expand_queries(const uint32_t* fa_qt, const uint32_t* fa_d, const uint32_t* fb_qoff, const uint32_t* fb_d, const uint32_t* combo_tags, uint32_t n_combos,
                                   uint32_t n_pairs, uint32_t* qt, uint32_t* qoff, int32_t* q, hipStream_t stream);
// chain_far.hip
hipError_t cl_chain_far_init(const ClChainDevice& D, const uint32_t* d_base, uint32_t max_padded, uint32_t r_pad, int32_t sig_bias, uint32_t band_shift,
                             uint32_t off_bits, uint32_t* key_off, uint32_t* key_band, uint32_t* idx, hipStream_t stream);
size_t cl_chain_far_sort_temp_bytes(uint32_t n);
hipError_t cl_chain_far_sort32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                               uint32_t n, int end_bit, hipStream_t stream);
hipError_t cl_chain_far_node_keys(const uint32_t* order, uint32_t n, uint32_t shift, uint32_t* node_key, hipStream_t stream);
hipError_t cl_chain_far_layout(const uint32_t* perm, const uint32_t* key, uint32_t n, uint32_t* arena, uint32_t ord_off, const uint32_t* ix, uint32_t n_ix,
                               hipStream_t stream);
hipError_t cl_chain_far_seal(const ClChainDevice& D, const ClFarDevice& F, const uint32_t* items, uint32_t item0, uint32_t n_items, uint32_t n_big, hipStream_t stream, hipEvent_t done);
hipError_t cl_chain_far_launch(const ClChainDevice& D, const ClFarDevice& F, uint32_t first, uint32_t count, uint32_t end_block, hipStream_t stream, hipEvent_t done);
hipError_t cl_chain_far_merge(const ClChainDevice& D, const int* slot, uint32_t first, uint32_t count, uint32_t share_n, uint32_t share_i, hipStream_t stream);
hipError_t cl_chain_sort_values(const float* val, uint32_t n, int* keys_in, uint32_t* idx_in, int* keys_out, uint32_t* idx_out,
                                void* temp, size_t* temp_bytes, hipStream_t stream);

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

// post_switch_distances.hpp:44-81; 0xFFFFFFFF = none (the reference's size_t(-1), used modulo 2^32)
struct PostSwitchTable {
    uint64_t n = 0;
    std::vector<uint32_t> d;
    void build(const cl_base_graph& g, const clhost::PathMergeTable& pm) {
        n = g.n_nodes;
        d.assign(pm.chain_size() * n, 0);
        std::vector<uint32_t> order;
        clhost::topological_order(g, order);
        for (uint32_t v : order)
            for (uint64_t p = 0; p < pm.chain_size(); ++p) {
                uint32_t* row = &d[p * n];
                const uint32_t pr = pm.predecessor_index(v, p);
                for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) {
                    const uint32_t u = g.prev_idx[e];
                    if (pm.index_on(u, p) == pr) { row[v] = 1; break; }
                    if (pm.predecessor_index(u, p) == pr) {
                        const uint64_t thru = (uint64_t)row[u] + 1;
                        if (row[v] == 0 || row[v] > thru) row[v] = (uint32_t)thru;
                    }
                }
            }
    }
    uint32_t distance(uint64_t v, uint64_t p) const { const uint32_t x = d[p * n + v]; return x == 0 ? kNone : x; }
};

using clhost::anchor_weight;
using clhost::min_source_sink;

inline int enc(float f) { int b; memcpy(&b, &f, 4); if (b == (int)0x80000000) b = 0; return b >= 0 ? b : b ^ 0x7FFFFFFF; }   // -0.0f == +0.0f, as on the device
inline float dec(int k) { int b = k >= 0 ? k : k ^ 0x7FFFFFFF; float f; memcpy(&f, &b, 4); return f; }

// heap node of the r-th smallest key in a MaxSearchTree / outer OrthogonalMaxSearchTree of n keys
// (implicit complete binary tree filled by an in-order walk, max_search_tree.hpp:113-150)
std::vector<uint32_t> heap_of_rank(size_t n) {
    std::vector<uint32_t> h(n);
    size_t next = 0;
    std::vector<std::pair<size_t, bool>> st;
    if (n) st.emplace_back(0, false);
    while (!st.empty()) {
        auto& top = st.back();
        if (!top.second) {
            top.second = true;
            if (2 * top.first + 1 < n) st.emplace_back(2 * top.first + 1, false);
        } else {
            const size_t x = top.first;
            h[next++] = (uint32_t)x;
            st.pop_back();
            if (2 * x + 2 < n) st.emplace_back(2 * x + 2, false);
        }
    }
    return h;
}

inline bool in_subtree(size_t y, size_t x) {
    while (y > x) y = (y - 1) / 2;
    return y == x;
}

// Replays the unit order of range_max over rank interval [rlo, rhi) (max_search_tree.hpp:361-444,
// orthogonal_max_search_tree.hpp:343-544) on a tree of n keys: calls node(x) for the nodes on the search path that lie
// in range and subtree(x) for the off-path subtrees, in the order the reference inspects them; stops at the first
// callback that returns true.
template <class NodeF, class SubF>
void replay_units(size_t n, const std::vector<uint32_t>& rank_of_heap, size_t rlo, size_t rhi, NodeF node, SubF subtree) {
    auto in = [&](size_t x) { return rank_of_heap[x] >= rlo && rank_of_heap[x] < rhi; };
    size_t c = 0;
    while (c < n && !in(c)) c = rank_of_heap[c] >= rhi ? 2 * c + 1 : 2 * c + 2;
    if (c >= n) return;
    if (node(c)) return;
    size_t lc = 2 * c + 1, rc = 2 * c + 2;
    while (lc < n) {
        if (rank_of_heap[lc] >= rlo) {
            if (node(lc)) return;
            if (2 * lc + 2 < n && subtree(2 * lc + 2)) return;
            lc = 2 * lc + 1;
        } else lc = 2 * lc + 2;
    }
    while (rc < n) {
        if (rank_of_heap[rc] < rhi) {
            if (node(rc)) return;
            if (2 * rc + 1 < n && subtree(2 * rc + 1)) return;
            rc = 2 * rc + 2;
        } else rc = 2 * rc + 1;
    }
}

// records ordered by (key, slot): slots are unique, so a placement by slot followed by a stable counting sort on the key
// replaces the comparison sort (the big trees of a 1 Mbp problem hold every match pair)
template <class KeyF, class SlotF>
void sort_by_key_then_slot(std::vector<uint32_t>& items, uint64_t n_slots, KeyF key, SlotF slot, bool already_by_slot = false) {
    if (items.size() < 4096) {
        std::sort(items.begin(), items.end(), [&](uint32_t a, uint32_t b) { return key(a) != key(b) ? key(a) < key(b) : slot(a) < slot(b); });
        return;
    }
    std::vector<uint32_t> by_slot;
    if (already_by_slot) by_slot = items;   // (the caller's list comes in slot order: a run of one shift of the (shift, slot) order)
    else {
        std::vector<uint32_t> at(n_slots, 0xFFFFFFFFu);
        for (uint32_t it : items) at[slot(it)] = it;
        by_slot.reserve(items.size());
        for (uint64_t sl = 0; sl < n_slots; ++sl) if (at[sl] != 0xFFFFFFFFu) by_slot.push_back(at[sl]);
    }
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (uint32_t it : by_slot) { lo = std::min<int64_t>(lo, key(it)); hi = std::max<int64_t>(hi, key(it)); }
    if ((uint64_t)(hi - lo) > 64 * (uint64_t)by_slot.size() + (1u << 16)) {   // keys far sparser than items: a comparison sort on the key alone (stable: slot order is kept)
        std::stable_sort(by_slot.begin(), by_slot.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
        items.swap(by_slot);
        return;
    }
    std::vector<uint32_t> start((size_t)(hi - lo) + 2, 0);
    for (uint32_t it : by_slot) ++start[(size_t)(key(it) - lo) + 1];
    for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
    for (uint32_t it : by_slot) items[start[(size_t)(key(it) - lo)]++] = it;
}

// a gap-free search tree of the reference, as its sorted key list + implicit heap layout (built on first use)
struct GapFreeTree {
    struct Member { uint32_t off, slot, rec; };
    std::vector<Member> mem;
    std::vector<uint32_t> heap, rank_of_heap;
    bool built = false;
};

struct Combo {
    uint32_t p1 = 0, p2 = 0;
    ClRawVec<uint32_t> rec_s, ins_t, off;   // (filled completely by the record / query passes)
    ClRawVec<int32_t> sigma;
    uint32_t *qt = nullptr, *qoff = nullptr;   // dense queries [M]: slices of ONE block per array for all combinations (chain_dp_batch_impl: dense_qt / dense_qoff / dense_q) —
    int32_t* q = nullptr;                      // a polishing step runs thousands of small DPs over hundreds of combinations: three heap blocks per combination were most of their "queries" phase
    std::vector<uint32_t> prefix;
    // device
    DevBuf<uint32_t> d_rec_s, d_ins_t, d_off, d_prefix, d_qt, d_qoff, d_own_rec;
    DevBuf<int32_t> d_sigma, d_q;
    DevBuf<float> d_val;
    DevBuf<int> d_acc;
    // lazily built for tie resolution, per chaining instance of the batch
    struct SubRecs {
        std::vector<uint32_t> recs;         // the instance's records in this combination
        std::vector<uint32_t> ortho_order;  // ... sorted by (sigma, slot)
        std::vector<uint32_t> ortho_heap, ortho_rank_of_heap;
        std::unordered_map<int32_t, GapFreeTree> diag_tree;
    };
    std::unordered_map<uint32_t, SubRecs> per_sub;
    bool split_built = false;
    void release(bool quiesced = false) {
        d_rec_s.release(quiesced); d_ins_t.release(quiesced); d_off.release(quiesced); d_prefix.release(quiesced); d_qt.release(quiesced); d_qoff.release(quiesced);
        d_sigma.release(quiesced); d_q.release(quiesced); d_val.release(quiesced); d_acc.release(quiesced); d_own_rec.release(quiesced);
    }
};

}  // namespace

extern "C" {

void cl_chain_params_default(cl_chain_params* p) {
    // src/parameters.cpp:39-59
    const double go[3] = {1.25, 50.0, 5000.0}, ge[3] = {2.5, 0.1, 0.0015};
    for (int i = 0; i < 3; ++i) { p->gap_open[i] = go[i]; p->gap_extend[i] = ge[i]; }
    p->anchor_score_function = 2;
    p->pair_count_power = 0.5;
    p->length_intercept = 2250.0;
    p->length_decay_power = 2.0;
    p->global_anchoring = 1;   // src/parameters.cpp:60
}

void cl_chain_result_free(cl_chain_result* r) {
    if (!r) return;
    free(r->anchors);
    free(r->dp);
    memset(r, 0, sizeof(*r));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// chain_dp_batch: K independent chaining instances in ONE device pass.
//
// sparse == false: sparse_affine_chain_dp (anchorer.hpp:1812-2471); sparse == true: sparse_chain_dp (:1511-1750), which is the
// same sweep with ONE tree per (chain of e1, chain of e2), no shift condition and no gap cost: on the device it is the
// affine machinery with every shift set to 0, so that only the "same diagonal" maximum is ever fed.
//
// Batching (the fill-in re-anchoring of anchorer.hpp:619-699 runs one small DP per gap of the chain, thousands per merge):
// the device only ever COMPARES the path coordinates (index of the end node / predecessor index of the start node), the
// arithmetic uses the shifts, which are stored separately.  Instance k therefore gets its graph-1 coordinates shifted UP by
// the total size of the instances before it and its graph-2 coordinates shifted up by the total size of the instances AFTER
// it: a record of another instance then fails one of the two range conditions (index1 <= query1, index2 < query2), and
// the whole batch is one all-pairs sweep over the concatenated match pairs.
namespace {

// what the DP needs of its graph 1 alone — topological positions and depths (longest path from a source) — so that the two whole-graph DPs of one cl_anchor_chain
// (the gap-free one of the scale estimate, the affine one) share one walk over the graph
struct GraphOrder {
    std::vector<uint32_t> pos, depth;
    void build(const cl_base_graph& g) {
        std::vector<uint32_t> order;
        clhost::topological_order(g, order);
        pos.resize(g.n_nodes);
        for (uint32_t i = 0; i < order.size(); ++i) pos[order[i]] = i;
        // depth = longest path from a source: a predecessor m of m' ends strictly before m' starts, so
        // depth(b1(m')) >= depth(b1(m)) + len(m); pairs whose start depths fall into one window of min_len consecutive
        // depths can never precede one another and are finalised together on the device ("group")
        depth.assign(g.n_nodes, 0);
        for (uint32_t v : order)
            for (uint64_t e = g.next_off[v]; e < g.next_off[v + 1]; ++e)
                depth[g.next_idx[e]] = std::max(depth[g.next_idx[e]], depth[v] + 1);
    }
};

struct ChainSub {
    const cl_base_graph* g[2] = {nullptr, nullptr};   // DP orientation: g[0] plays graph1
    const GraphOrder* order1 = nullptr;                // of g[0], prebuilt (built here if null)
    bool tableau = true;                               // graphs carry sentinels: PathMerge gets the pseudo-path (path_merge.hpp:148-160)
    bool chain_merge = false;                          // tables built here are ChainMerge tables (the CLI's -g 1: anchorer.hpp:659-660 with XMerge = ChainMerge)
    const cl_match_sets* ms = nullptr;                 // the sets; DP set s = ms set order[s] (identity if null), and with
    const uint64_t* order = nullptr;                   // swap_sides the DP's graph-1 walks are the sets' walks2
    bool swap_sides = false;                           // (anchorer.hpp:1179-1182), without copying anything
    uint64_t num_match_sets = 0;
    const clhost::PathMergeTable* x[2] = {nullptr, nullptr};   // prebuilt tables of g[0], g[1] (built here if null)
    const PostSwitchTable* sw[2] = {nullptr, nullptr};
    bool anchored = false;                             // sources / sinks given (global anchoring, fill-in)
    std::vector<uint32_t> src[2], snk[2];
    std::vector<uint32_t> tag[2];                      // local chain id -> batch-wide chain tag; empty = identity
    // masked matches (anchorer.hpp:144; MatchBank skips them, match_bank.hpp:187-215,252-268): (set of `ms`, idx1, idx2) in the sets' own
    // sides -> true if the pair does not take part in the DP.  Empty = no mask
    std::function<bool(uint64_t, uint32_t, uint32_t)> masked;
};

struct ChainSubResult {
    std::vector<uint32_t> chain;     // (set, idx1, idx2) per anchor, DP orientation
    std::vector<int64_t> gap;        // affine only: [n+1] gap before anchor i; [0] from the sources, [n] to the sinks
    std::vector<double> gap_score;   // (anchorer.hpp:2443-2468)
    uint64_t n_ties = 0;
};

struct ChainTimings { float device_ms = 0, prep_ms = 0, index_ms = 0, traceback_ms = 0; uint64_t n_pairs = 0; double pair_evals = 0; uint32_t n_combos = 0; };

struct SubCtx {
    clhost::PathMergeTable own_x[2];
    PostSwitchTable own_sw[2];
    const clhost::PathMergeTable* x[2] = {nullptr, nullptr};
    const PostSwitchTable* sw[2] = {nullptr, nullptr};
    GraphOrder own_order;
    const uint32_t *pos1 = nullptr, *depth1 = nullptr;
    std::vector<char> has_start, after_end;
    std::vector<std::pair<uint32_t, uint32_t>> walk_marks;   // (first, last node) of every graph-1 walk when some pairs are masked
    uint32_t pair_lo = 0, pair_hi = 0;
    uint32_t off_a = 0, off_b = 0;
    float min_score = 0.0f;
};

struct Pair { uint32_t sub, set, i1, i2, b1, e1, b2, e2; };

// the match sets of an instance as the DP sees them
struct SetView {
    const cl_match_sets* ms;
    const uint64_t* order;
    const uint64_t *so[2], *wo[2];
    const uint32_t* nd[2];
    explicit SetView(const ChainSub& sb) : ms(sb.ms), order(sb.order) {
        const int a = sb.swap_sides ? 1 : 0;
        so[a] = ms->set_off1; wo[a] = ms->walk_off1; nd[a] = ms->nodes1;
        so[1 - a] = ms->set_off2; wo[1 - a] = ms->walk_off2; nd[1 - a] = ms->nodes2;
    }
    uint64_t orig(uint64_t s) const { return order ? order[s] : s; }
    uint64_t n_walks(int side, uint64_t s) const { const uint64_t o = orig(s); return so[side][o + 1] - so[side][o]; }
    uint64_t walk(int side, uint64_t s, uint64_t j) const { return so[side][orig(s)] + j; }
    uint32_t front(int side, uint64_t w) const { return nd[side][wo[side][w]]; }
    uint32_t back(int side, uint64_t w) const { return nd[side][wo[side][w + 1] - 1]; }
    uint64_t length(uint64_t s) const { const uint64_t w = so[0][orig(s)]; return wo[0][w + 1] - wo[0][w]; }   // walks of a set share their length
    double weight(const cl_chain_params& cp, uint64_t s) const {
        const uint64_t o = orig(s);
        return anchor_weight(cp, ms->count1[o], ms->count2[o], length(s), ms->full_length[o]);
    }
};

}  // namespace

constexpr int kWalkStalled = -1000;   // internal: the walk kernel's workgroups were not all resident (chain_dp_batch retries without it)

// The walk's workgroups take a compute unit each (1 024 threads, up to 120 KB of LDS) and wait for one another: the DPs of several contexts of ONE process that are in
// flight together must fit the device together, or a walk gives up its bounded wait and the whole DP is repeated on the per-block kernels — 184 s instead of 11 s for a
// 156-combination merge of 50 x 1 Mbp beside three other workers (round 6, profiles/r06_configs4.json).  So a DP books the compute units its walk launches need before it
// enqueues them and gives them back when the device is done with it; a DP that does not fit beside the ones in flight waits for them (two 156-combination walks never
// fitted 256 units: they took turns by stalling).  Per device; the stall fallback stays for what is left (other processes on the device).
namespace {
struct WalkUnits {
    std::mutex m;
    std::condition_variable cv;
    uint32_t in_use[16] = {};
    static constexpr uint32_t kBudget = 232;   // of 256: the far / near / seal launches of the same DPs need room as well
    void take(int device, uint32_t need) {
        need = std::min(need, kBudget);
        std::unique_lock<std::mutex> lock(m);
        uint32_t& u = in_use[device & 15];
        cv.wait(lock, [&] { return u == 0 || u + need <= kBudget; });
        u += need;
    }
    void give(int device, uint32_t need) {
        need = std::min(need, kBudget);
        { std::lock_guard<std::mutex> lock(m); in_use[device & 15] -= need; }
        cv.notify_all();
    }
};
WalkUnits g_walk_units;
thread_local bool g_walk_exclusive = false;   // this thread's DP books the whole budget (the second attempt after a stall)
}  // namespace

static int chain_dp_batch_impl(cl_context* ctx, const std::vector<ChainSub>& subs, const cl_chain_params* cp, double local_scale,
                               bool sparse, std::vector<ChainSubResult>& results, ChainTimings& tm, std::vector<float>* dp_out, bool allow_walk);

// The walk kernel's workgroups (one per chain combination) wait for one another and must all be resident at once.  That holds whenever the
// device is not packed with OTHER contexts' workgroups; if a bounded wait ever expires, nothing has been decided yet (the status word is
// read before any result is used) and the DP is simply run again on the per-block kernels, which need no co-residency.
static int chain_dp_batch(cl_context* ctx, const std::vector<ChainSub>& subs, const cl_chain_params* cp, double local_scale,
                          bool sparse, std::vector<ChainSubResult>& results, ChainTimings& tm, std::vector<float>* dp_out) {
    const ChainTimings before = tm;
    ++cl_fallbacks.chain_dps;
    int rc = chain_dp_batch_impl(ctx, subs, cp, local_scale, sparse, results, tm, dp_out, true);
    if (rc != kWalkStalled) return rc;
    ++cl_fallbacks.walk_stalls;
    if (ctx->peers.n > 1) return CL_ERR_HIP;   // (inside a merge group the other members have gone on with this member's share: no second attempt)
    // round 6: ONE more attempt on the walk kernels with the device's whole walk budget booked for this DP (no other DP of the process in flight beside it) before the
    // per-block kernels, which take minutes where the walk takes seconds (a 156-combination DP of 1.19 M pairs: 184 s against 11 s)
    static const char* debug_stall = getenv("CL_CHAIN_DEBUG_STALL");   // (test hook: "first" = only the shared attempt "stalls"; any other value = every walk does: straight to the per-block kernels)
    if (!debug_stall || !strcmp(debug_stall, "first")) {
        if (getenv("CL_CHAIN_TIMING")) fprintf(stderr, "[chain_dp_batch]   walk kernel stalled: repeating the DP alone on the device\n");
        tm = before;
        g_walk_exclusive = true;
        rc = chain_dp_batch_impl(ctx, subs, cp, local_scale, sparse, results, tm, dp_out, true);
        g_walk_exclusive = false;
        if (rc != kWalkStalled) return rc;
        ++cl_fallbacks.walk_stalls;
    }
    if (getenv("CL_CHAIN_TIMING")) fprintf(stderr, "[chain_dp_batch]   walk kernel stalled: repeating the DP on the per-block kernels\n");
    tm = before;
    rc = chain_dp_batch_impl(ctx, subs, cp, local_scale, sparse, results, tm, dp_out, false);
    return rc == kWalkStalled ? CL_ERR_HIP : rc;
}

static int chain_dp_batch_impl(cl_context* ctx, const std::vector<ChainSub>& subs, const cl_chain_params* cp, double local_scale,
                               bool sparse, std::vector<ChainSubResult>& results, ChainTimings& tm, std::vector<float>* dp_out, bool allow_walk) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { const int jrc = cl_stitch_join(ctx); if (jrc) return jrc; }   // (a stitch pass of this context may still be out on the auxiliary streams this DP is about to use)
    const auto T0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const size_t K = subs.size();
    results.assign(K, ChainSubResult());
    std::vector<SubCtx> sc(K);
    const bool timing = getenv("CL_CHAIN_TIMING") != nullptr;
    auto tlap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[chain_dp_batch]   %-26s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tlap).count());
        tlap = std::chrono::steady_clock::now();
    };

    // match pairs in MatchBank iteration order (match_bank.hpp:252-268), instance by instance: slot = position in that order
    std::vector<Pair> pairs;
    uint64_t min_len = UINT64_MAX, span_a = 0, span_b = 0;
    for (size_t k = 0; k < K; ++k) {
        const ChainSub& sb = subs[k];
        const SetView ms(sb);
        sc[k].pair_lo = (uint32_t)pairs.size();
        if (sb.num_match_sets > sb.ms->n_sets) { cl_set_error(ctx, "num_match_sets exceeds the number of sets"); return CL_ERR_INVALID_ARGUMENT; }
        {   // the sets are the caller's: a walk is a non-empty run of node ids of its graph (anything else would index the tables below out of range)
            std::atomic<uint64_t> bad_set{UINT64_MAX};
            cl_parallel_for(sb.num_match_sets, [&](uint64_t sb0, uint64_t se0) {
                for (uint64_t s = sb0; s < se0; ++s)
                    for (int side = 0; side < 2; ++side) {
                        const uint64_t n_nodes = sb.g[side]->n_nodes;
                        for (uint64_t j = 0; j < ms.n_walks(side, s); ++j) {
                            const uint64_t w = ms.walk(side, s, j);
                            if (ms.wo[side][w + 1] <= ms.wo[side][w] || ms.front(side, w) >= n_nodes || ms.back(side, w) >= n_nodes) { bad_set = s; return; }
                        }
                    }
            }, 4096);
            if (bad_set.load() != UINT64_MAX) {
                const uint64_t s = bad_set.load();
                cl_set_error(ctx, "match set %llu (of %llu) holds an empty walk or a node id outside its graph (%llu and %llu nodes; the set has %llu x %llu walks of length %llu)",
                             (unsigned long long)s, (unsigned long long)sb.num_match_sets, (unsigned long long)sb.g[0]->n_nodes, (unsigned long long)sb.g[1]->n_nodes,
                             (unsigned long long)ms.n_walks(0, s), (unsigned long long)ms.n_walks(1, s), (unsigned long long)(ms.n_walks(0, s) ? ms.length(s) : 0));
                return CL_ERR_INVALID_ARGUMENT;
            }
        }
        if (!sb.masked && sb.num_match_sets >= 4096) {
            // (the usual case, no mask: every set's pairs have their place by a prefix sum over n1 * n2, the sets are filled in side by side)
            const uint64_t ns = sb.num_match_sets;
            std::vector<uint64_t> at(ns + 1, 0);
            bool too_many = false;
            for (uint64_t s = 0; s < ns; ++s) {
                const uint64_t n1 = ms.n_walks(0, s), n2 = ms.n_walks(1, s);
                too_many |= n1 >= 65535 || n2 >= 65535;
                if (n1 && n2) min_len = std::min<uint64_t>(min_len, ms.length(s));
                at[s + 1] = at[s] + n1 * n2;
            }
            if (too_many) { cl_set_error(ctx, "a match set has too many walks"); return CL_ERR_INVALID_ARGUMENT; }
            const uint64_t base = pairs.size();
            if (base + at[ns] >= (1ull << 31)) { cl_set_error(ctx, "too many match pairs"); return CL_ERR_INVALID_ARGUMENT; }
            pairs.resize(base + at[ns]);
            cl_parallel_for(ns, [&](uint64_t sb0, uint64_t se0) {
                for (uint64_t s = sb0; s < se0; ++s) {
                    const uint64_t n1 = ms.n_walks(0, s), n2 = ms.n_walks(1, s);
                    Pair* out = pairs.data() + base + at[s];
                    for (uint64_t j = 0; j < n1; ++j) {
                        const uint64_t w1 = ms.walk(0, s, j);
                        const uint32_t b1 = ms.front(0, w1), e1 = ms.back(0, w1);
                        for (uint64_t q = 0; q < n2; ++q) {
                            const uint64_t w2 = ms.walk(1, s, q);
                            *out++ = Pair{(uint32_t)k, (uint32_t)s, (uint32_t)j, (uint32_t)q, b1, e1, ms.front(1, w2), ms.back(1, w2)};
                        }
                    }
                }
            }, 2048);
        } else
        for (uint64_t s = 0; s < sb.num_match_sets; ++s) {
            const uint64_t n1 = ms.n_walks(0, s), n2 = ms.n_walks(1, s);
            if (n1 >= 65535 || n2 >= 65535) { cl_set_error(ctx, "match set %llu has too many walks", (unsigned long long)s); return CL_ERR_INVALID_ARGUMENT; }
            if (n1 && n2) min_len = std::min<uint64_t>(min_len, ms.length(s));
            const uint64_t ms_set = sb.order ? sb.order[s] : s;
            for (uint64_t j = 0; j < n1; ++j) {
                const uint64_t w1 = ms.walk(0, s, j);
                const uint32_t b1 = ms.front(0, w1), e1 = ms.back(0, w1);
                if (sb.masked) sc[k].walk_marks.emplace_back(b1, e1);   // the forward-edge masks see every walk, masked or not
                for (uint64_t q = 0; q < n2; ++q) {
                    if (sb.masked && (sb.swap_sides ? sb.masked(ms_set, (uint32_t)q, (uint32_t)j) : sb.masked(ms_set, (uint32_t)j, (uint32_t)q))) continue;
                    const uint64_t w2 = ms.walk(1, s, q);
                    pairs.push_back(Pair{(uint32_t)k, (uint32_t)s, (uint32_t)j, (uint32_t)q, b1, e1, ms.front(1, w2), ms.back(1, w2)});
                }
            }
        }
        if (pairs.size() >= (1ull << 31)) { cl_set_error(ctx, "too many match pairs"); return CL_ERR_INVALID_ARGUMENT; }
        sc[k].pair_hi = (uint32_t)pairs.size();
        if (sc[k].pair_hi > sc[k].pair_lo) { span_a += sb.g[0]->n_nodes + 2; span_b += sb.g[1]->n_nodes + 2; }
    }
    const uint64_t M = pairs.size();
    tm.n_pairs = M;
    if (M == 0) return CL_OK;
    if (span_a >= 0xFFFFFFFFull || span_b >= 0xFFFFFFFFull) { cl_set_error(ctx, "batch too large for 32-bit path coordinates"); return CL_ERR_INVALID_ARGUMENT; }
    if (min_len == 0 || min_len == UINT64_MAX) min_len = 1;

    lap("pairs");
    // per-instance coordinate systems; chain tags -> dense combination table
    uint32_t n_tag[2] = {0, 0};
    {
        uint64_t run_a = 0, run_b = span_b;
        for (size_t k = 0; k < K; ++k) {
            SubCtx& c = sc[k];
            if (c.pair_hi == c.pair_lo) continue;
            const ChainSub& sb = subs[k];
            for (int side = 0; side < 2; ++side) {
                if (sb.x[side] && sb.sw[side]) { c.x[side] = sb.x[side]; c.sw[side] = sb.sw[side]; }
                else {
                    if (!(sb.chain_merge ? c.own_x[side].build_chain_merge(*sb.g[side], sb.tableau) : c.own_x[side].build(*sb.g[side], sb.tableau))) { cl_set_error(ctx, "graph is not acyclic"); return CL_ERR_CYCLIC_GRAPH; }
                    c.own_sw[side].build(*sb.g[side], c.own_x[side]);
                    c.x[side] = &c.own_x[side];
                    c.sw[side] = &c.own_sw[side];
                }
                for (uint32_t p = 0; p < c.x[side]->chain_size(); ++p)
                    n_tag[side] = std::max(n_tag[side], (sb.tag[side].empty() ? p : sb.tag[side][p]) + 1);
            }
            run_b -= sb.g[1]->n_nodes + 2;
            c.off_a = (uint32_t)run_a;
            c.off_b = (uint32_t)run_b;
            run_a += sb.g[0]->n_nodes + 2;
            const cl_base_graph& g1 = *sb.g[0];
            const GraphOrder* go = sb.order1;
            if (!go) { c.own_order.build(g1); go = &c.own_order; }
            c.pos1 = go->pos.data();
            c.depth1 = go->depth.data();
            c.has_start.assign(g1.n_nodes, 0);
            c.after_end.assign(g1.n_nodes, 0);
            for (uint32_t s = c.pair_lo; s < c.pair_hi; ++s) { c.has_start[pairs[s].b1] = 1; c.after_end[pairs[s].e1] = 1; }
            for (const auto& m : c.walk_marks) { c.has_start[m.first] = 1; c.after_end[m.second] = 1; }
            // nodes that follow the end of some match (anchorer.hpp:1776-1797)
            std::vector<uint32_t> st;
            for (uint64_t v = 0; v < g1.n_nodes; ++v)
                if (c.after_end[v]) {
                    st.push_back((uint32_t)v);
                    while (!st.empty()) {
                        const uint32_t h = st.back();
                        st.pop_back();
                        for (uint64_t e = g1.next_off[h]; e < g1.next_off[h + 1]; ++e)
                            if (!c.after_end[g1.next_idx[e]]) { c.after_end[g1.next_idx[e]] = 1; st.push_back(g1.next_idx[e]); }
                    }
                }
        }
    }
    auto tag_of = [&](uint32_t k, int side, uint32_t p) { return subs[k].tag[side].empty() ? p : subs[k].tag[side][p]; };
    lap("tables");

    // processing order: by the depth of the pair's first graph-1 node, stable in slot order
    std::vector<uint32_t> by_s(M);  // sorted index -> slot
    {   // counting sort on the depth (stable)
        std::vector<uint32_t> key(M);
        uint32_t max_depth = 0;
        for (uint32_t slot = 0; slot < M; ++slot) { key[slot] = sc[pairs[slot].sub].depth1[pairs[slot].b1]; max_depth = std::max(max_depth, key[slot]); }
        std::vector<uint32_t> start((size_t)max_depth + 2, 0);
        for (uint32_t slot = 0; slot < M; ++slot) ++start[key[slot] + 1];
        for (size_t d = 0; d <= max_depth; ++d) start[d + 1] += start[d];
        for (uint32_t slot = 0; slot < M; ++slot) by_s[start[key[slot]]++] = slot;
    }
    std::vector<uint32_t> s_of_slot(M);
    for (uint32_t s = 0; s < M; ++s) s_of_slot[by_s[s]] = s;

    lap("sort by depth");
    std::vector<float> weight(M);
    {   // the weight depends on the match set only
        std::vector<std::vector<float>> set_weight(K);
        for (size_t k = 0; k < K; ++k) {
            if (sc[k].pair_hi == sc[k].pair_lo) continue;
            const SetView ms(subs[k]);
            set_weight[k].resize(subs[k].num_match_sets);
            cl_parallel_for(subs[k].num_match_sets, [&](uint64_t b, uint64_t e) {
                for (uint64_t st = b; st < e; ++st)
                    if (ms.n_walks(0, st)) set_weight[k][st] = (float)ms.weight(*cp, st);
            }, 4096);
        }
        cl_parallel_for(M, [&](uint64_t b, uint64_t e) {
            for (uint64_t s = b; s < e; ++s) { const Pair& p = pairs[by_s[s]]; weight[s] = set_weight[p.sub][p.set]; }
        });
    }
    // Anchored chains (global anchoring, anchorer.hpp:1069-1076; fill-in, :683-693): a chain's first anchor pays the lead
    // indel from the sources (affine, :2026-2039) or must be reachable from them (sparse, :1562-1582); its last anchor pays
    // the final indel to the sinks (:2426-2438 / :1724-1741).
    auto score_gap = [&](int32_t gap) -> float {   // anchorer.hpp:1929-1944
        float score = CL_CHAIN_NEG;
        if (gap == 0) score = 0.0f;
        else if (gap != INT32_MAX)
            for (int pw = 0; pw < 3; ++pw) score = std::max<float>(score, (float)(-local_scale * (cp->gap_open[pw] + cp->gap_extend[pw] * std::abs(gap))));
        return score;
    };
    auto measure_gap = [&](const SubCtx& c, uint32_t a1, uint32_t a2, uint32_t c1, uint32_t c2) -> int32_t {   // anchorer.hpp:1906-1927
        int32_t gap = INT32_MAX;
        if ((a1 == c1 || c.x[0]->reachable(a1, c1)) && (a2 == c2 || c.x[1]->reachable(a2, c2)))
            c.x[0]->for_each_chain_on(a1, [&](uint32_t p1) {
                c.x[1]->for_each_chain_on(a2, [&](uint32_t p2) {
                    const uint32_t src = c.x[0]->index_on(a1, p1) - c.x[1]->index_on(a2, p2);
                    const uint32_t qry = c.x[0]->predecessor_index(c1, p1) - c.x[1]->predecessor_index(c2, p2) + c.sw[0]->distance(c1, p1) - c.sw[1]->distance(c2, p2);
                    const int32_t here_gap = (int32_t)(src - qry);
                    if (std::abs(here_gap) < std::abs(gap)) gap = here_gap;
                });
            });
        return gap;
    };
    // the reference compares |gap| against the SIGNED running value (anchorer.hpp:1954, 1971, 1991)
    auto gap_from_sources = [&](uint32_t k, uint32_t c1, uint32_t c2) {
        int32_t best = INT32_MAX;
        for (uint32_t a : subs[k].src[0]) for (uint32_t b : subs[k].src[1]) { const int32_t h = measure_gap(sc[k], a, b, c1, c2); if (std::abs(h) < best) best = h; }
        return best;
    };
    auto gap_to_sinks = [&](uint32_t k, uint32_t a1, uint32_t a2) {
        int32_t best = INT32_MAX;
        for (uint32_t c : subs[k].snk[0]) for (uint32_t d : subs[k].snk[1]) { const int32_t h = measure_gap(sc[k], a1, a2, c, d); if (std::abs(h) < best) best = h; }
        return best;
    };
    lap("weights");
    std::vector<float> init_w(weight);   // the value of a chain that STARTS at the pair
    std::vector<float> final_term(M, 0.0f);  // per slot
    cl_parallel_for(M, [&](uint64_t s_begin, uint64_t s_end) {
    for (uint32_t s = (uint32_t)s_begin; s < s_end; ++s) {
        const uint32_t slot = by_s[s];
        const Pair& p = pairs[slot];
        const ChainSub& sb = subs[p.sub];
        if (!sb.anchored) continue;
        const SubCtx& c = sc[p.sub];
        if (!sparse) {
            const float lead = score_gap(gap_from_sources(p.sub, p.b1, p.b2));
            init_w[s] = lead == CL_CHAIN_NEG ? CL_CHAIN_NEG : weight[s] + lead;
            final_term[slot] = score_gap(gap_to_sinks(p.sub, p.e1, p.e2));
        } else {
            bool f1 = false, f2 = false, t = false;
            for (uint32_t a : sb.src[0]) if (a == p.b1 || c.x[0]->reachable(a, p.b1)) { f1 = true; break; }
            for (uint32_t b : sb.src[1]) if (b == p.b2 || c.x[1]->reachable(b, p.b2)) { f2 = true; break; }
            if (!f1 || !f2) init_w[s] = CL_CHAIN_NEG;
            for (uint32_t a : sb.snk[0]) {
                for (uint32_t b : sb.snk[1])
                    if ((a == p.e1 || c.x[0]->reachable(p.e1, a)) && (b == p.e2 || c.x[1]->reachable(p.e2, b))) { t = true; break; }
                if (t) break;
            }
            final_term[slot] = t ? 0.0f : CL_CHAIN_NEG;
        }
    }
    });
    if (!sparse)   // the score of aligning nothing: one indel from the sources to the sinks (anchorer.hpp:2419-2424)
        for (size_t k = 0; k < K; ++k) {
            if (!subs[k].anchored || sc[k].pair_hi == sc[k].pair_lo) continue;
            int32_t best = INT32_MAX;
            for (uint32_t c : subs[k].snk[0]) for (uint32_t d : subs[k].snk[1]) for (uint32_t a : subs[k].src[0]) for (uint32_t b : subs[k].src[1]) {
                const int32_t h = measure_gap(sc[k], a, b, c, d);
                if (std::abs(h) < best) best = h;
            }
            sc[k].min_score = score_gap(best);
        }

    lap("anchoring terms");
    // (chain1, chain2) combinations that hold at least one pair; the most populated one goes first (it is the one the
    // intra kernel keeps in registers)
    std::vector<uint32_t> combo_of((size_t)n_tag[0] * n_tag[1], kNone);
    std::vector<Combo> combos;
    std::vector<uint32_t> rec_off(M + 1, 0);
    ClRawVec<uint32_t> rec_combo, rec_pos;
    {
        // records per pair: one per (chain through e1, chain through e2); sparse_chain_dp files a match under the first chain
        // of each end only (anchorer.hpp:1621-1630).  Counted and filled in parallel over the pairs, numbered serially: a
        // combination's id is its order of first appearance, a record's position the number of earlier records of its combination
        cl_parallel_for(M, [&](uint64_t s_begin, uint64_t s_end) {
            for (uint64_t s = s_begin; s < s_end; ++s) {
                if (sparse) { rec_off[s + 1] = 1; continue; }
                const Pair& p = pairs[by_s[s]];
                const SubCtx& c = sc[p.sub];
                uint32_t k1 = 0, k2 = 0;
                c.x[0]->for_each_chain_on(p.e1, [&](uint32_t) { ++k1; });
                c.x[1]->for_each_chain_on(p.e2, [&](uint32_t) { ++k2; });
                rec_off[s + 1] = k1 * k2;
            }
        });
        for (uint64_t s = 0; s < M; ++s) rec_off[s + 1] += rec_off[s];
        const uint64_t R = rec_off[M];
        ClRawVec<uint32_t> r_tag(R), r_ins(R), r_off(R);
        ClRawVec<int32_t> r_sig(R);
        cl_parallel_for(M, [&](uint64_t s_begin, uint64_t s_end) {
            for (uint64_t s = s_begin; s < s_end; ++s) {
                const Pair& p = pairs[by_s[s]];
                const SubCtx& c = sc[p.sub];
                uint64_t r = rec_off[s];
                bool first1 = true;
                c.x[0]->for_each_chain_on(p.e1, [&](uint32_t p1) {
                    const bool take1 = !sparse || first1;
                    first1 = false;
                    if (!take1) return;
                    bool first2 = true;
                    c.x[1]->for_each_chain_on(p.e2, [&](uint32_t p2) {
                        const bool take2 = !sparse || first2;
                        first2 = false;
                        if (!take2) return;
                        const uint32_t i1 = c.x[0]->index_on(p.e1, p1), i2 = c.x[1]->index_on(p.e2, p2);
                        r_tag[r] = tag_of(p.sub, 0, p1) * n_tag[1] + tag_of(p.sub, 1, p2);
                        r_ins[r] = i1 + c.off_a;
                        r_off[r] = i2 + c.off_b;
                        r_sig[r] = sparse ? 0 : (int32_t)(i1 - i2);
                        ++r;
                    });
                });
            }
        });
        rec_combo.resize(R);
        rec_pos.resize(R);
        std::vector<uint32_t> combo_size;
        const size_t n_tags = combo_of.size();
        static const uint64_t par_min = [] { const char* e = getenv("CL_CHAIN_PAR_RECORDS_MIN"); return e ? (uint64_t)atoll(e) : (uint64_t)(1u << 20); }();   // (0 in tests: every size)
        if (R >= par_min && R > 0 && n_tags <= 4096) {
            // the same numbering in parallel: chunks of records count their tags, a combination's id is the order of first appearance of
            // its tag (smallest first record), a record's position the records of its combination in earlier chunks plus those before
            // it in its own
            const uint64_t n_chunks = std::min<uint64_t>(64, (R + (1u << 18) - 1) >> 18), chunk = (R + n_chunks - 1) / n_chunks;
            std::vector<std::vector<uint32_t>> hist(n_chunks, std::vector<uint32_t>(n_tags, 0));
            std::vector<std::vector<uint64_t>> first(n_chunks, std::vector<uint64_t>(n_tags, UINT64_MAX));
            cl_parallel_for(n_chunks, [&](uint64_t cb, uint64_t ce) {
                for (uint64_t c = cb; c < ce; ++c)
                    for (uint64_t r = c * chunk; r < std::min<uint64_t>(R, (c + 1) * chunk); ++r) {
                        const uint32_t t = r_tag[r];
                        if (hist[c][t]++ == 0) first[c][t] = r;
                    }
            }, 1);
            std::vector<std::pair<uint64_t, uint32_t>> order;   // (first record, tag)
            for (uint32_t t = 0; t < n_tags; ++t) {
                uint64_t f = UINT64_MAX;
                for (uint64_t c = 0; c < n_chunks; ++c) f = std::min(f, first[c][t]);
                if (f != UINT64_MAX) order.emplace_back(f, t);
            }
            std::sort(order.begin(), order.end());
            for (const auto& o : order) {
                combo_of[o.second] = (uint32_t)combos.size();
                combos.emplace_back();
                combos.back().p1 = o.second / n_tag[1];
                combos.back().p2 = o.second % n_tag[1];
                combo_size.push_back(0);
            }
            std::vector<std::vector<uint32_t>> base(n_chunks, std::vector<uint32_t>(n_tags, 0));
            for (uint32_t t = 0; t < n_tags; ++t) {
                uint32_t run = 0;
                for (uint64_t c = 0; c < n_chunks; ++c) { base[c][t] = run; run += hist[c][t]; }
                if (combo_of[t] != kNone) combo_size[combo_of[t]] = run;
            }
            cl_parallel_for(n_chunks, [&](uint64_t cb, uint64_t ce) {
                for (uint64_t c = cb; c < ce; ++c) {
                    std::vector<uint32_t>& next = base[c];
                    for (uint64_t r = c * chunk; r < std::min<uint64_t>(R, (c + 1) * chunk); ++r) {
                        rec_combo[r] = combo_of[r_tag[r]];
                        rec_pos[r] = next[r_tag[r]]++;
                    }
                }
            }, 1);
        } else
        for (uint64_t r = 0; r < R; ++r) {
            uint32_t& ci = combo_of[r_tag[r]];
            if (ci == kNone) {
                ci = (uint32_t)combos.size();
                combos.emplace_back();
                combos.back().p1 = r_tag[r] / n_tag[1];
                combos.back().p2 = r_tag[r] % n_tag[1];
                combo_size.push_back(0);
            }
            rec_combo[r] = ci;
            rec_pos[r] = combo_size[ci]++;
        }
        for (size_t ci = 0; ci < combos.size(); ++ci) {
            combos[ci].rec_s.resize(combo_size[ci]);
            combos[ci].ins_t.resize(combo_size[ci]);
            combos[ci].off.resize(combo_size[ci]);
            combos[ci].sigma.resize(combo_size[ci]);
        }
        cl_parallel_for(M, [&](uint64_t s_begin, uint64_t s_end) {
            for (uint64_t s = s_begin; s < s_end; ++s)
                for (uint64_t r = rec_off[s]; r < rec_off[s + 1]; ++r) {
                    Combo& cb = combos[rec_combo[r]];
                    const uint32_t at = rec_pos[r];
                    cb.rec_s[at] = (uint32_t)s;
                    cb.ins_t[at] = r_ins[r];
                    cb.off[at] = r_off[r];
                    cb.sigma[at] = r_sig[r];
                }
        });
    }
    if (combos.size() > 1) {
        size_t big = 0;
        for (size_t c = 1; c < combos.size(); ++c)
            if (combos[c].rec_s.size() > combos[big].rec_s.size()) big = c;
        if (big != 0) {
            std::swap(combos[0], combos[big]);
            for (auto& rc : rec_combo) rc = rc == 0 ? (uint32_t)big : rc == big ? 0u : rc;
            for (auto& v : combo_of) v = v == 0 ? (uint32_t)big : v == big ? 0u : v;
        }
    }
    lap("records");
    if (timing) {
        size_t total = 0, largest = 0;
        for (const Combo& c : combos) { total += c.rec_s.size(); largest = std::max(largest, c.rec_s.size()); }
        fprintf(stderr, "[chain_dp_batch]   %zu pairs, %zu combinations, %zu records (largest combination %zu)\n", (size_t)M, combos.size(), total, largest);
    }
    tm.n_combos = (uint32_t)combos.size();
    for (const Combo& c : combos) tm.pair_evals += 0.5 * (double)c.rec_s.size() * (double)M;
    const uint32_t n_blocks = (uint32_t)((M + kChainBlock - 1) / kChainBlock);
    // The queries are dense — one per pair and combination — but they factor: the insertion bound depends on (pair, chain of graph 1) only, the offset
    // bound on (pair, chain of graph 2), and the shift is a difference of one term of each (all modulo 2^32, as the reference's size_t arithmetic
    // truncated).  A single-instance DP of some size keeps the factors on the host (n_tag[0] + n_tag[1] arrays instead of 3 x combinations: 0.5 GB
    // instead of 9.4 GB at 25 + 25 paths and 1.25 M pairs) and lets the device multiply them out (chain_expand_queries_kernel); the traceback
    // multiplies out the handful it looks at.  CL_CHAIN_DENSE_QUERIES=1: the dense tables of rounds 1-3 (A/B).
    static const bool dense_env = getenv("CL_CHAIN_DENSE_QUERIES") != nullptr;
    const bool factored = K == 1 && !dense_env && (uint64_t)combos.size() * M > 100000;
    ClRawVec<uint32_t> fa_qt, fa_d, fb_qoff, fb_d;   // [tag][M]
    ClRawVec<uint32_t> dense_qt, dense_qoff;         // [combination][M] (the dense tables: small and batched DPs)
    ClRawVec<int32_t> dense_q;
    if (factored) {
        fa_qt.resize((size_t)n_tag[0] * M);
        fb_qoff.resize((size_t)n_tag[1] * M);
        if (!sparse) { fa_d.resize((size_t)n_tag[0] * M); fb_d.resize((size_t)n_tag[1] * M); }
        cl_parallel_for(M, [&](uint64_t s_begin, uint64_t s_end) {
            for (uint32_t s = (uint32_t)s_begin; s < s_end; ++s) {
                const Pair& p = pairs[by_s[s]];
                const SubCtx& c = sc[p.sub];
                const bool start = c.has_start[p.b1];
                for (uint32_t p1 = 0; p1 < c.x[0]->chain_size(); ++p1) {
                    const uint32_t pr = start ? c.x[0]->predecessor_index(p.b1, p1) : kNone;
                    // a forward edge exists only from a node that follows some match end (forward_edges.hpp:40-53)
                    const bool ok = pr != kNone && c.after_end[c.x[0]->node_at(p1, pr)];
                    const size_t at = (size_t)tag_of(p.sub, 0, p1) * M + s;
                    fa_qt[at] = ok ? pr + c.off_a : kNone;
                    if (!sparse) fa_d[at] = ok ? pr + c.sw[0]->distance(p.b1, p1) : 0u;
                }
                for (uint32_t p2 = 0; p2 < c.x[1]->chain_size(); ++p2) {
                    const uint32_t pr2 = c.x[1]->predecessor_index(p.b2, p2);
                    const size_t at = (size_t)tag_of(p.sub, 1, p2) * M + s;
                    fb_qoff[at] = pr2 + 1u + c.off_b;
                    if (!sparse) fb_d[at] = pr2 + c.sw[1]->distance(p.b2, p2);
                }
            }
        });
    } else {
    dense_qt.resize(combos.size() * (size_t)M);
    dense_qoff.resize(combos.size() * (size_t)M);
    dense_q.resize(combos.size() * (size_t)M);
    for (size_t ci = 0; ci < combos.size(); ++ci) { combos[ci].qt = dense_qt.data() + ci * (size_t)M; combos[ci].qoff = dense_qoff.data() + ci * (size_t)M; combos[ci].q = dense_q.data() + ci * (size_t)M; }
    cl_parallel_for(combos.size() * (size_t)M, [&](uint64_t b, uint64_t e) {   // 15 MB per combination at 1.25 M pairs, 25 combinations at the root
        std::fill(dense_qt.data() + b, dense_qt.data() + e, kNone);
        std::memset(dense_qoff.data() + b, 0, (e - b) * sizeof(uint32_t));
        std::memset(dense_q.data() + b, 0, (e - b) * sizeof(int32_t));
    }, 1u << 18);
    cl_parallel_for(M, [&](uint64_t s_begin, uint64_t s_end) {
    for (uint32_t s = (uint32_t)s_begin; s < s_end; ++s) {
        const Pair& p = pairs[by_s[s]];
        const SubCtx& c = sc[p.sub];
        if (!c.has_start[p.b1]) continue;
        for (uint32_t p1 = 0; p1 < c.x[0]->chain_size(); ++p1) {
            const uint32_t pr = c.x[0]->predecessor_index(p.b1, p1);
            // a forward edge exists only from a node that follows some match end (forward_edges.hpp:40-53)
            if (pr == kNone || !c.after_end[c.x[0]->node_at(p1, pr)]) continue;
            for (uint32_t p2 = 0; p2 < c.x[1]->chain_size(); ++p2) {
                const uint32_t ci = combo_of[(size_t)tag_of(p.sub, 0, p1) * n_tag[1] + tag_of(p.sub, 1, p2)];
                if (ci == kNone) continue;
                Combo& cb = combos[ci];
                cb.qt[s] = pr + c.off_a;
                cb.qoff[s] = c.x[1]->predecessor_index(p.b2, p2) + 1u + c.off_b;
                cb.q[s] = sparse ? 0 : (int32_t)(pr - c.x[1]->predecessor_index(p.b2, p2) + c.sw[0]->distance(p.b1, p1) - c.sw[1]->distance(p.b2, p2));
            }
        }
    }
    });
    }
    // the query of pair s in combination ci, whichever way it is kept
    auto query_at = [&](size_t ci, uint32_t s, uint32_t& qt, uint32_t& qoff, int32_t& q) {
        const Combo& c = combos[ci];
        if (!factored) { qt = c.qt[s]; qoff = c.qoff[s]; q = c.q[s]; return; }
        qt = fa_qt[(size_t)c.p1 * M + s];
        if (qt == kNone) { qoff = 0; q = 0; return; }
        qoff = fb_qoff[(size_t)c.p2 * M + s];
        q = sparse ? 0 : (int32_t)(fa_d[(size_t)c.p1 * M + s] - fb_d[(size_t)c.p2 * M + s]);
    };
    for (Combo& c : combos) {
        c.prefix.assign(n_blocks + 1, 0);
        size_t r = 0;
        for (uint32_t b = 0; b <= n_blocks; ++b) {
            const uint64_t first = (uint64_t)b * kChainBlock;
            while (r < c.rec_s.size() && c.rec_s[r] < first) ++r;
            c.prefix[b] = (uint32_t)r;
        }
    }

    lap("queries");
    // ---- device ----------------------------------------------------------------------------------------------------
    int rc = CL_OK;
    DevBuf<ClChainCombo> d_combos;
    DevBuf<float> d_weight, d_init, d_dp;
    DevBuf<uint32_t> d_rec_off, d_rec_combo, d_rec_pos, d_group, d_grp_base, d_grp_total, d_group_end, d_status;
    DevBuf<unsigned long long> d_xch, d_xdp, d_hacc;
    DevBuf<int> d_dp_enc;             // DPs over more combinations than the walk kernels take: the running dp maxima of the per-group path, order-preserving encoding
    DevBuf<char> d_pack;              // a small DP's arrays in one block (below)
    size_t pack_dp_off = 0, pack_acc_stride = 0, pack_down_bytes = 0;
    DevBuf<uint32_t> d_xred;
    DevBuf<int> d_row;                // the traceback's row of query results, when it fetches them step by step
    DevBuf<uint32_t> d_blk_rec, d_blk_q, d_blk_own, d_fa[4], d_ctags;   // factored queries: every combination's arrays as views into a few blocks
    DevBuf<float> d_blk_val;
    DevBuf<int> d_blk_acc;
    DevBuf<int> k_in, k_out;          // value index of the traceback (built on demand)
    DevBuf<uint32_t> i_in, i_out;
    DevBuf<char> vtemp;
    // branch-and-bound far pass (chain_far.hip)
    DevBuf<int> d_far_rec;
    DevBuf<uint32_t> d_far_base, d_far_u32[7], d_far_perm[2 * kFarMaxLevels], d_far_arena, d_far_tab, d_seal_items;
    DevBuf<char> d_far_temp;
    ClFarDevice F{};
    std::vector<uint32_t> group_first;   // first pair of every group + M (the per-group path of DPs over more combinations than the walk kernels take)
    std::vector<uint32_t> seal_off;   // [n_macro + 1] into d_seal_items
    std::vector<uint32_t> seal_big;   // [n_macro] how many of a macro-block's items (its first ones) are nodes of 4 096 records or more
    // the walk kernel (one workgroup per combination, all resident) replaces the per-block intra launches up to
    // kChainWalkMaxCombos combinations; CL_CHAIN_OLD_WALK=1 forces the per-block path (A/B measurements)
    static const bool old_walk_env = getenv("CL_CHAIN_OLD_WALK") != nullptr;
    // beyond kChainWalkMaxCombos a workgroup of the walk takes two or three combinations (chain_walk_fold_kernel: 768 combinations, a root of
    // 27 + 27 paths); CL_CHAIN_WALK_FOLD=2/3 folds smaller DPs too (tests)
    static const uint32_t fold_env = [] { const char* e = getenv("CL_CHAIN_WALK_FOLD"); const int v = e ? atoi(e) : 0; return v == 2 || v == 3 ? (uint32_t)v : 0u; }();
    const uint32_t walk_fold = std::max<uint32_t>(combos.size() > 1 ? fold_env : 0u, (uint32_t)((combos.size() + kChainWalkMaxCombos - 1) / kChainWalkMaxCombos));
    static const bool group_forced_env = [] { const char* e = getenv("CL_CHAIN_GROUP_PATH"); return e && e[0] == 'f'; }();
    const bool use_walk = allow_walk && walk_fold <= 3 && !old_walk_env && !group_forced_env;
    std::vector<ClChainCombo> hc(combos.size());
    auto cleanup = [&]() {
        cl_ctx_quiesce(ctx);   // once, for the ~40 blocks that go back to the pool below
        for (Combo& c : combos) c.release(true);
        d_combos.release(true); d_weight.release(true); d_init.release(true); d_dp.release(true); d_rec_off.release(true); d_rec_combo.release(true); d_rec_pos.release(true); d_group.release(true); d_grp_base.release(true); d_grp_total.release(true);
        d_dp_enc.release(true); d_group_end.release(true); d_status.release(true); d_xch.release(true); d_xdp.release(true); d_hacc.release(true); d_xred.release(true); d_pack.release(true);
        k_in.release(true); k_out.release(true); i_in.release(true); i_out.release(true); vtemp.release(true); d_row.release(true);
        d_blk_rec.release(true); d_blk_q.release(true); d_blk_own.release(true); d_blk_val.release(true); d_blk_acc.release(true); d_ctags.release(true);
        for (auto& b : d_fa) b.release(true);
        d_far_rec.release(true); d_far_base.release(true); d_seal_items.release(true); d_far_temp.release(true);
        for (auto& b : d_far_perm) b.release(true);
        for (auto& b : d_far_u32) b.release(true);
        d_far_arena.release(true); d_far_tab.release(true);
    };
#define CH(x) do { rc = (x); if (rc) { cleanup(); return rc; } } while (0)
    // A SMALL DP (the thousands of realignments of a polishing step hold a few hundred match pairs each: 113 copies and 126 stream waits per merge,
    // most of its time) packs every array below into ONE device block, filled on the host and sent with one copy; the DP values and every
    // combination's query results lie together at its end and come back with one copy.  Large DPs keep an allocation and a copy per array.
    size_t pack_total = 0;
    auto reserve = [&](size_t bytes) { const size_t at = pack_total; pack_total += (bytes + 255) & ~(size_t)255; return at; };
    struct PackOff { size_t rec_s, ins_t, off, sigma, prefix, qt, qoff, q, val, own, acc; };
    std::vector<PackOff> po(combos.size());
    for (size_t ci = 0; ci < combos.size(); ++ci) {
        const Combo& c = combos[ci];
        const size_t n = c.rec_s.size();
        po[ci] = PackOff{reserve(n * 4), reserve(n * 4), reserve(n * 4), reserve(n * 4), reserve(c.prefix.size() * 4), reserve(M * 4), reserve(M * 4), reserve(M * 4),
                         0, use_walk ? reserve(M * 4) : 0, 0};
    }
    const size_t o_combos = reserve(combos.size() * sizeof(ClChainCombo)), o_weight = reserve(M * 4), o_init = reserve(M * 4), o_rec_off = reserve((M + 1) * 4),
                 o_rec_combo = reserve(rec_combo.size() * 4), o_rec_pos = reserve(rec_pos.size() * 4);
    const size_t o_xch = use_walk ? reserve(combos.size() * kChainMacro * sizeof(unsigned long long)) : 0, o_status = use_walk ? reserve(32 * sizeof(uint32_t)) : 0;   // zeroed
    const size_t o_dp = reserve(M * 4);   // what comes back: the DP values, every combination's stored values (the traceback's value index) and query results
    for (size_t ci = 0; ci < combos.size(); ++ci) po[ci].val = reserve(7 * combos[ci].rec_s.size() * 4);
    for (size_t ci = 0; ci < combos.size(); ++ci) po[ci].acc = reserve(M * 7 * 4);
    const bool packed = pack_total <= (2u << 20) && !combos.empty();
    // the walk over several compute units per combination (chain_walk2.hip: a main workgroup with a window of 128 x qpt queries and helper
    // workgroups for the rest of the macro-block) wherever main + helpers of every combination fit the chip at one workgroup per compute unit;
    // CL_CHAIN_WALK2=0 keeps chain_walk_kernel (A/B), CL_CHAIN_WALK2_QPT=1/2 the window, CL_CHAIN_WALK2_HELPERS=n the helpers (0: the main
    // workgroup evaluates everything itself out of LDS — the path it takes when a helper is late)
    static const bool walk2_env = [] { const char* e = getenv("CL_CHAIN_WALK2"); return !e || e[0] != '0'; }();
    // (window: 2 x 1 Mbp affine / gap-free DP 157 / 90 ms with 128 queries, 165 / 92 ms with 256; 10 x 1 Mbp merges of 1 / 4 / 25 combinations, device time of
    // both DPs on one context: 165 / 285 / 1 158 ms here against 273 / 390 / 874 ms on chain_walk_kernel — at 25 combinations 200 workgroups of 1 024 threads
    // and 120 KB of LDS leave the far and near launches too little of the chip, so the one-workgroup walk stays above CL_CHAIN_WALK2_MAX combinations)
    static const uint32_t walk2_qpt = [] { const char* e = getenv("CL_CHAIN_WALK2_QPT"); return e && e[0] == '2' ? 2u : 1u; }();
    static const uint32_t walk2_max = [] { const char* e = getenv("CL_CHAIN_WALK2_MAX"); const int v = e ? atoi(e) : 0; return v > 0 ? (uint32_t)v : 8u; }();
    static const int walk2_help_env = [] { const char* e = getenv("CL_CHAIN_WALK2_HELPERS"); return e ? atoi(e) : -1; }();
    uint32_t walk2_help = cl_chain_walk2_helpers(walk2_qpt);
    const bool use_walk2 = use_walk && walk_fold <= 1 && !packed && walk2_env && combos.size() <= walk2_max && ((combos.size() + 7) & ~(size_t)7) * (1 + walk2_help) <= 256;
    if (walk2_help_env >= 0) walk2_help = (uint32_t)std::min(walk2_help_env, 7);
    if (packed) {
        CH(d_pack.alloc(ctx, pack_total));
        char* dev = d_pack.p;
        std::vector<char> stage(pack_total);
        auto put = [&](size_t at, const void* src, size_t bytes) { if (bytes) memcpy(stage.data() + at, src, bytes); };
        for (size_t ci = 0; ci < combos.size(); ++ci) {
            Combo& c = combos[ci];
            const size_t n = c.rec_s.size();
            const PackOff& o = po[ci];
            put(o.rec_s, c.rec_s.data(), n * 4); put(o.ins_t, c.ins_t.data(), n * 4); put(o.off, c.off.data(), n * 4); put(o.sigma, c.sigma.data(), n * 4);
            put(o.prefix, c.prefix.data(), c.prefix.size() * 4); put(o.qt, c.qt, M * 4); put(o.qoff, c.qoff, M * 4); put(o.q, c.q, M * 4);
            if (use_walk) memset(stage.data() + o.own, 0xFF, M * 4);
            int* a = reinterpret_cast<int*>(stage.data() + o.acc);
            for (size_t i = 0; i < (size_t)M * 7; ++i) a[i] = enc(CL_CHAIN_NEG);
            c.d_rec_s.view((uint32_t*)(dev + o.rec_s), n); c.d_ins_t.view((uint32_t*)(dev + o.ins_t), n); c.d_off.view((uint32_t*)(dev + o.off), n);
            c.d_sigma.view((int32_t*)(dev + o.sigma), n); c.d_prefix.view((uint32_t*)(dev + o.prefix), c.prefix.size());
            c.d_qt.view((uint32_t*)(dev + o.qt), M); c.d_qoff.view((uint32_t*)(dev + o.qoff), M); c.d_q.view((int32_t*)(dev + o.q), M);
            c.d_val.view((float*)(dev + o.val), 7 * n); c.d_acc.view((int*)(dev + o.acc), (size_t)M * 7);
            if (use_walk) c.d_own_rec.view((uint32_t*)(dev + o.own), M);
            hc[ci] = ClChainCombo{(uint32_t)n, c.d_rec_s.p, c.d_ins_t.p, c.d_off.p, c.d_sigma.p, c.d_val.p, c.d_prefix.p,
                                  c.d_qt.p, c.d_qoff.p, c.d_q.p, c.d_acc.p, c.d_own_rec.p};
        }
        put(o_combos, hc.data(), hc.size() * sizeof(ClChainCombo)); put(o_weight, weight.data(), M * 4); put(o_init, init_w.data(), M * 4);
        put(o_rec_off, rec_off.data(), (M + 1) * 4); put(o_rec_combo, rec_combo.data(), rec_combo.size() * 4); put(o_rec_pos, rec_pos.data(), rec_pos.size() * 4);
        d_combos.view((ClChainCombo*)(dev + o_combos), hc.size()); d_weight.view((float*)(dev + o_weight), M); d_init.view((float*)(dev + o_init), M);
        d_dp.view((float*)(dev + o_dp), M);
        if (use_walk) { d_xch.view((unsigned long long*)(dev + o_xch), combos.size() * kChainMacro); d_status.view((uint32_t*)(dev + o_status), 32); }
        d_rec_off.view((uint32_t*)(dev + o_rec_off), M + 1); d_rec_combo.view((uint32_t*)(dev + o_rec_combo), rec_combo.size()); d_rec_pos.view((uint32_t*)(dev + o_rec_pos), rec_pos.size());
        pack_dp_off = o_dp;
        pack_acc_stride = combos.size() > 1 ? po[1].acc - po[0].acc : (((size_t)M * 7 * 4 + 255) & ~(size_t)255);
        pack_down_bytes = pack_total - o_dp;
        if (hipMemcpyAsync(dev, stage.data(), pack_total, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            cleanup(); cl_set_error(ctx, "upload failed"); return CL_ERR_HIP;
        }
    } else if (factored) {
        // a few blocks instead of eleven allocations and eight copies per combination (625 combinations: 7 000 allocations, 5 000 copies)
        const size_t C = combos.size();
        std::vector<size_t> rbase(C + 1, 0);
        for (size_t ci = 0; ci < C; ++ci) rbase[ci + 1] = rbase[ci] + combos[ci].rec_s.size();
        const size_t R_all = rbase[C], pw = (size_t)n_blocks + 1;
        ClRawVec<uint32_t> stage(4 * R_all + C * pw);
        cl_parallel_for(C, [&](uint64_t c_begin, uint64_t c_end) {
            for (uint64_t ci = c_begin; ci < c_end; ++ci) {
                const Combo& c = combos[ci];
                const size_t n = c.rec_s.size();
                if (n) {
                    memcpy(&stage[rbase[ci]], c.rec_s.data(), n * 4); memcpy(&stage[R_all + rbase[ci]], c.ins_t.data(), n * 4);
                    memcpy(&stage[2 * R_all + rbase[ci]], c.off.data(), n * 4); memcpy(&stage[3 * R_all + rbase[ci]], c.sigma.data(), n * 4);
                }
                memcpy(&stage[4 * R_all + ci * pw], c.prefix.data(), pw * 4);
            }
        }, 1);
        std::vector<uint32_t> ctags(2 * C);
        for (size_t ci = 0; ci < C; ++ci) { ctags[2 * ci] = combos[ci].p1; ctags[2 * ci + 1] = combos[ci].p2; }
        CH(d_blk_rec.upload_async(ctx, stage));
        CH(d_ctags.upload_async(ctx, ctags));
        CH(d_fa[0].upload_async(ctx, fa_qt)); CH(d_fa[1].upload_async(ctx, fb_qoff));
        if (!sparse) { CH(d_fa[2].upload_async(ctx, fa_d)); CH(d_fa[3].upload_async(ctx, fb_d)); }
        CH(d_blk_val.alloc(ctx, 7 * R_all));
        CH(d_blk_q.alloc(ctx, 3 * C * (size_t)M));
        CH(d_blk_acc.alloc(ctx, C * (size_t)M * 7));
        CH(d_blk_own.alloc(ctx, C * (size_t)M));
        if (hipMemsetD32Async((hipDeviceptr_t)d_blk_acc.p, enc(CL_CHAIN_NEG), C * (size_t)M * 7, ctx->stream) != hipSuccess ||
            hipMemsetAsync(d_blk_own.p, 0xFF, C * (size_t)M * sizeof(uint32_t), ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipMemsetAsync failed"); return CL_ERR_HIP; }
        if (cl_chain_expand_queries(d_fa[0].p, d_fa[2].p, d_fa[1].p, d_fa[3].p, d_ctags.p, (uint32_t)C, (uint32_t)M, d_blk_q.p, d_blk_q.p + C * (size_t)M,
                                    (int32_t*)(d_blk_q.p + 2 * C * (size_t)M), ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "query expansion failed"); return CL_ERR_HIP; }
        for (size_t ci = 0; ci < C; ++ci) {
            Combo& c = combos[ci];
            const size_t n = c.rec_s.size();
            c.d_rec_s.view(d_blk_rec.p + rbase[ci], n); c.d_ins_t.view(d_blk_rec.p + R_all + rbase[ci], n); c.d_off.view(d_blk_rec.p + 2 * R_all + rbase[ci], n);
            c.d_sigma.view((int32_t*)(d_blk_rec.p + 3 * R_all + rbase[ci]), n); c.d_prefix.view(d_blk_rec.p + 4 * R_all + ci * pw, pw);
            c.d_qt.view(d_blk_q.p + ci * (size_t)M, M); c.d_qoff.view(d_blk_q.p + (C + ci) * (size_t)M, M); c.d_q.view((int32_t*)(d_blk_q.p + (2 * C + ci) * (size_t)M), M);
            c.d_val.view(d_blk_val.p + 7 * rbase[ci], 7 * n); c.d_acc.view(d_blk_acc.p + ci * (size_t)M * 7, (size_t)M * 7);
            if (use_walk) c.d_own_rec.view(d_blk_own.p + ci * (size_t)M, M);
            hc[ci] = ClChainCombo{(uint32_t)n, c.d_rec_s.p, c.d_ins_t.p, c.d_off.p, c.d_sigma.p, c.d_val.p, c.d_prefix.p,
                                  c.d_qt.p, c.d_qoff.p, c.d_q.p, c.d_acc.p, c.d_own_rec.p};
        }
        CH(d_combos.upload_async(ctx, hc));
        CH(d_weight.upload_async(ctx, weight));
        CH(d_init.upload_async(ctx, init_w));
        CH(d_dp.alloc(ctx, M));
        CH(d_rec_off.upload_async(ctx, rec_off)); CH(d_rec_combo.upload_async(ctx, rec_combo)); CH(d_rec_pos.upload_async(ctx, rec_pos));
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "upload failed"); return CL_ERR_HIP; }
    } else {
    for (size_t ci = 0; ci < combos.size(); ++ci) {
        Combo& c = combos[ci];
        // (enqueued only: one wait for all of them below — a merge of small graphs used to spend most of its time in these round trips)
        CH(c.d_rec_s.upload_async(ctx, c.rec_s)); CH(c.d_ins_t.upload_async(ctx, c.ins_t)); CH(c.d_off.upload_async(ctx, c.off));
        CH(c.d_sigma.upload_async(ctx, c.sigma)); CH(c.d_prefix.upload_async(ctx, c.prefix));
        struct Slice { const void* p; size_t n; const void* data() const { return p; } size_t size() const { return n; } bool empty() const { return n == 0; } };
        CH(c.d_qt.upload_async(ctx, Slice{c.qt, (size_t)M})); CH(c.d_qoff.upload_async(ctx, Slice{c.qoff, (size_t)M})); CH(c.d_q.upload_async(ctx, Slice{c.q, (size_t)M}));
        CH(c.d_val.alloc(ctx, 7 * c.rec_s.size()));
        CH(c.d_acc.alloc(ctx, M * 7));
        if (hipMemsetD32Async((hipDeviceptr_t)c.d_acc.p, enc(CL_CHAIN_NEG), M * 7, ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipMemsetD32Async failed"); return CL_ERR_HIP; }
        if (use_walk) {
            CH(c.d_own_rec.alloc(ctx, M));
            if (hipMemsetAsync(c.d_own_rec.p, 0xFF, M * sizeof(uint32_t), ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipMemsetAsync failed"); return CL_ERR_HIP; }
        }
        hc[ci] = ClChainCombo{(uint32_t)c.rec_s.size(), c.d_rec_s.p, c.d_ins_t.p, c.d_off.p, c.d_sigma.p, c.d_val.p, c.d_prefix.p,
                              c.d_qt.p, c.d_qoff.p, c.d_q.p, c.d_acc.p, c.d_own_rec.p};
    }
    CH(d_combos.upload_async(ctx, hc));
    CH(d_weight.upload_async(ctx, weight));
    CH(d_init.upload_async(ctx, init_w));
    CH(d_dp.alloc(ctx, M));
    CH(d_rec_off.upload_async(ctx, rec_off)); CH(d_rec_combo.upload_async(ctx, rec_combo)); CH(d_rec_pos.upload_async(ctx, rec_pos));
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "upload failed"); return CL_ERR_HIP; }
    }
    {
        // groups = maximal runs (in depth order) of pairs none of which can precede another: a predecessor m of m' ends
        // strictly before m' starts, so depth(b1(m')) >= depth(b1(m)) + len(m); a run is closed as soon as a pair starts at or
        // beyond the smallest depth(b1) + len of the run.  The device finalises a whole group at once.
        std::vector<uint32_t> group(M);
        std::vector<std::vector<uint32_t>> set_len(K);
        group_first.clear();
        for (size_t k = 0; k < K; ++k) {
            if (sc[k].pair_hi == sc[k].pair_lo) continue;
            const SetView ms(subs[k]);
            set_len[k].resize(subs[k].num_match_sets);
            for (uint64_t st = 0; st < subs[k].num_match_sets; ++st)
                if (ms.n_walks(0, st)) set_len[k][st] = (uint32_t)ms.length(st);
        }
        uint32_t gid = 0;
        uint64_t min_end = 0;
        for (uint32_t s = 0; s < M; ++s) {
            const Pair& p = pairs[by_s[s]];
            const uint64_t start = sc[p.sub].depth1[p.b1], end = start + set_len[p.sub][p.set];
            if (s == 0) min_end = end;
            else if (start >= min_end) { ++gid; min_end = end; }
            else min_end = std::min(min_end, end);
            group[s] = gid;
            if (s == 0 || group[s] != group[s - 1]) group_first.push_back(s);
        }
        group_first.push_back((uint32_t)M);
        if (timing) {   // groups per block of kChainBlock pairs (each costs the sequential kernel two barriers + one LDS broadcast)
            std::vector<uint32_t> hist;
            for (uint64_t b0 = 0; b0 < M; b0 += kChainBlock) {
                const uint64_t b1 = std::min<uint64_t>(M, b0 + kChainBlock);
                const uint32_t g = group[b1 - 1] - group[b0] + 1;
                if (hist.size() <= g) hist.resize(g + 1, 0);
                ++hist[g];
            }
            fprintf(stderr, "[chain_dp_batch]   %u groups; groups per block:", gid + 1);
            for (size_t g = 0; g < hist.size(); ++g) if (hist[g]) fprintf(stderr, " %zu:%u", g, hist[g]);
            fprintf(stderr, "\n");
        }
        CH(d_group.upload_async(ctx, group));       // (these four copies are waited for together at the end of the block)
        std::vector<uint32_t> group_end;
        if (use_walk) {
            group_end.resize(M);
            for (uint64_t s1 = M; s1 > 0;) {
                uint64_t s0 = s1 - 1;
                while (s0 > 0 && group[s0 - 1] == group[s1 - 1]) --s0;
                for (uint64_t s = s0; s < s1; ++s) group_end[s] = (uint32_t)s1;
                s1 = s0;
            }
            CH(d_group_end.upload_async(ctx, group_end));
            if (use_walk2) {
                CH(d_xdp.alloc(ctx, combos.size() * kChainMacro));
                CH(d_hacc.alloc(ctx, combos.size() * kChainMacro * 8));
                if (hipMemsetAsync(d_xdp.p, 0, combos.size() * kChainMacro * sizeof(unsigned long long), ctx->stream) != hipSuccess ||
                    hipMemsetAsync(d_hacc.p, 0, combos.size() * kChainMacro * 8 * sizeof(unsigned long long), ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipMemsetAsync failed"); return CL_ERR_HIP; }
            }
            if (!d_pack.p) {   // (a small DP's pack holds them, zeroed)
                CH(d_xch.alloc(ctx, combos.size() * kChainMacro));
                CH(d_status.alloc(ctx, 32));
                if (hipMemsetAsync(d_xch.p, 0, combos.size() * kChainMacro * sizeof(unsigned long long), ctx->stream) != hipSuccess ||
                    hipMemsetAsync(d_status.p, 0, 32 * sizeof(uint32_t), ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipMemsetAsync failed"); return CL_ERR_HIP; }
            }
            // the exchange between the walk's workgroups: granule sweep for few combinations, reduction for many (CL_CHAIN_WALK_REDUCE=0/1 pins it: A/B)
            static const char* reduce_env = getenv("CL_CHAIN_WALK_REDUCE");
            const bool reduce = walk_fold > 1 || (reduce_env ? reduce_env[0] == '1' : combos.size() > kChainWalkSweepCombos);
            if (reduce && combos.size() > 1) {
                CH(d_xred.alloc(ctx, 2 * M));
                if (hipMemsetAsync(d_xred.p, 0, 2 * M * sizeof(uint32_t), ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipMemsetAsync failed"); return CL_ERR_HIP; }
            }
        }
        // LDS slots of the sequential kernel: records of a (block, group) are laid out in pair order
        std::vector<uint32_t> grp_base(M), grp_total(M);
        for (uint64_t s0 = 0; s0 < M;) {
            const uint64_t block_end = std::min<uint64_t>(M, (s0 / kChainBlock + 1) * kChainBlock);
            uint64_t s1 = s0;
            while (s1 < block_end && group[s1] == group[s0]) ++s1;
            const uint32_t total = rec_off[s1] - rec_off[s0];
            for (uint64_t s = s0; s < s1; ++s) { grp_base[s] = rec_off[s] - rec_off[s0]; grp_total[s] = total; }
            s0 = s1;
        }
        CH(d_grp_base.upload_async(ctx, grp_base));
        CH(d_grp_total.upload_async(ctx, grp_total));
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { cleanup(); cl_set_error(ctx, "upload failed"); return CL_ERR_HIP; }
    }
    // more combinations than the walk kernels take: group by group, parallel over the combinations (chain_kernels.hip: chain_group_*; CL_CHAIN_GROUP_PATH=0: the one-workgroup
    // per-block kernels of rounds 1-5, A/B)
    static const bool group_path_env = [] { const char* e = getenv("CL_CHAIN_GROUP_PATH"); return !e || e[0] != '0'; }();
    static const bool group_path_forced = [] { const char* e = getenv("CL_CHAIN_GROUP_PATH"); return e && e[0] == 'f'; }();   // =force: every DP, whatever its combinations (tests)
    const bool group_path = group_path_forced || (group_path_env && walk_fold > 3 && !old_walk_env);
    if (group_path) {
        std::vector<int> dp0(M);
        for (uint64_t s2 = 0; s2 < M; ++s2) dp0[s2] = enc(init_w[s2]);
        CH(d_dp_enc.upload(ctx, dp0));
    }
    ClChainDevice D{};
    D.n_pairs = (uint32_t)M;
    D.n_combos = (uint32_t)combos.size();
    D.combos = d_combos.p;
    D.weight = d_weight.p;
    D.init = d_init.p;
    D.dp = d_dp.p;
    D.rec_off = d_rec_off.p;
    D.rec_combo = d_rec_combo.p;
    D.rec_pos = d_rec_pos.p;
    D.group = d_group.p;
    D.grp_base = d_grp_base.p;
    D.grp_total = d_grp_total.p;
    for (int i = 0; i < 3; ++i) { D.params.gap_open[i] = cp->gap_open[i]; D.params.gap_extend[i] = cp->gap_extend[i]; }
    D.params.scale = local_scale;
    D.sparse = sparse ? 1u : 0u;
    D.group_end = d_group_end.p;
    D.xch = d_xch.p;
    D.xred = d_xred.p;
    D.xdp = d_xdp.p;
    D.hacc = d_hacc.p;
    static const bool walk2_debug = getenv("CL_CHAIN_WALK2_DEBUG") != nullptr;   // in-kernel counters and clocks of the walk (status[8..]), printed with CL_CHAIN_TIMING
    D.debug = use_walk2 && walk2_debug ? 1u : 0u;
    D.status = d_status.p;


    // ---- branch-and-bound far pass: record image, the two static orders per level, the sealing schedule -------------------
    static const bool no_far_env = getenv("CL_CHAIN_NO_FAR_PRUNE") != nullptr;
    const uint32_t n_macro_all = (uint32_t)((M + kChainMacro - 1) / kChainMacro);
    // far launches in flight: 4 by default in the affine DP.  Measured: the 2 x 1 Mbp affine DP's device time is 1045 / 540 / 393 / 372 / 340 ms at 1 / 2 / 3 /
    // 4 / 6 in flight (a far launch is 1024 queries, too few to fill the chip alone); the 10 x 1 Mbp MSA with four worker contexts is
    // within 3 % for 2, 3 and 4 and 5 % slower at 6 (the near pass grows with the lag).  CL_CHAIN_FAR_LAG=1..8 for measurements
    // The gap-free DP's far pass opens a tenth of the leaves, so its near pass is what grows: 175 ms at 2 in flight, 253 ms at 4 — it keeps 2
    static const uint32_t far_lag_env = [] { const char* e = getenv("CL_CHAIN_FAR_LAG"); int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= (int)kFarLag ? v : 0); }();
    const uint32_t far_lag = far_lag_env ? far_lag_env : 2u;   // round 3 (8-ary search trees, leaves scanned by eight lanes): 2 x 1 Mbp affine 313 / 396 / 341 / 393 ms at 2 / 3 / 4 / 6; root of 10 x 1 Mbp 1.22 s at 2, 1.39 s at 4
    bool use_far = use_walk && !no_far_env && n_macro_all >= far_lag + 3;
    if (use_far) {
        uint32_t max_n = 0;
        int64_t smin = INT64_MAX, smax = INT64_MIN;
        uint32_t max_off = 0;
        {   // extremes of the shifts and offsets: one pass over every record (27 M at the root of ten sequences), combination by combination on the pool's threads
            std::vector<int64_t> lo(combos.size(), INT64_MAX), hi(combos.size(), INT64_MIN);
            std::vector<uint32_t> mo(combos.size(), 0);
            cl_parallel_for(combos.size(), [&](uint64_t b, uint64_t e) {
                for (uint64_t ci = b; ci < e; ++ci) {
                    const Combo& c = combos[ci];
                    int32_t l = INT32_MAX, h = INT32_MIN;
                    uint32_t m = 0;
                    const size_t n = c.rec_s.size();
                    for (size_t r = 0; r < n; ++r) { const int32_t sg = c.sigma[r]; l = std::min(l, sg); h = std::max(h, sg); m = std::max(m, c.off[r]); }
                    if (n) { lo[ci] = l; hi[ci] = h; mo[ci] = m; }
                }
            }, 1);
            for (size_t ci = 0; ci < combos.size(); ++ci) {
                max_n = std::max<uint32_t>(max_n, (uint32_t)combos[ci].rec_s.size());
                smin = std::min(smin, lo[ci]); smax = std::max(smax, hi[ci]); max_off = std::max(max_off, mo[ci]);
            }
        }
        uint32_t n_levels = 1;
        while (n_levels < (uint32_t)kFarMaxLevels && (64ull << (kFarFanShift * n_levels)) <= max_n) ++n_levels;
        const uint64_t n_top = 64ull << (kFarFanShift * (n_levels - 1));
        std::vector<uint32_t> base(combos.size());
        uint64_t r_pad = 0, max_padded = 0;
        for (size_t ci = 0; ci < combos.size(); ++ci) {
            base[ci] = (uint32_t)r_pad;
            const uint64_t padded = (combos[ci].rec_s.size() + n_top - 1) / n_top * n_top;
            max_padded = std::max(max_padded, padded);
            r_pad += padded;
        }
        // shift buckets: kFarBandShift wide (CL_CHAIN_FAR_BAND for measurements) unless bucket number and offset then do not fit the 32 bits of a
        // bucket key: bucket << off_bits | offset, off_bits enough for the largest offset + 1 (what a query's offset bound can be)
        static const int band_env = [] { const char* e = getenv("CL_CHAIN_FAR_BAND"); int v = e ? atoi(e) : 0; return v >= 4 && v <= 24 ? v : kFarBandShift; }();
        uint32_t band_shift = (uint32_t)band_env;
        uint32_t off_bits = 1;
        while (off_bits < 31 && ((uint64_t)max_off + 1) >> off_bits) ++off_bits;
        // buckets are numbered from 1; bucket + 2 must stay below 2^(32 - off_bits) - 1 (the query's upper neighbour bucket, and the padding key above all)
        while (smin != INT64_MAX && band_shift < 30 && (((smax - smin) >> band_shift) + 4) >= (int64_t)((1ull << (32 - off_bits)) - 1)) ++band_shift;
        const int64_t bias = (1ll << band_shift) - smin;
        if (r_pad == 0 || r_pad >= (1ull << 31) || smin == INT64_MAX || smax + bias >= (1ll << 31) || bias >= (1ll << 31) || off_bits >= 28 ||
            (((smax - smin) >> band_shift) + 4) >= (int64_t)((1ull << (32 - off_bits)) - 1)) use_far = false;
        if (use_far) {
            const uint32_t R = (uint32_t)r_pad;
            F.n_levels = n_levels;
            F.r_pad = R;
            F.sig_bias = (int32_t)bias;
            F.band_shift = band_shift;
            F.off_bits = off_bits;
            double pw = 1e300, omax = 0, emax = 0;
            for (int k = 0; k < 3; ++k) {
                pw = std::min(pw, local_scale * (cp->gap_open[k] + cp->gap_extend[k] * (double)(1ull << band_shift)));
                omax = std::max(omax, cp->gap_open[k]);
                emax = std::max(emax, cp->gap_extend[k]);
            }
            F.band_pen = sparse ? 0.0 : pw;
            F.slack_t0 = local_scale * emax * (double)std::max<int64_t>(std::llabs(smin), std::llabs(smax)) + local_scale * omax;
            F.slack_e0 = local_scale * emax;
            // (the record image — 48 B per padded record — is allocated below, once the arena is known to fit 32-bit offsets: at the root of a 50-sequence MSA
            // it is 33 GB, and round 4 allocated it BEFORE finding out that the far pass could not be used there: the context held 105 GB for a DP of 71)
            CH(d_far_base.upload(ctx, base));
            D.far_base = d_far_base.p;
            // the arena: per level the two blocked orders (2 R words each) and their index arrays (R / 8, R / 64, ...)
            // Table entries are positions in blocks of 8 words (far_at, chain_far.hip): 2^35 words.  CL_CHAIN_FAR_ARENA_SKIP=<words> leaves that many words unused in
            // front (tests: positions beyond 2^32 words on a small DP; the pages are never touched)
            static const uint64_t skip_env = [] { const char* e = getenv("CL_CHAIN_FAR_ARENA_SKIP"); return e ? (uint64_t)strtoull(e, nullptr, 10) & ~7ull : 0ull; }();
            uint64_t words = skip_env;
            std::vector<uint32_t> tab((size_t)kFarMaxLevels * kFarTabWidth, 0xFFFFFFFFu);
            for (uint32_t l = 0; l < n_levels; ++l) {
                for (int side = 0; side < (sparse ? 1 : 2); ++side) {
                    tab[l * kFarTabWidth + side] = (uint32_t)(words >> 3);
                    words += 2ull * R;
                    for (uint32_t j = 0; j <= l; ++j) {
                        tab[l * kFarTabWidth + 2 + side * kFarMaxLevels + j] = (uint32_t)(words >> 3);
                        words += ((uint64_t)R >> (3 * (j + 1))) + 8;
                        words = (words + 7) & ~7ull;    // 32-byte blocks stay aligned
                    }
                }
            }
            if (words >= (1ull << 35) - 8) use_far = false;
            if (use_far && (uint64_t)R >= (1ull << 27)) {
                // a far pass this large (the root of a 50-sequence MSA: 690 M padded records, 185 B each with the sorts' scratch) is taken only if the device has
                // the room now — the DP is correct without it, an allocation that fails half-way would end the merge
                size_t free_b = 0, total_b = 0;
                const uint64_t need = (uint64_t)R * (48 + 7 * 4 + 8 * 2 + n_levels * (sparse ? 1 : 2) * 4) + words * 4;
                // ... and only if the context's last affine DP did not end up sweeping all pairs anyway: on 50 x 100 kbp (dense matches: a third of the leaves opened)
                // the DPs of the 156-combination merges choose the sweep, the 625-combination root then does too, and the sweep inside the far pass is 15 % slower than
                // without its structures (163 against 142 s); on 50 x 1 Mbp they choose branch-and-bound (1.3 % opened) and the root takes 36 instead of 196 s
                if (!sparse && ctx->far_last_choice == 2) {
                    if (timing) fprintf(stderr, "[chain_dp_batch]   far pass not taken: the context's last affine DP chose the all-pairs sweep\n");
                    use_far = false;
                } else
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need + need / 8 + (8ull << 30) > (uint64_t)free_b) {
                    if (timing) fprintf(stderr, "[chain_dp_batch]   far pass not taken: it needs %.1f GB, the device has %.1f GB free\n", need * 1.125 / 1e9, free_b / 1e9);
                    use_far = false;
                }
            }
            if (use_far) {
            CH(d_far_rec.alloc(ctx, (size_t)R * 12));
            D.far_rec = d_far_rec.p;
            CH(d_far_arena.alloc(ctx, words));
            CH(d_far_tab.upload(ctx, tab));
            F.arena = d_far_arena.p;
            F.tab_dev = d_far_tab.p;
            for (uint32_t l = 0; l < (uint32_t)kFarMaxLevels; ++l) for (int w = 0; w < kFarTabWidth; ++w) F.tab[l][w] = tab[l * kFarTabWidth + w];
            // scratch: 0 key_off, 1 idx, 2 order_off, 3 order_band, 4 node keys in, 5 node keys out / sorted-key sink, 6 key_band
            for (int i = 0; i < 7; ++i) CH(d_far_u32[i].alloc(ctx, R));
            const size_t temp_bytes = cl_chain_far_sort_temp_bytes(R);
            CH(d_far_temp.alloc(ctx, temp_bytes));
            hipError_t fe = cl_chain_far_init(D, d_far_base.p, (uint32_t)max_padded, R, F.sig_bias, F.band_shift, F.off_bits, d_far_u32[0].p, d_far_u32[6].p, d_far_u32[1].p, ctx->stream);
            if (fe == hipSuccess) fe = cl_chain_far_sort32(d_far_temp.p, temp_bytes, d_far_u32[0].p, d_far_u32[5].p, d_far_u32[1].p, d_far_u32[2].p, R, 32, ctx->stream);
            if (fe == hipSuccess && !sparse) fe = cl_chain_far_sort32(d_far_temp.p, temp_bytes, d_far_u32[6].p, d_far_u32[5].p, d_far_u32[1].p, d_far_u32[3].p, R, 32, ctx->stream);
            for (uint32_t l = 0; l < n_levels && fe == hipSuccess; ++l) {
                const uint32_t shift = kFarLeafShift + kFarFanShift * l;
                int bits = 1;
                while (((uint64_t)R >> shift) >> bits) ++bits;
                DevBuf<uint32_t>& perm_o = d_far_perm[2 * l];
                DevBuf<uint32_t>& perm_b = d_far_perm[2 * l + 1];
                CH(perm_o.alloc(ctx, R));
                fe = cl_chain_far_node_keys(d_far_u32[2].p, R, shift, d_far_u32[4].p, ctx->stream);
                if (fe == hipSuccess) fe = cl_chain_far_sort32(d_far_temp.p, temp_bytes, d_far_u32[4].p, d_far_u32[5].p, d_far_u32[2].p, perm_o.p, R, bits, ctx->stream);
                if (fe == hipSuccess) fe = cl_chain_far_layout(perm_o.p, d_far_u32[0].p, R, F.arena, F.tab[l][0], &F.tab[l][2], l + 1, ctx->stream);
                F.perm_o[l] = perm_o.p;
                if (!sparse) {
                    CH(perm_b.alloc(ctx, R));
                    if (fe == hipSuccess) fe = cl_chain_far_node_keys(d_far_u32[3].p, R, shift, d_far_u32[4].p, ctx->stream);
                    if (fe == hipSuccess) fe = cl_chain_far_sort32(d_far_temp.p, temp_bytes, d_far_u32[4].p, d_far_u32[5].p, d_far_u32[3].p, perm_b.p, R, bits, ctx->stream);
                    if (fe == hipSuccess) fe = cl_chain_far_layout(perm_b.p, d_far_u32[6].p, R, F.arena, F.tab[l][1], &F.tab[l][2 + kFarMaxLevels], l + 1, ctx->stream);
                    F.perm_b[l] = perm_b.p;
                } else {   // sparse_chain_dp has no shifts: one order
                    F.perm_b[l] = perm_o.p;
                }
            }
            if (fe != hipSuccess) { cl_set_error(ctx, "far pass setup failed: %s", hipGetErrorString(fe)); cleanup(); return CL_ERR_HIP; }
            // sealing schedule: a node is sealed right after the walk of the macro-block that finalises its last record
            std::vector<std::vector<uint32_t>> by_macro(n_macro_all);
            for (size_t ci = 0; ci < combos.size(); ++ci) {
                const Combo& c = combos[ci];
                const uint64_t nc = c.rec_s.size();
                for (uint32_t l = 0; l < n_levels; ++l) {
                    const uint32_t shift = kFarLeafShift + kFarFanShift * l;
                    const uint64_t nn = 1ull << shift;
                    for (uint64_t a = 0; a < nc; a += nn) {
                        const uint64_t last = std::min(a + nn, nc) - 1;
                        by_macro[c.rec_s[last] / kChainMacro].push_back((l << 28) | (uint32_t)((base[ci] + a) >> shift));
                    }
                }
            }
            std::vector<uint32_t> items;
            seal_off.assign(n_macro_all + 1, 0);
            static const bool seal_wave_only = getenv("CL_CHAIN_SEAL_WAVE") != nullptr;   // A/B: every node by one wave (rounds 2-3)
            seal_big.assign(n_macro_all, 0);
            for (uint32_t k = 0; k < n_macro_all; ++k) {
                // large nodes first: they get a workgroup each (far_seal_big_kernel), the others a wave each
                std::stable_sort(by_macro[k].begin(), by_macro[k].end(), [](uint32_t a, uint32_t b) { return (a >> 28) > (b >> 28); });
                if (!seal_wave_only) for (uint32_t it : by_macro[k]) seal_big[k] += (it >> 28) >= 2 ? 1u : 0u;
                items.insert(items.end(), by_macro[k].begin(), by_macro[k].end());
                seal_off[k + 1] = (uint32_t)items.size();
            }
            CH(d_seal_items.upload(ctx, items));
            for (int i = 0; i < 7; ++i) d_far_u32[i].release();   // scratch
            }
        }
    }
    lap("far pass setup");
    tm.prep_ms += ms_since(T0);
    lap("upload");
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) { cleanup(); cl_set_error(ctx, "hipEventCreate failed"); return CL_ERR_HIP; }
    auto hip_fail = [&](hipError_t e, const char* what) {
        cl_set_error(ctx, "%s failed: %s", what, hipGetErrorString(e));
        (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
        cleanup();
        return e == hipErrorOutOfMemory ? CL_ERR_OUT_OF_MEMORY : CL_ERR_HIP;
    };
    // Two streams.  The context's stream is the sequential one: per block b the one-workgroup walk of the block, preceded by
    // the "near" pass over the blocks the far launch could not see yet (folded into the walk kernel when there is a single
    // combination).  The bulk is evaluated on an auxiliary stream ("far"): one launch serves a GROUP of kChainFarGroup
    // consecutive blocks (a far launch lasts at least one tile, ~70 us, however few predecessors there are, and the far
    // stream sets the pace: fewer, fatter launches) and reads the blocks before g0 - 1, final once walk(g0 - 2) is done.
    static const uint32_t far_group = getenv("CL_CHAIN_FAR_GROUP") ? std::max(1, atoi(getenv("CL_CHAIN_FAR_GROUP"))) : kChainFarGroup;
    // consecutive far launches alternate between auxiliary streams: while their grids are small their latency floors overlap
    static const uint32_t far_streams = getenv("CL_CHAIN_FAR_STREAMS") ? std::min(4, std::max(1, atoi(getenv("CL_CHAIN_FAR_STREAMS")))) : kChainFarStreams;
    std::vector<hipEvent_t> ev_intra(n_blocks, nullptr), ev_far(n_blocks, nullptr);
    // what the traceback's tie resolution needs of every combination — its records in (shift, slot) order with the implicit-heap layout of the
    // reference's outer tree — depends on the records alone, not on the DP: built on the pool's threads while the device runs the DP (at the
    // root of a ten-sequence tree the traceback spent 70 of its 166 ms building them on first use)
    std::unordered_map<uint64_t, GapFreeTree> sparse_trees;   // sparse mode: (chain2 tag, instance) -> tree
    auto finish_tree = [&](GapFreeTree* tree, bool by_slot) {    // members -> (offset, slot) order + the implicit heap layout
        {
            std::vector<uint32_t> idx(tree->mem.size());
            std::iota(idx.begin(), idx.end(), 0u);
            const auto& m0 = tree->mem;
            sort_by_key_then_slot(idx, M, [&](uint32_t i) { return (int64_t)m0[i].off; }, [&](uint32_t i) { return m0[i].slot; }, by_slot);
            std::vector<GapFreeTree::Member> sorted(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) sorted[i] = m0[idx[i]];
            tree->mem.swap(sorted);
        }
        tree->heap = heap_of_rank(tree->mem.size());
        tree->rank_of_heap.resize(tree->mem.size());
        for (size_t r = 0; r < tree->mem.size(); ++r) tree->rank_of_heap[tree->heap[r]] = (uint32_t)r;
        tree->built = true;
    };
    std::thread ortho_prebuild;
    struct ThreadJoiner { std::thread& t; ~ThreadJoiner() { if (t.joinable()) t.join(); } } ortho_joiner{ortho_prebuild};
    if (sparse && K == 1 && M >= (1u << 16)) {
        // the gap-free DP's trees — one per chain of graph 2, holding every record filed under it, keyed (offset, slot) — depend on the records alone as well: a tie on a
        // tree of 1.25 M members cost the traceback 19 ms on first use (a leaf merge of 10 x 1 Mbp); built here beside the device's DP
        std::vector<uint32_t> p2s;
        for (const Combo& c : combos) if (std::find(p2s.begin(), p2s.end(), c.p2) == p2s.end()) p2s.push_back(c.p2);
        for (uint32_t p2 : p2s) sparse_trees[((uint64_t)p2 << 32) | 0u];   // (the map is complete before the thread starts: no insertion races with the traceback)
        ortho_prebuild = std::thread([&, p2s] {
            cl_parallel_for(p2s.size(), [&](uint64_t b, uint64_t e) {
                for (uint64_t i = b; i < e; ++i) {
                    GapFreeTree* tree = &sparse_trees.find(((uint64_t)p2s[i] << 32) | 0u)->second;
                    for (Combo& oc : combos)
                        if (oc.p2 == p2s[i])
                            for (uint32_t r = 0; r < oc.rec_s.size(); ++r) tree->mem.push_back(GapFreeTree::Member{oc.off[r], by_s[oc.rec_s[r]], r});
                    finish_tree(tree, false);
                }
            }, 1);
        });
    }
    if (!sparse && K == 1 && M >= (1u << 16))
        ortho_prebuild = std::thread([&] {
            cl_parallel_for(combos.size(), [&](uint64_t b, uint64_t e) {
                for (uint64_t ci = b; ci < e; ++ci) {
                    Combo& c = combos[ci];
                    Combo::SubRecs& sr = c.per_sub[0];
                    sr.recs.resize(c.rec_s.size());
                    std::iota(sr.recs.begin(), sr.recs.end(), 0u);
                    sr.ortho_order = sr.recs;
                    sort_by_key_then_slot(sr.ortho_order, M, [&](uint32_t r) { return (int64_t)c.sigma[r]; }, [&](uint32_t r) { return by_s[c.rec_s[r]]; });
                    sr.ortho_heap = heap_of_rank(sr.ortho_order.size());
                    sr.ortho_rank_of_heap.resize(sr.ortho_order.size());
                    for (size_t r = 0; r < sr.ortho_order.size(); ++r) sr.ortho_rank_of_heap[sr.ortho_heap[r]] = (uint32_t)r;
                    c.split_built = true;
                }
            }, 1);
        });
    // compute units the walk launches of this DP occupy at a time (see WalkUnits)
    const uint32_t walk_units = !use_walk ? 0u : use_walk2 ? (uint32_t)(((combos.size() + 7) & ~(size_t)7) * (1 + walk2_help))
                                                           : (uint32_t)((combos.size() + std::max<uint32_t>(walk_fold, 1u) - 1) / std::max<uint32_t>(walk_fold, 1u));
    const uint32_t walk_units_booked = g_walk_exclusive && walk_units ? WalkUnits::kBudget : walk_units;
    struct UnitsHeld { int device; uint32_t n; bool held; ~UnitsHeld() { if (held) g_walk_units.give(device, n); } } units_held{ctx->device, walk_units_booked, false};
    if (walk_units && ctx->peers.n <= 1) { g_walk_units.take(ctx->device, walk_units_booked); units_held.held = true; }
    hipError_t he = hipEventRecord(ev0, ctx->stream);
    if (he == hipSuccess) he = hipEventRecord(ctx->ev_fork, ctx->stream);
    for (uint32_t f = 0; f < far_streams && he == hipSuccess; ++f) he = hipStreamWaitEvent(ctx->aux[f], ctx->ev_fork, 0);
    auto max_recs = [&](uint32_t lo, uint32_t hi) {
        uint32_t m = 0;
        for (const Combo& c : combos) m = std::max(m, c.prefix[hi] - c.prefix[lo]);
        return m;
    };
    if (use_walk) {
        // macro-blocks of kChainMacro pairs: walk(k) on the context's stream is one launch of n_combos workgroups; the records of
        // macro-block k - 1 reach k's queries by a brute-force "near" launch in front of it; everything older by a "far" launch on
        // an auxiliary stream that may start as soon as walk(k - 2) is done
        uint32_t max_n = 0;
        for (const Combo& c : combos) max_n = std::max<uint32_t>(max_n, (uint32_t)c.rec_s.size());
        if (he == hipSuccess) he = cl_chain_launch_own_rec(D, max_n, ctx->stream);
        const uint32_t n_macro = (uint32_t)((M + kChainMacro - 1) / kChainMacro), bpm = kChainMacro / kChainBlock;
        std::vector<hipEvent_t> ev_walk(n_macro, nullptr), ev_seal(n_macro, nullptr);
        ev_far.assign(n_macro, nullptr);
        static_assert(kFarLag + 1 <= (uint32_t)kNumAuxStreams, "one auxiliary stream per far launch in flight plus the sealing stream");
        hipStream_t seal_stream = ctx->aux[far_lag];   // right behind the far streams: the context's own streams are few (cl_context_create)
        if (use_far) {
            if (he == hipSuccess) he = hipStreamWaitEvent(seal_stream, ctx->ev_fork, 0);
            for (uint32_t f = far_streams; f < far_lag && he == hipSuccess; ++f) he = hipStreamWaitEvent(ctx->aux[f], ctx->ev_fork, 0);
        }
        // Branch-and-bound or sweep?  The far kernels count the leaves they had to open against the leaves in range (status[2..5]).
        // At a few checkpoints the host waits for the launches so far and reads the counts: once they cover enough queries with a
        // long history, more than one leaf in sixteen opened means the sweep is the faster far pass for this input (repeats so
        // dense that every pair is near every other) and it takes over for the rest of the DP.  One decision per DP, taken between
        // launches: nothing is in flight when the mode changes.
        bool far_bb = use_far, far_decided = !use_far;
        static const char* far_mode_env = getenv("CL_CHAIN_FAR_MODE");   // A/B switch: "sweep" / "bb" pin the mode
        if (use_far && far_mode_env) { far_decided = true; far_bb = far_mode_env[0] != 's'; }
        // A merge group (cl_peer_api.cpp) shares the far pass of this DP: every member takes the same decision from the same numbers — affine,
        // 2 .. kPeerMaxCombos combinations, at least 16 macro-blocks — and the choice of far pass is not left to counts that differ by member
        auto& peers = ctx->peers;
        const bool shared = use_far && far_bb && peers.n > 1 && !sparse && combos.size() >= 2 && combos.size() <= kPeerMaxCombos && n_macro >= 16 &&
                            !(far_mode_env && far_mode_env[0] == 's') && peers.epoch + 1 < (1u << 12);
        uint32_t share_epoch = 0;
        if (shared) {
            far_decided = true;
            share_epoch = ++peers.epoch;
            peers.epoch_mark = std::max(peers.epoch_mark, peers.epoch);
            ++peers.shared_dps;
            F.share_n = peers.n;
            F.share_i = peers.me;
            // The ring slots of an inbox are reused by the NEXT shared DP: a member that has gone on to it must not store into a slot another member has
            // not folded yet (round-4 advisor: slots k and k' collide whenever k - k' = 16 mod 32, whatever the epochs' parity).  Every member raises its
            // "done" word in the others' memory when its serial stream has passed the last block of a shared DP; the far streams — the only ones that
            // store into other inboxes — wait for the words of the previous shared DP before their first launch of this one
            for (uint32_t m = 0; m < peers.n && he == hipSuccess; ++m) {
                if (m == peers.me) continue;
                for (uint32_t f = 0; f < far_lag && he == hipSuccess; ++f)
                    he = hipStreamWaitValue32(ctx->aux[f], peers.flags + kPeerDoneAt + m, peers.last_shared_epoch, hipStreamWaitValueGte, 0xFFFFFFFFu);
            }
            peers.last_shared_epoch = share_epoch;
        }
        // the events that order the streams ride on the launches that they follow (hipExtLaunchKernel's stop event) instead of being runtime calls of
        // their own: 3 of ~12 calls per macro-block.  CL_CHAIN_EXT_EVENTS=0: hipEventRecord as in rounds 2-3 (A/B)
        static const bool ext_events = [] { const char* e = getenv("CL_CHAIN_EXT_EVENTS"); return !e || e[0] != '0'; }();
        // Round 5: far(k) goes out as soon as the seal it waits for has been enqueued — at the end of iteration k - lag - 1 instead of iteration k — so that the
        // serial stream can take far(k)'s event IN FRONT of near(k) (far(k) has had lag + 1 blocks to finish by then) instead of between near(k) and walk(k): there the
        // cross-stream hop sat between two dependent launches of the serial chain (kernel trace, round 4: gap 7.4 | near 22.8 | gap 14.9 | walk 55.2 us against ~7 us
        // for a plain launch-to-launch dependency).  CL_CHAIN_FAR_WAIT_LATE=1: as in round 4 (A/B).  Not inside a merge group (the peers' slots are per iteration)
        static const bool far_early = getenv("CL_CHAIN_FAR_WAIT_LATE") == nullptr;
        std::vector<uint8_t> far_issued(n_macro, 0);   // 1: far(k) is out, 2: ... and was sent early
        auto issue_far_bb = [&](uint32_t kk, bool early) -> hipError_t {
            const uint32_t first_k = kk * kChainMacro, count_k = (uint32_t)std::min<uint64_t>(kChainMacro, M - first_k), near_lo_k = (kk - far_lag) * bpm;
            hipStream_t far_stream = ctx->aux[kk % far_lag];
            // every node inside the records [0, prefix[near_lo]) was sealed by seal(kk - lag - 1) or earlier
            hipError_t e = hipStreamWaitEvent(far_stream, ev_seal[kk - far_lag - 1], 0);
            // CL_CHAIN_DEBUG_SKIP_FAR=1 (measurements only, WRONG RESULTS): the DP without its far launches = the serial walk / near chain alone,
            // i.e. what a merge would cost its leader if other devices took the far pass off it
            static const bool skip_far = getenv("CL_CHAIN_DEBUG_SKIP_FAR") != nullptr;
            bool recorded = false;
            if (e == hipSuccess && !skip_far) {
                // (the launch records ev_far[kk] itself: one runtime call less per macro-block)
                if (ext_events) { e = cl_ring_event(ctx, 1, kk, &ev_far[kk]); recorded = e == hipSuccess; }
                if (e == hipSuccess) e = cl_chain_far_launch(D, F, first_k, count_k, near_lo_k, far_stream, recorded ? ev_far[kk] : nullptr);
            }
            if (!recorded) {
                if (e == hipSuccess) e = cl_ring_event(ctx, 1, kk, &ev_far[kk]);
                if (e == hipSuccess) e = hipEventRecord(ev_far[kk], far_stream);
            }
            far_issued[kk] = early ? 2 : 1;
            return e;
        };
        for (uint32_t k = 0; k < n_macro && he == hipSuccess; ++k) {
            if (far_decided && timing && use_far && k >= 256 && (k & (k - 1)) == 0) {   // (CL_CHAIN_TIMING: how the opened share moves after the decision)
                unsigned long long cnt[2] = {0, 0};
                if (hipStreamSynchronize(ctx->stream) == hipSuccess) {
                    for (uint32_t f = 0; f < far_lag; ++f) (void)hipStreamSynchronize(ctx->aux[f]);
                    if (cl_copy_sync(ctx, cnt, d_status.p + 2, sizeof(cnt), hipMemcpyDeviceToHost) == hipSuccess)
                        fprintf(stderr, "[chain_dp_batch]   far pass after %u macro-blocks: %llu of %llu leaves opened so far (%s)\n", k, cnt[0], cnt[1], far_bb ? "branch-and-bound" : "all-pairs sweep");
                }
            }
            if (!far_decided && k >= 96 && (k & (k - 1)) == 0) {   // k = 128, 256, 512, ...
                he = hipStreamSynchronize(ctx->stream);
                for (uint32_t f = 0; f < far_lag && he == hipSuccess; ++f) he = hipStreamSynchronize(ctx->aux[f]);
                unsigned long long cnt[2] = {0, 0};
                if (he == hipSuccess) he = cl_copy_sync(ctx, cnt, d_status.p + 2, sizeof(cnt), hipMemcpyDeviceToHost);
                if (he == hipSuccess && cnt[1] >= (1ull << 22)) {
                    far_decided = true;
                    far_bb = cnt[0] * 16 <= cnt[1];
                    if (!sparse) ctx->far_last_choice = far_bb ? 1 : 2;
                    if (timing) fprintf(stderr, "[chain_dp_batch]   far pass after %u macro-blocks: %llu of %llu leaves opened -> %s\n", k, cnt[0], cnt[1], far_bb ? "branch-and-bound" : "all-pairs sweep");
                }
            }
            const uint32_t first = k * kChainMacro, count = (uint32_t)std::min<uint64_t>(kChainMacro, M - first);
            const uint32_t lag = use_far ? far_lag : 1u;
            const uint32_t b0 = k * bpm, near_lo = k >= lag ? (k - lag) * bpm : 0;
            bool far_recorded = false;
            if (near_lo > 0) {
                hipStream_t far_stream = ctx->aux[k % (use_far ? far_lag : far_streams)];
                if (far_issued[k]) far_recorded = true;   // (sent at the end of iteration k - lag - 1)
                else if (use_far && far_bb && !shared) { he = issue_far_bb(k, false); far_recorded = true; }
                else if (use_far && far_bb) {
                    // every node inside the records [0, prefix[near_lo]) was sealed by seal(k - lag - 1) or earlier
                    he = hipStreamWaitEvent(far_stream, ev_seal[k - lag - 1], 0);
                    {
                        // this member's combinations; what it finds goes into slot k of the others' inboxes, then its arrival word there
                        ClFarDevice Fk = F;
                        // (the slot depends on the epoch's parity too: a member that has gone on to the next shared DP while another still folds the
                        // last blocks of this one then writes the other half of the ring — 2 * lag + 2 <= 18 run-ahead inside a DP was the only bound before)
                        const uint32_t slot = (k + (kPeerRing / 2) * (share_epoch & 1u)) % kPeerRing, word = (share_epoch << 20) | (k + 1);
                        uint32_t o = 0;
                        for (uint32_t m = 0; m < peers.n; ++m) if (m != peers.me) Fk.peer_out[o++] = peers.peer_inbox[m] + (size_t)slot * kPeerSlotInts;
                        if (he == hipSuccess) he = cl_chain_far_launch(D, Fk, first, count, near_lo, far_stream, nullptr);
                        for (uint32_t m = 0; m < peers.n && he == hipSuccess; ++m)
                            if (m != peers.me) he = hipStreamWriteValue32(far_stream, peers.peer_flags[m] + (size_t)peers.me * kPeerRing + slot, word, 0);
                        ++peers.shared_far_launches;
                    }
                } else if (use_far) {
                    // the all-pairs sweep has taken over (see the checkpoints below); same lag, so that sweeps run side by side
                    he = hipStreamWaitEvent(far_stream, ev_walk[k - lag - 1], 0);
                    if (he == hipSuccess) he = cl_chain_launch_inter(D, first, count, 0, near_lo, max_recs(0, near_lo), kChainFarTile, far_stream);
                } else {
                    he = hipStreamWaitEvent(far_stream, ev_walk[k - 2], 0);
                    const uint32_t recs = max_recs(0, near_lo);
                    uint32_t tile = kChainFarTile;
                    if (sparse) {
                        tile = kChainNearTile;
                        while (tile < kChainFarTile && (recs + tile - 1) / tile > kChainFullGrid) tile *= 2;
                    }
                    if (he == hipSuccess) he = cl_chain_launch_inter(D, first, count, 0, near_lo, recs, tile, far_stream);
                }
                if (!far_recorded) {
                    if (he == hipSuccess) he = cl_ring_event(ctx, 1, k, &ev_far[k]);
                    if (he == hipSuccess) he = hipEventRecord(ev_far[k], far_stream);
                }
            }
            hipEvent_t ev_near_a = nullptr;
            // the serial stream takes the event of a far(k) that went out early IN FRONT of the near launch (see issue_far_bb above)
            bool far_waited = false;
            if (he == hipSuccess && ev_far[k] && far_issued[k] == 2) { he = hipStreamWaitEvent(ctx->stream, ev_far[k], 0); far_waited = true; }
            if (he == hipSuccess && b0 > near_lo) {
                // the far pass stops at a leaf boundary: the near launch starts there.  Only the records of macro-block k - 1 need walk(k - 1): they
                // are swept on the serial stream; the older macro-blocks of the near range were final one walk earlier and are swept on a stream
                // of their own beside walk(k - 1) — with CL_CHAIN_NEAR_SPLIT=1 only: by default one launch on the serial stream, as in rounds 2-3
                static const bool near_split = [] { const char* e = getenv("CL_CHAIN_NEAR_SPLIT"); return e && e[0] == '1'; }();   // (measured SLOWER: 2 x 1 Mbp affine DP 162 -> 230 ms — the extra stream and its two event hops per macro-block cost more than the sweep they take off the serial stream; off unless asked for)
                const uint32_t b_split = near_split && k >= 2 ? std::max(near_lo, (k - 1) * bpm) : near_lo;
                ClChainDevice Dn = D;
                if (use_far && near_lo > 0) Dn.lo_mask = 63u;
                if (b_split > near_lo) {
                    hipStream_t near_stream = ctx->aux[(use_far ? far_lag : far_streams) + 1];
                    he = hipStreamWaitEvent(near_stream, ev_walk[k - 2], 0);
                    if (he == hipSuccess) he = cl_chain_launch_inter(Dn, first, count, near_lo, b_split, max_recs(near_lo, b_split) + Dn.lo_mask, kChainNearTile, near_stream);
                    if (he == hipSuccess) he = cl_ring_event(ctx, 3, k, &ev_near_a);
                    if (he == hipSuccess) he = hipEventRecord(ev_near_a, near_stream);
                    Dn.lo_mask = 0;
                }
                if (he == hipSuccess) he = cl_chain_launch_inter(Dn, first, count, b_split, b0, max_recs(b_split, b0) + Dn.lo_mask, kChainNearTile, ctx->stream);
            }
            if (he == hipSuccess && ev_near_a) he = hipStreamWaitEvent(ctx->stream, ev_near_a, 0);
            if (he == hipSuccess && ev_far[k] && !far_waited) he = hipStreamWaitEvent(ctx->stream, ev_far[k], 0);
            if (shared && near_lo > 0) {
                // the other members' combinations of this macro-block: wait for their arrival words, fold their slot into the running maxima
                const uint32_t slot = (k + (kPeerRing / 2) * (share_epoch & 1u)) % kPeerRing, word = (share_epoch << 20) | (k + 1);
                for (uint32_t m = 0; m < peers.n && he == hipSuccess; ++m)
                    if (m != peers.me) he = hipStreamWaitValue32(ctx->stream, peers.flags + (size_t)m * kPeerRing + slot, word, hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (he == hipSuccess) he = cl_chain_far_merge(D, peers.inbox + (size_t)slot * kPeerSlotInts, first, count, peers.n, peers.me, ctx->stream);
                ++peers.merged_blocks;
            }
            if (he == hipSuccess) he = cl_ring_event(ctx, 0, k, &ev_walk[k]);
            if (he == hipSuccess) he = use_walk2 ? cl_chain_launch_walk2(D, first, count, walk2_qpt, walk2_help, ctx->stream, ext_events ? ev_walk[k] : nullptr)
                                                 : walk_fold > 1 ? cl_chain_launch_walk_fold(D, walk_fold, first, count, ctx->stream, ext_events ? ev_walk[k] : nullptr)
                                                 : cl_chain_launch_walk(D, first, count, ctx->stream, ext_events ? ev_walk[k] : nullptr);
            if (he == hipSuccess && !ext_events) he = hipEventRecord(ev_walk[k], ctx->stream);
            if (use_far && far_bb && he == hipSuccess && k + far_lag + 1 < n_macro) {
                he = hipStreamWaitEvent(seal_stream, ev_walk[k], 0);
                if (he == hipSuccess) he = cl_ring_event(ctx, 2, k, &ev_seal[k]);
                if (he == hipSuccess) he = cl_chain_far_seal(D, F, d_seal_items.p, seal_off[k], seal_off[k + 1] - seal_off[k], seal_big[k], seal_stream, ext_events ? ev_seal[k] : nullptr);
                if (he == hipSuccess && !ext_events) he = hipEventRecord(ev_seal[k], seal_stream);
                if (he == hipSuccess && far_early && !shared) he = issue_far_bb(k + far_lag + 1, true);
            }
        }
        if (shared) {   // this member has folded every slot of the DP (stream order: behind its last walk)
            for (uint32_t m = 0; m < peers.n && he == hipSuccess; ++m)
                if (m != peers.me) he = hipStreamWriteValue32(ctx->stream, peers.peer_flags[m] + kPeerDoneAt + peers.me, share_epoch, 0);
        }
        if (use_far && he == hipSuccess) he = hipStreamSynchronize(seal_stream);
        // (the walk path's events are the context's ring events: nothing to destroy)
        ev_far.assign(ev_far.size(), nullptr);
    } else if (group_path) {
        for (size_t gi = 0; gi + 1 < group_first.size() && he == hipSuccess; ++gi)
            he = cl_chain_launch_group(D, group_first[gi], group_first[gi + 1], rec_off[group_first[gi + 1]] - rec_off[group_first[gi]], d_dp_enc.p, ctx->stream);
    } else {
        uint32_t near_lo = 0;   // first block the current group's far launch did not cover
        for (uint32_t b = 0; b < n_blocks && he == hipSuccess; ++b) {
            const uint32_t first = b * kChainBlock, count = (uint32_t)std::min<uint64_t>(kChainBlock, M - first);
            if (b % far_group == 0) {
                near_lo = b > 1 ? b - 1 : 0;
                if (near_lo > 0) {
                    // far predecessors of the whole group: blocks [0, near_lo), final once walk(near_lo - 1) is done
                    hipStream_t far_stream = ctx->aux[(b / far_group) % far_streams];
                    he = hipStreamWaitEvent(far_stream, ev_intra[near_lo - 1], 0);
                    // a launch ends with one atomic merge per query and workgroup: small tiles (low latency) only where that is
                    // cheap — sparse mode has one maximum per query — and while the grid does not fill the chip
                    const uint32_t recs = max_recs(0, near_lo);
                    uint32_t tile = kChainFarTile;
                    if (sparse) {
                        tile = kChainNearTile;
                        while (tile < kChainFarTile && (recs + tile - 1) / tile > kChainFullGrid) tile *= 2;
                    }
                    const uint32_t group_count = (uint32_t)std::min<uint64_t>((uint64_t)far_group * kChainBlock, M - first);
                    if (he == hipSuccess) he = cl_chain_launch_inter(D, first, group_count, 0, near_lo, recs, tile, far_stream);
                    if (he == hipSuccess) he = hipEventCreateWithFlags(&ev_far[b], hipEventDisableTiming);
                    if (he == hipSuccess) he = hipEventRecord(ev_far[b], far_stream);
                }
            }
            // near predecessors: blocks [near_lo, b), on the sequential stream right after walk(b - 1)
            const bool fuse_near = combos.size() == 1;
            if (he == hipSuccess && b > near_lo && !fuse_near) he = cl_chain_launch_inter(D, first, count, near_lo, b, max_recs(near_lo, b), kChainNearTile, ctx->stream);
            if (he == hipSuccess && ev_far[b]) he = hipStreamWaitEvent(ctx->stream, ev_far[b], 0);
            if (he == hipSuccess) he = cl_chain_launch_intra(D, first, count, fuse_near ? b - near_lo : 0u, ctx->stream);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&ev_intra[b], hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventRecord(ev_intra[b], ctx->stream);
        }
    }
    if (he == hipSuccess) he = hipEventRecord(ev1, ctx->stream);
    lap("enqueue (host)");
    if (timing) {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        fprintf(stderr, "[chain_dp_batch]   device memory held by the context: %.1f MB (%s queries, far pass %s: %u padded records in %u levels, %u + %u tags, walk %s)\n", ctx->dev_live_bytes / 1048576.0,
                factored ? "factored" : "dense", use_far ? "on" : "off", use_far ? F.r_pad : 0u, use_far ? F.n_levels : 0u, n_tag[0], n_tag[1],
                !use_walk ? "off" : use_walk2 ? "2" : walk_fold > 1 ? "fold" : "1");
    }
    if (he == hipSuccess && ctx->peers.n > 1) {
        // inside a merge group the serial stream waits for the other members' arrival words: a member that has failed would leave this
        // call waiting for ever, so the wait is a poll with a limit (CL_PEER_TIMEOUT_S, default 600 s); past it the context is unusable
        static const double limit_s = [] { const char* e = getenv("CL_PEER_TIMEOUT_S"); const double v = e ? atof(e) : 600.0; return v > 0 ? v : 600.0; }();
        const auto t_wait = std::chrono::steady_clock::now();
        while ((he = hipStreamQuery(ctx->stream)) == hipErrorNotReady) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count() > limit_s) {
                (void)hipGetLastError();
                ctx->poisoned = true;
                cl_set_error(ctx, "chaining DP: the other members of the merge group did not deliver within %.0f s (a member failed?); this context cannot be used any more", limit_s);
                for (auto e : ev_intra) if (e) (void)hipEventDestroy(e);
                return CL_ERR_HIP;   // (device buffers of this DP are left to the stuck stream: releasing them would wait for it)
            }
            usleep(50);
        }
        if (he == hipErrorNotReady) he = hipSuccess;
        (void)hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    for (uint32_t f = 0; f < std::max<uint32_t>(far_streams, far_lag) && he == hipSuccess; ++f) he = hipStreamSynchronize(ctx->aux[f]);
    if (units_held.held) { g_walk_units.give(ctx->device, walk_units_booked); units_held.held = false; }   // (the device is done with this DP's launches)
    for (auto e : ev_intra) if (e) (void)hipEventDestroy(e);
    for (auto e : ev_far) if (e) (void)hipEventDestroy(e);
    if (he != hipSuccess) return hip_fail(he, "chaining DP kernels");
    if (use_walk) {
        uint32_t status = 0;
        he = cl_copy_sync(ctx, &status, d_status.p, sizeof(status), hipMemcpyDeviceToHost);
        if (he != hipSuccess) return hip_fail(he, "chaining DP status");
        if (D.debug) {
            uint32_t dbg[32] = {};
            (void)cl_copy_sync(ctx, dbg, d_status.p, sizeof(dbg), hipMemcpyDeviceToHost);
            const double steps = std::max(1u, dbg[8]);
            fprintf(stderr, "[chain_dp_batch]   walk2: %u steps of combination 0, %u queries finalised, %u polled at their finalisation, %u evaluated out of LDS (no helper result); "
                            "wave 0 per step (100 MHz ticks): before the barrier %.1f, in the barrier %.1f, behind it %.1f; helper batches %u, records %u, polls without news %u; wave 0 finalised in %u steps: %.1f ticks before the barrier there, %.1f behind the barrier before; per launch: %.1f ticks in front of the loop, %.1f in it; not asked %u, no granule of the block %u; shader clock %.0f MHz (s_memtime against the 100 MHz s_memrealtime)\n",
                    dbg[8], dbg[9], dbg[10], dbg[11], dbg[12] / steps, dbg[13] / steps, dbg[14] / steps, dbg[16], dbg[17], dbg[18], dbg[19], dbg[20] / (double)std::max(1u, dbg[19]), dbg[21] / (double)std::max(1u, dbg[19]), dbg[22] / (double)std::max(1u, dbg[24]), dbg[23] / (double)std::max(1u, dbg[24]), dbg[25], dbg[26], 100.0 * dbg[27] / (double)std::max(1u, dbg[28]));
        }
        static const char* debug_stall = getenv("CL_CHAIN_DEBUG_STALL");   // test hook: behave as if a wait had expired ("first": not in the attempt that has the device to itself)
        if (debug_stall && !(g_walk_exclusive && !strcmp(debug_stall, "first"))) status = 1;
        if (status != 0) {
            cl_set_error(ctx, "chaining DP: a workgroup of the walk kernel gave up waiting for its siblings (%zu combinations not all resident?)", combos.size());
            (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
            cleanup();
            return kWalkStalled;
        }
    }
    {
        float dev_ms = 0;
        (void)hipEventElapsedTime(&dev_ms, ev0, ev1);
        tm.device_ms += dev_ms;
    }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);

    lap("sync");
    std::vector<float> dp_sorted(M);
    std::vector<std::vector<int>> acc_own;
    std::vector<const int*> acc(combos.size(), nullptr);
    std::vector<char> pack_host;
    if (d_pack.p) {
        // a small DP: its DP values and query results lie together at the end of the block
        pack_host.resize(pack_down_bytes);
        he = cl_copy_sync(ctx, pack_host.data(), d_pack.p + pack_dp_off, pack_down_bytes, hipMemcpyDeviceToHost);
        if (he == hipSuccess) {
            memcpy(dp_sorted.data(), pack_host.data(), M * sizeof(float));
            const size_t first_acc = (size_t)((char*)combos[0].d_acc.p - d_pack.p) - pack_dp_off;
            for (size_t ci = 0; ci < combos.size(); ++ci) acc[ci] = reinterpret_cast<const int*>(pack_host.data() + first_acc + ci * pack_acc_stride);
        }
    } else
    he = cl_copy_sync(ctx, dp_sorted.data(), d_dp.p, M * sizeof(float), hipMemcpyDeviceToHost);
    // the stored query results of every combination (7 per pair): into the context's page-locked area when it can be had — 35 MB per
    // combination at 1.25 M pairs, 25 combinations at the root of a 10-sequence tree.  Beyond kLazyAccBytes (a root of 25 + 25 paths: 625
    // combinations, 21.9 GB at 1.25 M pairs — 4.3 s of copies into 22 GB of pageable memory, more than the DP itself) nothing is downloaded: the
    // traceback reads the results of the pair it stands on and nothing else, so it fetches that row (7 x combinations words) step by step
    // (chain_acc_row_kernel; a launch, a 17 KB copy and a wait per chain step).  CL_CHAIN_LAZY_ACC=0/1 pins the choice (tests).
    constexpr size_t kLazyAccBytes = 1536ull << 20;
    static const char* lazy_env = getenv("CL_CHAIN_LAZY_ACC");
    const bool lazy_acc = !d_pack.p && (lazy_env ? lazy_env[0] == '1' : combos.size() * (size_t)M * 7 * sizeof(int) > kLazyAccBytes);
    std::vector<int> row_host;
    uint32_t row_s = kNone;
    int* row_pin = nullptr;
    if (he == hipSuccess && lazy_acc) {
        rc = d_row.alloc(ctx, combos.size() * 7);
        if (rc) { cleanup(); return rc; }
        row_pin = (int*)cl_pinned(ctx, combos.size() * 7 * sizeof(int));
        if (!row_pin) { row_host.resize(combos.size() * 7); row_pin = row_host.data(); }
    }
    // row of pair s: 7 words per combination
    auto acc_row = [&](uint32_t s) -> const int* {
        if (row_s == s) return row_pin;
        hipError_t e = cl_chain_acc_row(D, s, d_row.p, ctx->stream);
        if (e == hipSuccess) e = cl_copy_sync(ctx, row_pin, d_row.p, combos.size() * 7 * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { cl_set_error(ctx, "traceback: fetching the query results of pair %u failed: %s", s, hipGetErrorString(e)); return nullptr; }
        row_s = s;
        return row_pin;
    };
    if (he == hipSuccess && !d_pack.p && !lazy_acc) {
        const size_t per = (size_t)M * 7;
        int* pin = (int*)cl_pinned(ctx, combos.size() * per * sizeof(int));
        if (pin) {
            for (size_t ci = 0; ci < combos.size() && he == hipSuccess; ++ci) {
                acc[ci] = pin + ci * per;
                he = hipMemcpyAsync(pin + ci * per, combos[ci].d_acc.p, per * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
            }
            if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
        } else {
            acc_own.resize(combos.size());
            for (size_t ci = 0; ci < combos.size() && he == hipSuccess; ++ci) {
                acc_own[ci].resize(per);
                acc[ci] = acc_own[ci].data();
                he = cl_copy_sync(ctx, acc_own[ci].data(), combos[ci].d_acc.p, per * sizeof(int), hipMemcpyDeviceToHost);
            }
        }
    }
    if (he != hipSuccess) { cl_set_error(ctx, "download failed: %s", hipGetErrorString(he)); cleanup(); return CL_ERR_HIP; }
    if (ortho_prebuild.joinable()) ortho_prebuild.join();
    lap("download");
    double t_split = 0, t_ortho = 0, t_diag = 0, t_gapfree = 0, t_cand = 0;   // CL_CHAIN_TIMING: where the traceback's time goes
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tadd = [&](double& acc_ms, std::chrono::steady_clock::time_point t0) { acc_ms += std::chrono::duration<double, std::milli>(tnow() - t0).count(); };

    const auto T1 = std::chrono::steady_clock::now();
    // value index: per combination and tree kind, (encoded stored value, record) sorted by value — built on first use: the
    // traceback touches only the (combination, kind) pairs that win a step of the chain
    std::vector<std::vector<std::vector<int>>> vkeys(combos.size(), std::vector<std::vector<int>>(7));
    std::vector<std::vector<std::vector<uint32_t>>> vrecs(combos.size(), std::vector<std::vector<uint32_t>>(7));
    std::vector<std::vector<char>> vbuilt(combos.size(), std::vector<char>(7, 0));
    size_t vtemp_bytes = 0;
    float index_ms = 0;
    // (the streams have run dry by now — the download above waited for them: one check for the five blocks)
    auto release_index = [&]() { cl_ctx_quiesce(ctx); k_in.release(true); k_out.release(true); i_in.release(true); i_out.release(true); vtemp.release(true); };
    if (!d_pack.p) {   // (a small DP builds its value index on the host)
        uint32_t nmax = 0;
        for (const Combo& c : combos) nmax = std::max<uint32_t>(nmax, (uint32_t)c.rec_s.size());
        CH(k_in.alloc(ctx, nmax)); CH(k_out.alloc(ctx, nmax)); CH(i_in.alloc(ctx, nmax)); CH(i_out.alloc(ctx, nmax));
        he = cl_chain_sort_values(nullptr, nmax, k_in.p, i_in.p, k_out.p, i_out.p, nullptr, &vtemp_bytes, ctx->stream);
        if (he == hipSuccess) { rc = vtemp.alloc(ctx, vtemp_bytes); if (rc) he = hipErrorOutOfMemory; }
        if (he != hipSuccess) { cl_set_error(ctx, "value index failed: %s", hipGetErrorString(he)); release_index(); cleanup(); return CL_ERR_HIP; }
    }
    auto value_index = [&](size_t ci, int kind) -> bool {
        if (vbuilt[ci][kind]) return true;
        const auto t = std::chrono::steady_clock::now();
        const uint32_t n = (uint32_t)combos[ci].rec_s.size();
        if (d_pack.p) {
            // a small DP: the stored values came back with the DP values; the same order as the device's stable radix sort
            const float* val = reinterpret_cast<const float*>(pack_host.data() + ((char*)combos[ci].d_val.p - d_pack.p - pack_dp_off)) + (size_t)kind * n;
            auto& recs = vrecs[ci][kind];
            recs.resize(n);
            std::iota(recs.begin(), recs.end(), 0u);
            std::vector<int> key(n);
            for (uint32_t r = 0; r < n; ++r) key[r] = enc(val[r]);
            std::stable_sort(recs.begin(), recs.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
            vkeys[ci][kind].resize(n);
            for (uint32_t i = 0; i < n; ++i) vkeys[ci][kind][i] = key[recs[i]];
            index_ms += ms_since(t);
            vbuilt[ci][kind] = 1;
            return true;
        }
        hipError_t e = cl_chain_sort_values(combos[ci].d_val.p + (size_t)kind * n, n, k_in.p, i_in.p, k_out.p, i_out.p, vtemp.p, &vtemp_bytes, ctx->stream);
        vkeys[ci][kind].resize(n);
        vrecs[ci][kind].resize(n);
        if (e == hipSuccess && n) e = hipMemcpyAsync(vkeys[ci][kind].data(), k_out.p, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n) e = hipMemcpyAsync(vrecs[ci][kind].data(), i_out.p, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        index_ms += ms_since(t);
        if (e != hipSuccess) { cl_set_error(ctx, "value index failed: %s", hipGetErrorString(e)); return false; }
        vbuilt[ci][kind] = 1;
        return true;
    };
    tm.index_ms += ms_since(T1);
    const auto T2 = std::chrono::steady_clock::now();

    // records of one instance inside a combination (the reference's trees belong to ONE chaining call)
    auto sub_recs = [&](Combo& c, uint32_t k) -> Combo::SubRecs& {
        if (!c.split_built) {
            const auto t0 = tnow();
            if (K == 1) {
                auto& recs = c.per_sub[0].recs;
                recs.resize(c.rec_s.size());
                std::iota(recs.begin(), recs.end(), 0u);
            } else {
                std::vector<Combo::SubRecs*> of_sub(K, nullptr);
                for (uint32_t r = 0; r < c.rec_s.size(); ++r) {
                    const uint32_t sub = pairs[by_s[c.rec_s[r]]].sub;
                    if (!of_sub[sub]) of_sub[sub] = &c.per_sub[sub];   // unordered_map: references stay valid across insertions
                    of_sub[sub]->recs.push_back(r);
                }
            }
            c.split_built = true;
            tadd(t_split, t0);
        }
        return c.per_sub[k];
    };

    // an instance's records of one combination in (shift, slot) order with the implicit-heap layout of the reference's outer tree
    auto ortho_of = [&](Combo::SubRecs& sr, const Combo& c) {
        if (!sr.ortho_order.empty()) return;
        const auto t0 = tnow();
        sr.ortho_order = sr.recs;
        sort_by_key_then_slot(sr.ortho_order, M, [&](uint32_t r) { return (int64_t)c.sigma[r]; }, [&](uint32_t r) { return by_s[c.rec_s[r]]; });
        sr.ortho_heap = heap_of_rank(sr.ortho_order.size());
        sr.ortho_rank_of_heap.resize(sr.ortho_order.size());
        for (size_t r = 0; r < sr.ortho_order.size(); ++r) sr.ortho_rank_of_heap[sr.ortho_heap[r]] = (uint32_t)r;
        tadd(t_ortho, t0);
    };
    uint64_t n_scanned = 0, n_steps = 0;
    // ---- optimum and traceback (anchorer.hpp:2483-2531), instance by instance -----------------------------------------
    for (uint32_t k = 0; k < K; ++k) {
        const SubCtx& sk = sc[k];
        if (sk.pair_hi == sk.pair_lo) continue;
        float opt = CL_CHAIN_NEG;
        uint32_t best_slot = kNone;
        for (uint32_t slot = sk.pair_lo; slot < sk.pair_hi; ++slot) {
            float v = dp_sorted[s_of_slot[slot]];
            const float f = final_term[slot];
            if (f == CL_CHAIN_NEG) v = f;
            else v += f;
            if (v > opt && v > sk.min_score) { opt = v; best_slot = slot; }
        }
        std::vector<uint32_t> chain_slots;
        std::vector<std::pair<uint32_t, uint32_t>> edges;   // (scratch of the steps below)
        std::vector<uint32_t> cand;
        uint64_t n_ties = 0;
        uint32_t here = best_slot;
        const uint32_t C1 = (uint32_t)sk.x[0]->chain_size(), C2 = (uint32_t)sk.x[1]->chain_size();
        while (here != kNone) {
            ++n_steps;
            chain_slots.push_back(here);
            const uint32_t s = s_of_slot[here];
            const float dpv = dp_sorted[s], w = weight[s];
            if (!(dpv > init_w[s])) break;  // no candidate was strictly greater than the chain that starts here: chain start
            const Pair& p = pairs[here];
            // the reference's candidate order: forward edges by the topological position of their source node, then chain1;
            // chain2 ascending; gap-free tree, then trees 0..5 (anchorer.hpp:2352-2413)
            edges.clear();                                      // (position of from-node, p1)
            for (uint32_t p1 = 0; p1 < C1; ++p1) {
                const uint32_t pr = sk.x[0]->predecessor_index(p.b1, p1);
                if (pr == kNone) continue;
                const uint64_t from = sk.x[0]->node_at(p1, pr);
                if (sk.has_start[p.b1] && sk.after_end[from]) edges.emplace_back(sk.pos1[from], p1);
            }
            std::sort(edges.begin(), edges.end());
            const int* arow = nullptr;
            if (lazy_acc && !(arow = acc_row(s))) { release_index(); cleanup(); return CL_ERR_HIP; }
            int win_combo = -1, win_kind = -1;
            for (size_t e = 0; e < edges.size() && win_combo < 0; ++e)
                for (uint32_t p2 = 0; p2 < C2 && win_combo < 0; ++p2) {
                    const uint32_t ci = combo_of[(size_t)tag_of(k, 0, edges[e].second) * n_tag[1] + tag_of(k, 1, p2)];
                    if (ci == kNone) continue;  // empty trees
                    const int* a = lazy_acc ? arow + (size_t)ci * 7 : &acc[ci][(size_t)s * 7];
                    uint32_t c_qt, c_qoff;
                    int32_t c_q;
                    query_at(ci, s, c_qt, c_qoff, c_q);
                    for (int kind = 0; kind < (sparse ? 1 : 7); ++kind) {
                        if (a[kind] == enc(CL_CHAIN_NEG)) continue;
                        const float stored = dec(a[kind]);
                        float cand;
                        if (kind == 0) cand = stored + w;
                        else {
                            const int pw = kind - 1;
                            const double pen = (pw % 2 == 1) ? local_scale * (cp->gap_open[pw / 2] + cp->gap_extend[pw / 2] * (double)c_q)
                                                             : local_scale * (cp->gap_open[pw / 2] - cp->gap_extend[pw / 2] * (double)c_q);
                            cand = (float)((double)(stored + w) - pen);
                        }
                        if (cand == dpv) { win_combo = (int)ci; win_kind = kind; break; }
                    }
                }
            if (win_combo < 0) { cl_set_error(ctx, "traceback: no candidate reproduces dp of pair %u", here); cleanup(); return CL_ERR_HIP; }
            Combo& c = combos[win_combo];
            uint32_t wq_t, wq_off;
            int32_t wq;
            query_at((size_t)win_combo, s, wq_t, wq_off, wq);
            // every predecessor whose stored value equals the query's maximum and which lies in the query's range
            cand.clear();
            {
                const auto t0 = tnow();
                const int target = lazy_acc ? arow[(size_t)win_combo * 7 + win_kind] : acc[win_combo][(size_t)s * 7 + win_kind];
                if (!value_index((size_t)win_combo, win_kind)) { release_index(); cleanup(); return CL_ERR_HIP; }
                const auto& keys = vkeys[win_combo][win_kind];
                const auto& recs = vrecs[win_combo][win_kind];
                const uint32_t qt = wq_t, qoff = wq_off;
                const int32_t qq = wq;
                for (size_t i = std::lower_bound(keys.begin(), keys.end(), target) - keys.begin(); i < keys.size() && keys[i] == target; ++i) {
                    ++n_scanned;
                    const uint32_t r = recs[i];
                    if (c.rec_s[r] >= s) continue;
                    const int32_t sg = c.sigma[r];
                    const bool kind_ok = win_kind == 0 ? sg == qq : ((win_kind - 1) % 2 == 1 ? sg < qq : sg > qq);
                    if (kind_ok && c.ins_t[r] <= qt && c.off[r] < qoff) cand.push_back(r);
                }
                tadd(t_cand, t0);
            }
            const uint32_t count = (uint32_t)cand.size();
            if (count == 0) { cl_set_error(ctx, "traceback: query of pair %u has no predecessor at its maximum", here); cleanup(); return CL_ERR_HIP; }
            uint32_t win_rec;
            if (count == 1) {
                win_rec = cand[0];
            } else {
                ++n_ties;
                auto slot_of_rec = [&](uint32_t r) { return by_s[c.rec_s[r]]; };
                if (win_kind == 0) {
                    // gap-free tree of this diagonal: records with the same shift, keyed (offset, match id);
                    // inside an off-path subtree the earliest inserted wins, insertion order = (position of e1, slot)
                    // members of the tree, as (offset, slot, record-or-none): affine mode -> the records of this combination with
                    // the query's shift; sparse mode -> every pair whose e2 is filed under chain2, whatever its chain1
                    // (search_trees[i][j] is built from search_tree_data[j] for every i, anchorer.hpp:1581-1592)
                    GapFreeTree* tree;
                    if (!sparse) {
                        Combo::SubRecs& sr = sub_recs(c, k);
                        tree = &sr.diag_tree[wq];
                        if (!tree->built) {   // the records of this shift are one run of the (shift, slot) order
                            const auto t0 = tnow();
                            ortho_of(sr, c);
                            const int32_t qq = wq;
                            auto it = std::partition_point(sr.ortho_order.begin(), sr.ortho_order.end(), [&](uint32_t r) { return c.sigma[r] < qq; });
                            for (; it != sr.ortho_order.end() && c.sigma[*it] == qq; ++it) tree->mem.push_back(GapFreeTree::Member{c.off[*it], slot_of_rec(*it), *it});
                            tadd(t_diag, t0);
                        }
                    } else {
                        tree = &sparse_trees[((uint64_t)c.p2 << 32) | k];
                        if (!tree->built)
                            for (Combo& oc : combos)
                                if (oc.p2 == c.p2)
                                    for (uint32_t r : sub_recs(oc, k).recs) tree->mem.push_back(GapFreeTree::Member{oc.off[r], by_s[oc.rec_s[r]], r});
                    }
                    if (!tree->built) {
                        const auto t0 = tnow();
                        finish_tree(tree, !sparse);   // (affine: the members are a run of the (shift, slot) order, i.e. in slot order already)
                        tadd(t_gapfree, t0);
                    }
                    const auto& mem = tree->mem;
                    const size_t n = mem.size();
                    const auto& h = tree->heap;
                    const auto& rank_of_heap = tree->rank_of_heap;
                    const size_t rhi = std::partition_point(mem.begin(), mem.end(), [&](const GapFreeTree::Member& m) { return m.off < wq_off; }) - mem.begin();
                    std::vector<std::pair<size_t, uint32_t>> ch;  // (heap node, record)
                    for (uint32_t r : cand) {
                        const GapFreeTree::Member key{c.off[r], slot_of_rec(r), r};
                        const size_t rk = std::lower_bound(mem.begin(), mem.end(), key, [](const GapFreeTree::Member& a, const GapFreeTree::Member& b) { return a.off != b.off ? a.off < b.off : a.slot < b.slot; }) - mem.begin();
                        ch.emplace_back(h[rk], r);
                    }
                    auto earlier = [&](uint32_t a, uint32_t b) {
                        const uint32_t pa = sk.pos1[pairs[slot_of_rec(a)].e1], pb = sk.pos1[pairs[slot_of_rec(b)].e1];
                        return pa != pb ? pa < pb : slot_of_rec(a) < slot_of_rec(b);
                    };
                    win_rec = kNone;
                    replay_units(n, rank_of_heap, 0, rhi,
                                 [&](size_t x) { for (auto& e : ch) if (e.first == x) { win_rec = e.second; return true; } return false; },
                                 [&](size_t x) {
                                     for (auto& e : ch) if (in_subtree(e.first, x) && (win_rec == kNone || earlier(e.second, win_rec))) win_rec = e.second;
                                     return win_rec != kNone;
                                 });
                } else {
                    // orthogonal tree of this combination: all its records keyed ((shift, match id), offset); inside a
                    // cross tree the values are (score, outer index) pairs, so the larger outer heap index wins
                    Combo::SubRecs& sr = sub_recs(c, k);
                    auto key_less = [&](uint32_t a, uint32_t b) { return c.sigma[a] != c.sigma[b] ? c.sigma[a] < c.sigma[b] : slot_of_rec(a) < slot_of_rec(b); };
                    ortho_of(sr, c);
                    const size_t n = sr.ortho_order.size();
                    const int32_t qq = wq;
                    const bool odd = (win_kind - 1) % 2 == 1;
                    // rank interval of the key1 range: shift < query (odd trees) or shift > query (even trees)
                    size_t lo = 0, hi = n;
                    auto first_ge = [&](int64_t v) {
                        return (size_t)(std::partition_point(sr.ortho_order.begin(), sr.ortho_order.end(), [&](uint32_t r) { return (int64_t)c.sigma[r] < v; }) - sr.ortho_order.begin());
                    };
                    if (odd) hi = first_ge(qq);
                    else lo = first_ge((int64_t)qq + 1);
                    std::vector<std::pair<size_t, uint32_t>> ch;
                    for (uint32_t r : cand) {
                        const size_t rk = std::lower_bound(sr.ortho_order.begin(), sr.ortho_order.end(), r, key_less) - sr.ortho_order.begin();
                        ch.emplace_back(sr.ortho_heap[rk], r);
                    }
                    win_rec = kNone;
                    size_t win_heap = 0;
                    replay_units(n, sr.ortho_rank_of_heap, lo, hi,
                                 [&](size_t x) { for (auto& e : ch) if (e.first == x) { win_rec = e.second; return true; } return false; },
                                 [&](size_t x) {
                                     for (auto& e : ch) if (in_subtree(e.first, x) && (win_rec == kNone || e.first > win_heap)) { win_rec = e.second; win_heap = e.first; }
                                     return win_rec != kNone;
                                 });
                }
                if (win_rec == kNone) { cl_set_error(ctx, "traceback: tie resolution failed for pair %u", here); cleanup(); return CL_ERR_HIP; }
            }
            here = by_s[c.rec_s[win_rec]];
            if (chain_slots.size() > M) { cl_set_error(ctx, "traceback loop"); cleanup(); return CL_ERR_HIP; }
        }
        std::reverse(chain_slots.begin(), chain_slots.end());
        ChainSubResult& res = results[k];
        res.n_ties = n_ties;
        const size_t na = chain_slots.size();
        res.chain.resize(3 * na);
        for (size_t i = 0; i < na; ++i) {
            const Pair& p = pairs[chain_slots[i]];
            res.chain[3 * i] = p.set;
            res.chain[3 * i + 1] = p.i1;
            res.chain[3 * i + 2] = p.i2;
        }
        if (!sparse) {   // gap annotation (anchorer.hpp:2443-2468)
            res.gap.assign(na + 1, 0);
            res.gap_score.assign(na + 1, 0.0);
            for (size_t i = 1; i < na; ++i) {
                const Pair& a = pairs[chain_slots[i - 1]];
                const Pair& b = pairs[chain_slots[i]];
                const int32_t g = measure_gap(sk, a.e1, a.e2, b.b1, b.b2);
                res.gap[i] = g;
                res.gap_score[i] = score_gap(g);
            }
            if (subs[k].anchored && na) {
                const Pair& f = pairs[chain_slots.front()];
                const Pair& l = pairs[chain_slots.back()];
                res.gap[0] = gap_from_sources(k, f.b1, f.b2);
                res.gap_score[0] = score_gap((int32_t)res.gap[0]);
                res.gap[na] = gap_to_sinks(k, l.e1, l.e2);
                res.gap_score[na] = score_gap((int32_t)res.gap[na]);
            }
        }
    }
    release_index();
    tm.index_ms += index_ms;
    tm.traceback_ms += ms_since(T2) - index_ms;
    if (timing) fprintf(stderr, "[chain_dp_batch]   traceback: %llu steps, %llu equal-valued records scanned; split %.1f ortho %.1f by-shift %.1f gap-free trees %.1f candidates (incl. value index) %.1f ms\n",
                        (unsigned long long)n_steps, (unsigned long long)n_scanned, t_split, t_ortho, t_diag, t_gapfree, t_cand);
    lap("traceback");
    if (dp_out) {
        dp_out->resize(M);
        for (uint32_t slot = 0; slot < M; ++slot) (*dp_out)[slot] = dp_sorted[s_of_slot[slot]];
    }
    cleanup();
    lap("release");
    return CL_OK;
#undef CH
}

// one instance on the two merge graphs (with their sentinels); global anchoring = chains start at the nodes after the
// source sentinel and end at the nodes before the sink sentinel (anchorer.hpp:1069-1076)
static ChainSub whole_graph_instance(const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms, uint64_t num_match_sets,
                                     bool global_anchoring) {
    ChainSub sb;
    sb.g[0] = g1; sb.g[1] = g2;
    sb.ms = ms;
    sb.num_match_sets = num_match_sets;
    sb.anchored = global_anchoring;
    if (global_anchoring)
        for (int side = 0; side < 2; ++side) {
            const cl_base_graph* g = sb.g[side];
            sb.src[side].assign(g->next_idx + g->next_off[g->src_id], g->next_idx + g->next_off[g->src_id + 1]);
            sb.snk[side].assign(g->prev_idx + g->prev_off[g->snk_id], g->prev_idx + g->prev_off[g->snk_id + 1]);
        }
    return sb;
}

static int chain_dp_impl(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                         uint64_t num_match_sets, const cl_chain_params* cp, double local_scale, int want_dp, bool sparse,
                         cl_chain_result* out) {
    if (!ctx || !g1 || !g2 || !ms || !cp || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    if (num_match_sets > ms->n_sets) { cl_set_error(ctx, "num_match_sets exceeds the number of sets"); return CL_ERR_INVALID_ARGUMENT; }
    std::vector<ChainSub> subs(1, whole_graph_instance(g1, g2, ms, num_match_sets, cp->global_anchoring != 0));
    std::vector<ChainSubResult> res;
    ChainTimings tm;
    std::vector<float> dp;
    const int rc = chain_dp_batch(ctx, subs, cp, local_scale, sparse, res, tm, want_dp ? &dp : nullptr);
    if (rc) return rc;
    out->n_pairs = tm.n_pairs;
    out->device_ms = tm.device_ms; out->prep_ms = tm.prep_ms; out->index_ms = tm.index_ms; out->traceback_ms = tm.traceback_ms;
    const ChainSubResult& r = res[0];
    const size_t na = r.chain.size() / 3;
    out->n_anchors = na;
    out->n_ties = r.n_ties;
    out->anchors = (uint32_t*)malloc((na ? na : 1) * 3 * sizeof(uint32_t));
    if (want_dp) out->dp = (float*)malloc((dp.size() ? dp.size() : 1) * sizeof(float));
    if (!out->anchors || (want_dp && !out->dp)) { cl_chain_result_free(out); return CL_ERR_OUT_OF_MEMORY; }
    if (na) memcpy(out->anchors, r.chain.data(), r.chain.size() * sizeof(uint32_t));
    if (want_dp && !dp.empty()) memcpy(out->dp, dp.data(), dp.size() * sizeof(float));
    if (!sparse && subs[0].anchored && na) {
        out->gap_before_first = r.gap[0]; out->gap_score_before_first = r.gap_score[0];
        out->gap_after_last = r.gap[na]; out->gap_score_after_last = r.gap_score[na];
    }
    return CL_OK;
}

extern "C" {

int cl_chain_sparse_affine(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                           uint64_t num_match_sets, const cl_chain_params* cp, double local_scale, int want_dp,
                           cl_chain_result* out) {
    cl_bind_device(ctx);
    return chain_dp_impl(ctx, g1, g2, ms, num_match_sets, cp, local_scale, want_dp, false, out);
}

int cl_chain_sparse(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                    uint64_t num_match_sets, const cl_chain_params* cp, int want_dp, cl_chain_result* out) {
    cl_bind_device(ctx);
    return chain_dp_impl(ctx, g1, g2, ms, num_match_sets, cp, 1.0, want_dp, true, out);
}

}  // extern "C"

// exhaustive_chain_dp with score_edges == false (anchorer.hpp:1342-1509 as dispatched at :1229-1232) and
// AnchorGraph::heaviest_weight_path (src/anchorer.cpp:68-133).  The reference materialises every edge; here an edge is the
// reachability test itself, evaluated twice per ordered pair (once for the in-degrees of Kahn's algorithm, once when the
// source node is popped), so a few tens of thousands of match pairs stay within memory.  Node order, edge order (ascending
// target id), the LIFO stack of topological_order.hpp:12-60 and the strict '>' updates decide ties exactly as there.
extern "C" int cl_chain_exhaustive(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                                   uint64_t num_match_sets, const cl_chain_params* cp, cl_chain_result* out) {
    if (!g1 || !g2 || !ms || !cp || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    if (num_match_sets > ms->n_sets) { cl_set_error(ctx, "num_match_sets exceeds the number of sets"); return CL_ERR_INVALID_ARGUMENT; }
    clhost::PathMergeTable x1, x2;
    if (!x1.build(*g1) || !x2.build(*g2)) { cl_set_error(ctx, "graph is not acyclic"); return CL_ERR_CYCLIC_GRAPH; }
    const ChainSub sb = whole_graph_instance(g1, g2, ms, num_match_sets, cp->global_anchoring != 0);
    const double lowest = std::numeric_limits<double>::lowest();
    struct Node { uint32_t set, i1, i2, b1, e1, b2, e2; double weight, initial, final_w; };
    std::vector<Node> nodes;
    for (uint64_t s = 0; s < num_match_sets; ++s) {
        const uint64_t a0 = ms->set_off1[s], a1 = ms->set_off1[s + 1], c0 = ms->set_off2[s], c1 = ms->set_off2[s + 1];
        if (a0 == a1) continue;
        const double w = anchor_weight(*cp, ms->count1[s], ms->count2[s], ms->walk_off1[a0 + 1] - ms->walk_off1[a0], ms->full_length[s]);
        for (uint64_t j = a0; j < a1; ++j)
            for (uint64_t q = c0; q < c1; ++q) {
                Node nd{(uint32_t)s, (uint32_t)(j - a0), (uint32_t)(q - c0), ms->nodes1[ms->walk_off1[j]], ms->nodes1[ms->walk_off1[j + 1] - 1],
                        ms->nodes2[ms->walk_off2[q]], ms->nodes2[ms->walk_off2[q + 1] - 1], w, 0.0, 0.0};
                if (sb.anchored) {
                    nd.initial = lowest;
                    for (uint32_t a : sb.src[0]) for (uint32_t b : sb.src[1])
                        if ((a == nd.b1 || x1.reachable(a, nd.b1)) && (b == nd.b2 || x2.reachable(b, nd.b2))) nd.initial = 0.0;
                    nd.final_w = lowest;
                    for (uint32_t a : sb.snk[0]) for (uint32_t b : sb.snk[1])
                        if ((a == nd.e1 || x1.reachable(nd.e1, a)) && (b == nd.e2 || x2.reachable(nd.e2, b))) nd.final_w = 0.0;
                }
                nodes.push_back(nd);
            }
    }
    const size_t N = nodes.size();
    out->n_pairs = N;
    auto edge = [&](size_t i, size_t j) { return x1.reachable(nodes[i].e1, nodes[j].b1) && x2.reachable(nodes[i].e2, nodes[j].b2); };
    std::vector<uint32_t> indeg(N, 0);
    cl_parallel_for(N, [&](uint64_t jb, uint64_t je) {
        for (size_t j = jb; j < je; ++j) {
            uint32_t d = 0;
            for (size_t i = 0; i < N; ++i) d += edge(i, j) ? 1u : 0u;
            indeg[j] = d;
        }
    }, 64);
    std::vector<double> dp(N);
    std::vector<size_t> back(N, SIZE_MAX), stack;
    for (size_t i = 0; i < N; ++i) { dp[i] = nodes[i].initial; if (!indeg[i]) stack.push_back(i); }
    size_t max_id = SIZE_MAX, seen = 0;
    double max_weight = 0.0;   // min_score: edges are not scored
    while (!stack.empty()) {
        const size_t v = stack.back();
        stack.pop_back();
        ++seen;
        const bool live = dp[v] != lowest;
        if (live) {
            dp[v] += nodes[v].weight;
            if (nodes[v].final_w != lowest && dp[v] + nodes[v].final_w > max_weight) { max_id = v; max_weight = dp[v] + nodes[v].final_w; }
        }
        for (size_t j = 0; j < N; ++j) {
            if (!edge(v, j)) continue;
            if (live && dp[v] + 0.0 > dp[j]) { dp[j] = dp[v] + 0.0; back[j] = v; }
            if (--indeg[j] == 0) stack.push_back(j);
        }
    }
    if (seen != N) { cl_set_error(ctx, "anchor graph is not acyclic"); return CL_ERR_CYCLIC_GRAPH; }
    std::vector<size_t> path;
    for (size_t v = max_id; v != SIZE_MAX; v = back[v]) path.push_back(v);
    std::reverse(path.begin(), path.end());
    out->n_anchors = path.size();
    out->anchors = (uint32_t*)malloc((path.empty() ? 1 : path.size()) * 3 * sizeof(uint32_t));
    if (!out->anchors) return CL_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < path.size(); ++i) {
        out->anchors[3 * i] = nodes[path[i]].set;
        out->anchors[3 * i + 1] = nodes[path[i]].i1;
        out->anchors[3 * i + 2] = nodes[path[i]].i2;
    }
    return CL_OK;
}

// =====================================================================================================================
// Anchorer::anchor_chain without fill-in re-anchoring and branch splitting:
//   budgeted greedy match selection + reorder      include/centrolign/anchorer.hpp:1108-1173
//   graph swap to save memory                       :1175-1192, 1309-1322
//   estimate_score_scale                            :998-1047  (sparse_chain_dp + extract_graphs_between + source_sink_minmax)
//   sparse_affine_chain_dp with the estimated scale :985, 1050-1089
//   gap / score annotation of the chain             :2443-2468, 1331-1339
// =====================================================================================================================
namespace {

struct OwnedMatchSets {
    std::vector<uint64_t> set_off1{0}, walk_off1{0}, set_off2{0}, walk_off2{0}, count1, count2, full_length;
    std::vector<uint32_t> nodes1, nodes2;
    cl_match_sets view() const {
        return cl_match_sets{count1.size(), set_off1.data(), walk_off1.data(), nodes1.data(), set_off2.data(), walk_off2.data(),
                             nodes2.data(), count1.data(), count2.data(), full_length.data()};
    }
};

// anchorer.hpp:1108-1173 on a permutation of the original set indices: `cur` is the current order of the caller's
// vector; returns the number of leading sets that take part
uint64_t select_matches(const cl_match_sets& ms, const cl_chain_params& cp, std::vector<uint64_t>& cur, uint64_t local_max) {
    const size_t n = cur.size();
    auto n_pairs = [&](uint64_t s) { return (ms.set_off1[s + 1] - ms.set_off1[s]) * (ms.set_off2[s + 1] - ms.set_off2[s]); };
    uint64_t total = 0;
    for (uint64_t s : cur) total += n_pairs(s);
    if (total <= local_max) return n;
    // the sort key and the weight at the set's own length, evaluated once per set (two pow() each) on the pool's threads: 150 000 - 310 000 sets per merge of 10 x 1 Mbp
    std::vector<double> wfull(n), wlen(n);
    std::atomic<bool> plain{true};   // no NaN among the keys (anchor weights of counts >= 1 never are): the radix order below is then the comparison sort's
    cl_parallel_for(n, [&](uint64_t b, uint64_t e) {
        bool ok = true;
        for (uint64_t i = b; i < e; ++i) {
            const uint64_t s = cur[i];
            wfull[i] = anchor_weight(cp, ms.count1[s], ms.count2[s], ms.full_length[s], ms.full_length[s]);
            const uint64_t w0 = ms.set_off1[s];
            const uint64_t len = ms.walk_off1[w0 + 1] - ms.walk_off1[w0];
            wlen[i] = anchor_weight(cp, ms.count1[s], ms.count2[s], len, len);
            ok = ok && wfull[i] == wfull[i];
        }
        if (!ok) plain = false;
    }, 8192);
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    if (plain && n >= 4096 && n < (1ull << 32)) {
        // std::stable_sort(order, wfull[i] > wfull[j]) as a stable LSD radix sort on the order-preserving integer image of the doubles (descending; -0.0 == +0.0
        // as the comparison has it): four passes of 16 bits over the indices instead of n log n comparisons through two indirections
        std::vector<uint64_t> key(n);
        for (size_t i = 0; i < n; ++i) {
            double w = wfull[i];
            if (w == 0.0) w = 0.0;   // -0.0 -> +0.0
            uint64_t b;
            memcpy(&b, &w, 8);
            b = (b >> 63) ? ~b : b | 0x8000000000000000ull;   // ascending in the double's order ...
            key[i] = ~b;                                        // ... descending
        }
        std::vector<uint32_t> tmp(n);
        std::vector<uint32_t> cnt(65536 + 1);
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 16 * pass;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (size_t i = 0; i < n; ++i) ++cnt[((key[i] >> sh) & 0xFFFF) + 1];
            if (cnt[((key[0] >> sh) & 0xFFFF) + 1] == n) continue;   // every key has this digit
            for (size_t d = 0; d < 65536; ++d) cnt[d + 1] += cnt[d];
            for (size_t k = 0; k < n; ++k) { const uint32_t i = order[k]; tmp[cnt[(key[i] >> sh) & 0xFFFF]++] = i; }
            order.swap(tmp);
        }
    } else {
        std::stable_sort(order.begin(), order.end(), [&](uint32_t i, uint32_t j) { return wfull[i] > wfull[j]; });
    }
    size_t removed = 0;
    uint64_t left = local_max;
    for (size_t i = 0; i < order.size(); ++i) {
        const uint64_t s = cur[order[i]];
        if (wlen[order[i]] < 0.0) { removed += order.size() - i; break; }
        const uint64_t pc = n_pairs(s);
        if (left >= pc) { left -= pc; std::swap(order[i - removed], order[i]); }
        else ++removed;
    }
    std::vector<uint64_t> next(n);
    for (size_t k = 0; k < n; ++k) next[k] = cur[order[k]];  // reorder(matches, invert(order)): new[k] = old[order[k]]
    cur.swap(next);
    return n - removed;
}

// anchor_t on the host (include/centrolign/anchorer.hpp:36-57), walks in parent-graph ids
struct HAnchor {
    std::vector<uint32_t> w1, w2;
    uint64_t count1 = 0, count2 = 0, full_length = 0, match_set = 0, idx1 = 0, idx2 = 0;
    double score = 0.0, gsb = 0.0, gsa = 0.0;
    int64_t gb = 0, ga = 0;
};

// node -> ids of the paths through it, ascending (StepIndex, include/centrolign/step_index.hpp:38-46)
struct PathsOfNode {
    std::vector<uint64_t> off;
    std::vector<uint32_t> path;
    void build(const cl_base_graph& g) {
        off.assign(g.n_nodes + 1, 0);
        for (uint64_t i = 0; i < g.path_off[g.n_paths]; ++i) ++off[g.path_nodes[i] + 1];
        for (uint64_t v = 0; v < g.n_nodes; ++v) off[v + 1] += off[v];
        path.resize(off[g.n_nodes]);
        std::vector<uint64_t> fill(off.begin(), off.end() - 1);
        for (uint64_t p = 0; p < g.n_paths; ++p)
            for (uint64_t i = g.path_off[p]; i < g.path_off[p + 1]; ++i) path[fill[g.path_nodes[i]]++] = (uint32_t)p;
    }
};

// one side of one fill-in subproblem: the extracted subgraph with the parent's paths projected onto it
// (Extractor::project_paths / do_project, anchorer.hpp:586-615)
struct FillSide {
    std::vector<uint64_t> path_off{0};
    std::vector<uint32_t> path_nodes, tag;
    cl_base_graph view;
    void project(const clhost::OwnedBatch::Side& sd, uint64_t k, const PathsOfNode& steps, uint32_t tag_base) {
        const uint64_t b = sd.node_off[k], n = sd.node_off[k + 1] - b;
        view = cl_base_graph{n, sd.label.data() + b, sd.next_off.data() + b, sd.next_idx.data(), sd.prev_off.data() + b, sd.prev_idx.data(),
                             0, nullptr, nullptr, UINT64_MAX, UINT64_MAX};
        std::vector<uint32_t> order;
        clhost::topological_order(view, order);
        std::vector<std::vector<uint32_t>> paths;
        std::unordered_map<uint32_t, uint32_t> local_of;   // parent path -> projected path, numbered by first encounter
        for (uint32_t v : order) {
            const uint64_t parent = sd.back[b + v];
            for (uint64_t i = steps.off[parent]; i < steps.off[parent + 1]; ++i) {
                auto it = local_of.find(steps.path[i]);
                if (it == local_of.end()) {
                    it = local_of.emplace(steps.path[i], (uint32_t)paths.size()).first;
                    paths.emplace_back();
                    tag.push_back(tag_base + steps.path[i]);
                }
                paths[it->second].push_back(v);
            }
        }
        for (auto& pth : paths) {
            path_nodes.insert(path_nodes.end(), pth.begin(), pth.end());
            path_off.push_back(path_nodes.size());
        }
        view.n_paths = paths.size();
        view.path_off = path_off.data();
        view.path_nodes = path_nodes.data();
    }
};

}  // namespace

extern "C" {

void cl_anchor_chain_result_free(cl_anchor_chain_result* r) {
    if (!r) return;
    free(r->anchors); free(r->gap_before); free(r->gap_after); free(r->gap_score_before); free(r->gap_score_after);
    free(r->score); free(r->set_order); free(r->walk_off); free(r->walk1); free(r->walk2); free(r->count1); free(r->count2);
    free(r->full_length);
    memset(r, 0, sizeof(*r));
}

}  // extern "C"

// scale_only: stop after estimate_score_scale (out->scale is the only field set)
// masked matches: sorted (set, idx1, idx2) triples in the indexing of the caller's sets (anchorer.hpp:144)
struct MaskSet {
    std::vector<std::array<uint64_t, 3>> v;
    bool has(uint64_t s, uint64_t a, uint64_t b) const { return std::binary_search(v.begin(), v.end(), std::array<uint64_t, 3>{s, a, b}); }
};

static int anchor_chain_impl(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                             const cl_anchor_params* ap, cl_anchor_chain_result* out, bool scale_only, const MaskSet* mask = nullptr,
                             const double* override_scale = nullptr, bool keep_scale_chain = false) {
    if (!ctx || !g1 || !g2 || !ms || !ap || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    if (ctx->poisoned) { cl_set_error(ctx, "this context was given up after a merge-group wait expired: destroy it and make a new one"); return CL_ERR_HIP; }
    const cl_chain_params& cp = ap->chain;
    // the CLI's -g (Anchorer::chaining_algorithm): Sparse chains on ChainMerge tables, without a scale estimate (anchorer.hpp:975-984, core.hpp:350-357)
    const int algo = ap->chaining_algorithm_plus_one ? ap->chaining_algorithm_plus_one - 1 : 2;
    if (algo != 1 && algo != 2) { cl_set_error(ctx, "chaining algorithm %d is not offered by cl_anchor_chain (1 = Sparse, 2 = SparseAffine; Exhaustive: cl_chain_exhaustive)", algo); return CL_ERR_INVALID_ARGUMENT; }
    const bool sparse_algo = algo == 1;
    clhost::PathMergeTable own_x1, own_x2;
    const clhost::PathMergeTable* const sx1 = cl_shared_table(g1), * const sx2 = cl_shared_table(g2);   // cl_core_align's, when it is the caller
    if ((!sx1 && !(sparse_algo ? own_x1.build_chain_merge(*g1) : own_x1.build(*g1))) || (!sx2 && !(sparse_algo ? own_x2.build_chain_merge(*g2) : own_x2.build(*g2)))) {
        cl_set_error(ctx, "graph is not acyclic");
        return CL_ERR_CYCLIC_GRAPH;
    }
    const clhost::PathMergeTable& x1 = sx1 ? *sx1 : own_x1;
    const clhost::PathMergeTable& x2 = sx2 ? *sx2 : own_x2;
    // anchorer.hpp:1175: the DP runs with the graphs swapped when that makes its tables smaller
    const bool swap = g1->n_nodes * x1.chain_size() > g2->n_nodes * x2.chain_size();
    // CL_CHAIN_TIMING=1: host phase times on stderr
    const bool timing = getenv("CL_CHAIN_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char* what, std::chrono::steady_clock::time_point& t) {
        if (timing) fprintf(stderr, "[cl_anchor_chain] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now() - t).count());
        t = now();
    };
    auto t_pre = now();
    PostSwitchTable sw1, sw2;
    PathsOfNode steps1, steps2;
    GraphOrder order_dp1;   // of the graph that plays graph 1 in the DPs
    const bool small_graphs = (g1->n_nodes + 1) * (g1->n_paths + 1) + (g2->n_nodes + 1) * (g2->n_paths + 1) < (1u << 18);
    cl_pool_run(small_graphs ? 1 : 5, [&](unsigned t0) {
      for (unsigned t = t0; t < 5; t += small_graphs ? 1 : 5) {
        if (t == 0) sw1.build(*g1, x1);
        else if (t == 1) sw2.build(*g2, x2);
        else if (t == 4) order_dp1.build(swap ? *g2 : *g1);
        else if (!ap->do_fill_in_anchoring) continue;
        else if (t == 2) steps1.build(*g1);
        else steps2.build(*g2);
      }
    });
    lap("post-switch tables, paths of nodes", t_pre);
    std::vector<uint64_t> cur(ms->n_sets);
    std::iota(cur.begin(), cur.end(), (uint64_t)0);

    // walks of (position in `cur`, idx) on the caller's sets
    auto walk = [&](int side, uint64_t set_pos, uint32_t idx, const uint32_t*& b, const uint32_t*& e) {
        const uint64_t s = cur[set_pos];
        const uint64_t w = (side ? ms->set_off2 : ms->set_off1)[s] + idx;
        const uint64_t* wo = side ? ms->walk_off2 : ms->walk_off1;
        const uint32_t* nd = side ? ms->nodes2 : ms->nodes1;
        b = nd + wo[w]; e = nd + wo[w + 1];
    };

    // ---- fill_in_anchor_chain (anchorer.hpp:619-699): re-anchor inside every gap of the chain with the matches that lie
    //      wholly inside it; all the gaps' DPs run as one batched device pass
    auto fill_in = [&](std::vector<HAnchor>& anchors, bool sparse, double anchor_scale) -> int {
        if (anchors.empty()) return CL_OK;
        auto t = now();
        std::vector<uint64_t> seg_off{0, anchors.size()}, walk_off{0};
        std::vector<uint32_t> w1, w2;
        for (const HAnchor& a : anchors) {
            w1.insert(w1.end(), a.w1.begin(), a.w1.end());
            w2.insert(w2.end(), a.w2.begin(), a.w2.end());
            walk_off.push_back(w1.size());
        }
        cl_anchor_segments sg{1, seg_off.data(), walk_off.data(), w1.data(), w2.data()};
        clhost::OwnedBatch ob;
        int rc = clhost::extract_stitch_batch(*g1, *g2, sg, ob, &x1, &x2);
        if (rc) { cl_set_error(ctx, "extraction failed"); return rc; }
        lap("fill-in: extraction", t);
        const size_t K = ob.only_del.size();
        // divvy_matches (anchorer.hpp:701-798): a walk goes to the gap that holds both its ends
        const uint32_t none = 0xFFFFFFFFu;
        std::vector<std::pair<uint32_t, uint32_t>> ft[2];
        for (int side = 0; side < 2; ++side) {
            ft[side].assign((side ? g2 : g1)->n_nodes, std::make_pair(none, none));
            const auto& sd = ob.side[side];
            for (size_t k = 0; k < K; ++k)
                for (uint64_t v = sd.node_off[k]; v < sd.node_off[k + 1]; ++v) ft[side][sd.back[v]] = std::make_pair((uint32_t)k, (uint32_t)(v - sd.node_off[k]));
        }
        struct Divvied {
            OwnedMatchSets sets;
            std::vector<uint64_t> origin_set;
            std::vector<std::vector<uint32_t>> origin_idx1, origin_idx2;
        };
        std::vector<Divvied> dv(K);
        struct Pending { uint32_t k; std::vector<std::vector<uint32_t>> walks1, walks2; std::vector<uint32_t> idx1, idx2; };
        // the sets are independent: gather each set's walks per gap in parallel, then append to the gaps in set order
        std::vector<std::vector<Pending>> pend_of(cur.size());
        cl_parallel_for(cur.size(), [&](uint64_t pos_begin, uint64_t pos_end) {
        for (size_t pos = pos_begin; pos < pos_end; ++pos) {
            const uint64_t s = cur[pos];
            std::vector<Pending>& pend = pend_of[pos];
            for (uint64_t j = 0; j < ms->set_off1[s + 1] - ms->set_off1[s]; ++j) {
                const uint32_t *b, *e;
                walk(0, pos, (uint32_t)j, b, e);
                const uint32_t k = ft[0][b[0]].first;
                if (k == none || k != ft[0][e[-1]].first) continue;
                auto it = std::find_if(pend.begin(), pend.end(), [&](const Pending& q) { return q.k == k; });
                if (it == pend.end()) { pend.emplace_back(); pend.back().k = k; it = pend.end() - 1; }
                it->idx1.push_back((uint32_t)j);
                it->walks1.emplace_back();
                for (const uint32_t* v = b; v != e; ++v) it->walks1.back().push_back(ft[0][*v].second);
            }
            if (pend.empty()) continue;
            for (uint64_t j = 0; j < ms->set_off2[s + 1] - ms->set_off2[s]; ++j) {
                const uint32_t *b, *e;
                walk(1, pos, (uint32_t)j, b, e);
                const uint32_t k = ft[1][b[0]].first;
                auto it = std::find_if(pend.begin(), pend.end(), [&](const Pending& q) { return q.k == k; });
                if (it == pend.end() || k != ft[1][e[-1]].first) continue;
                it->idx2.push_back((uint32_t)j);
                it->walks2.emplace_back();
                for (const uint32_t* v = b; v != e; ++v) it->walks2.back().push_back(ft[1][*v].second);
            }
        }
        }, 2048);
        for (size_t pos = 0; pos < cur.size(); ++pos) {
            const uint64_t s = cur[pos];
            for (Pending& q : pend_of[pos]) {
                if (q.walks2.empty()) continue;
                Divvied& d = dv[q.k];
                for (auto& w : q.walks1) { d.sets.nodes1.insert(d.sets.nodes1.end(), w.begin(), w.end()); d.sets.walk_off1.push_back(d.sets.nodes1.size()); }
                d.sets.set_off1.push_back(d.sets.walk_off1.size() - 1);
                for (auto& w : q.walks2) { d.sets.nodes2.insert(d.sets.nodes2.end(), w.begin(), w.end()); d.sets.walk_off2.push_back(d.sets.nodes2.size()); }
                d.sets.set_off2.push_back(d.sets.walk_off2.size() - 1);
                d.sets.count1.push_back(ms->count1[s]);
                d.sets.count2.push_back(ms->count2[s]);
                d.sets.full_length.push_back(ms->full_length[s]);
                d.origin_set.push_back(pos);
                d.origin_idx1.push_back(std::move(q.idx1));
                d.origin_idx2.push_back(std::move(q.idx2));
            }
        }
        lap("fill-in: divvy", t);
        // assign_reanchor_budget (src/anchorer.cpp:136-154)
        uint64_t total = 0;
        auto matrix_size = [&](size_t k) { return (ob.side[0].node_off[k + 1] - ob.side[0].node_off[k] + 1) * (ob.side[1].node_off[k + 1] - ob.side[1].node_off[k] + 1); };
        for (size_t k = 0; k < K; ++k) total += matrix_size(k);
        // the instances: inner anchor_chain (anchorer.hpp:1091-1329) on the gap's subgraphs with their sources / sinks
        struct Inst { size_t k; bool swap; std::vector<uint64_t> order; cl_match_sets view; FillSide side[2]; };
        std::vector<std::unique_ptr<Inst>> inst;
        std::vector<ChainSub> subs;
        for (size_t k = 0; k < K; ++k) {
            if (dv[k].origin_set.empty()) continue;
            inst.emplace_back(new Inst());
            Inst& in = *inst.back();
            in.k = k;
            in.side[0].project(ob.side[0], k, steps1, 0);
            in.side[1].project(ob.side[1], k, steps2, (uint32_t)g1->n_paths);
            in.swap = in.side[0].view.n_nodes * in.side[0].view.n_paths > in.side[1].view.n_nodes * in.side[1].view.n_paths;
            in.view = dv[k].sets.view();
            in.order.resize(in.view.n_sets);
            std::iota(in.order.begin(), in.order.end(), (uint64_t)0);
            const uint64_t budget = (uint64_t)ceil((double)ap->max_num_match_pairs * (double)matrix_size(k) / (double)total);
            const uint64_t n_use = select_matches(in.view, cp, in.order, budget);
            ChainSub sb;
            sb.tableau = false;
            sb.chain_merge = sparse_algo;
            sb.ms = &in.view;
            sb.order = in.order.data();
            sb.swap_sides = in.swap;
            sb.num_match_sets = n_use;
            sb.anchored = true;
            if (mask) {   // a divvied pair is masked if the pair it came from is (anchorer.hpp:662-680)
                const Divvied* dk = &dv[k];
                const std::vector<uint64_t>* order_now = &cur;
                sb.masked = [mask, dk, order_now](uint64_t set, uint32_t i1, uint32_t i2) {
                    return mask->has((*order_now)[dk->origin_set[set]], dk->origin_idx1[set][i1], dk->origin_idx2[set][i2]);
                };
            }
            for (int d = 0; d < 2; ++d) {
                const int from = in.swap ? 1 - d : d;
                const auto& sd = ob.side[from];
                sb.g[d] = &in.side[from].view;
                sb.tag[d] = in.side[from].tag;
                sb.src[d].assign(sd.src_idx.begin() + sd.src_off[k], sd.src_idx.begin() + sd.src_off[k + 1]);
                sb.snk[d].assign(sd.snk_idx.begin() + sd.snk_off[k], sd.snk_idx.begin() + sd.snk_off[k + 1]);
            }
            subs.push_back(std::move(sb));
        }
        lap("fill-in: instances", t);
        std::vector<ChainSubResult> res;
        if (!subs.empty()) {
            ChainTimings tm;
            rc = chain_dp_batch(ctx, subs, &cp, anchor_scale, sparse, res, tm, nullptr);
            if (rc) return rc;
            out->fill_in_pairs += tm.n_pairs;
            out->fill_in_device_ms += tm.device_ms;
        }
        lap("fill-in: batched DP", t);
        // translate the gap chains back (inner un-swap, anchorer.hpp:1309-1322) and merge (src/anchorer.cpp:157-222)
        std::vector<std::vector<HAnchor>> fill(K);
        for (size_t q = 0; q < inst.size(); ++q) {
            const Inst& in = *inst[q];
            const ChainSubResult& r = res[q];
            out->n_ties += r.n_ties;
            const Divvied& d = dv[in.k];
            const size_t na = r.chain.size() / 3;
            for (size_t i = 0; i < na; ++i) {
                const uint32_t pos = r.chain[3 * i];
                uint32_t a = r.chain[3 * i + 1], b = r.chain[3 * i + 2];
                if (in.swap) std::swap(a, b);
                const uint64_t set = in.order[pos];
                HAnchor h;
                for (int side = 0; side < 2; ++side) {
                    const auto& so = side ? d.sets.set_off2 : d.sets.set_off1;
                    const auto& wo = side ? d.sets.walk_off2 : d.sets.walk_off1;
                    const auto& nd = side ? d.sets.nodes2 : d.sets.nodes1;
                    const uint64_t w = so[set] + (side ? b : a);
                    auto& dst = side ? h.w2 : h.w1;
                    for (uint64_t v = wo[w]; v < wo[w + 1]; ++v) dst.push_back((uint32_t)ob.side[side].back[ob.side[side].node_off[in.k] + nd[v]]);
                }
                h.count1 = d.sets.count1[set]; h.count2 = d.sets.count2[set]; h.full_length = d.sets.full_length[set];
                h.score = anchor_weight(cp, h.count1, h.count2, h.w1.size(), h.full_length);
                // identity: the reference indexes its origin table with the POSITION in the (possibly reordered) gap match
                // sets (src/anchorer.cpp:207-212), kept as is; out of range -> all ones
                h.match_set = d.origin_set[pos];
                h.idx1 = a < d.origin_idx1[pos].size() ? d.origin_idx1[pos][a] : UINT64_MAX;
                h.idx2 = b < d.origin_idx2[pos].size() ? d.origin_idx2[pos][b] : UINT64_MAX;
                if (!sparse) {
                    h.gb = in.swap ? -r.gap[i] : r.gap[i];
                    h.ga = in.swap ? -r.gap[i + 1] : r.gap[i + 1];
                    h.gsb = r.gap_score[i];
                    h.gsa = r.gap_score[i + 1];
                }
                fill[in.k].push_back(std::move(h));
            }
        }
        std::vector<HAnchor> merged;
        for (size_t k = 0; k < K; ++k) {
            if (k != 0) {
                HAnchor& a = anchors[k - 1];
                if (!merged.empty()) { a.gb = merged.back().ga; a.gsb = merged.back().gsa; }
                merged.push_back(std::move(a));
            }
            for (size_t j = 0; j < fill[k].size(); ++j) {
                if (j == 0 && !merged.empty()) { merged.back().gsa = fill[k][j].gsb; merged.back().ga = fill[k][j].gb; }
                merged.push_back(std::move(fill[k][j]));
            }
        }
        anchors.swap(merged);
        lap("fill-in: merge", t);
        return CL_OK;
    };

    // ---- anchor_chain with a given algorithm and scale (anchorer.hpp:1050-1089)
    auto run = [&](bool sparse, double anchor_scale, std::vector<HAnchor>& anchors) -> int {
        const uint64_t local_max = std::min<uint64_t>((uint64_t)llround((anchor_scale / ap->score_scale) * (double)ap->max_num_match_pairs),
                                                     ap->max_num_match_pairs);
        auto t = now();
        const uint64_t n_use = select_matches(*ms, cp, cur, local_max);
        lap(sparse ? "sparse: select" : "affine: select", t);
        std::vector<ChainSub> subs(1, whole_graph_instance(swap ? g2 : g1, swap ? g1 : g2, ms, n_use, cp.global_anchoring != 0));
        subs[0].order = cur.data();
        subs[0].swap_sides = swap;
        if (mask) subs[0].masked = [mask](uint64_t set, uint32_t i1, uint32_t i2) { return mask->has(set, i1, i2); };
        subs[0].x[0] = swap ? &x2 : &x1; subs[0].x[1] = swap ? &x1 : &x2;
        subs[0].sw[0] = swap ? &sw2 : &sw1; subs[0].sw[1] = swap ? &sw1 : &sw2;
        subs[0].order1 = &order_dp1;
        std::vector<ChainSubResult> res;
        ChainTimings tm;
        int rc = chain_dp_batch(ctx, subs, &cp, anchor_scale, sparse, res, tm, nullptr);
        if (rc) return rc;
        lap(sparse ? "sparse: chain DP" : "affine: chain DP", t);
        if (timing) fprintf(stderr, "[cl_anchor_chain]   prep %.1f device %.1f index %.1f traceback %.1f ms, %llu pairs, %u combinations\n", tm.prep_ms, tm.device_ms, tm.index_ms, tm.traceback_ms, (unsigned long long)tm.n_pairs, tm.n_combos);
        out->dp_device_ms += tm.device_ms;
        out->dp_pair_evals += tm.pair_evals;
        if (!sparse) { out->dp_match_pairs = tm.n_pairs; out->dp_combinations = tm.n_combos; }
        out->n_ties += res[0].n_ties;
        const ChainSubResult& r = res[0];
        const size_t na = r.chain.size() / 3;
        anchors.assign(na, HAnchor());
        cl_parallel_for(na, [&](uint64_t i_begin, uint64_t i_end) {
        for (size_t i = i_begin; i < i_end; ++i) {
            HAnchor& h = anchors[i];
            h.match_set = r.chain[3 * i];
            h.idx1 = r.chain[3 * i + 1];
            h.idx2 = r.chain[3 * i + 2];
            if (swap) std::swap(h.idx1, h.idx2);
            const uint32_t *b, *e;
            walk(0, h.match_set, (uint32_t)h.idx1, b, e);
            h.w1.assign(b, e);
            walk(1, h.match_set, (uint32_t)h.idx2, b, e);
            h.w2.assign(b, e);
            const uint64_t s = cur[h.match_set];
            h.count1 = ms->count1[s]; h.count2 = ms->count2[s]; h.full_length = ms->full_length[s];
            h.score = anchor_weight(cp, h.count1, h.count2, h.w1.size(), h.full_length);
            if (!sparse) {   // annotation in the DP's orientation, negated back at anchorer.hpp:1318-1320
                h.gb = swap ? -r.gap[i] : r.gap[i];
                h.ga = swap ? -r.gap[i + 1] : r.gap[i + 1];
                h.gsb = r.gap_score[i];
                h.gsa = r.gap_score[i + 1];
            }
        }
        }, 512);
        lap("anchors", t);
        if (ap->do_fill_in_anchoring) return fill_in(anchors, sparse, anchor_scale);
        return CL_OK;
    };

    // ---- estimate_score_scale (anchorer.hpp:998-1047)
    double scale = 1.0;
    int rc;
    std::vector<HAnchor> sc;   // the chain of the estimate: estimate_score_scale's chain_out (anchorer.hpp:1042-1044)
    if (override_scale && !scale_only) scale = *override_scale;   // anchorer.hpp:975-978
    else if (sparse_algo && !scale_only) scale = 1.0;             // (:979: the estimate only adjusts gap penalties, which Sparse does not have)
    else if (ap->autocalibrate_gap_penalties) {
        if ((rc = run(true, 1.0, sc))) return rc;
        auto t = now();
        double total_weight = 0.0;
        uint64_t total_length = 0;
        const size_t na = sc.size();
        std::vector<uint64_t> seg_off{0, na}, walk_off{0};
        std::vector<uint32_t> w1, w2;
        for (const HAnchor& a : sc) {
            total_weight += anchor_weight(cp, a.count1, a.count2, a.w1.size(), a.full_length);
            total_length += a.w1.size();
            w1.insert(w1.end(), a.w1.begin(), a.w1.end());
            w2.insert(w2.end(), a.w2.begin(), a.w2.end());
            walk_off.push_back(w1.size());
        }
        cl_anchor_segments sg{na ? 1u : 0u, seg_off.data(), walk_off.data(), w1.data(), w2.data()};
        clhost::OwnedBatch ob;
        if ((rc = clhost::extract_stitch_batch(*g1, *g2, sg, ob, &x1, &x2))) { cl_set_error(ctx, "extraction failed"); return rc; }
        for (uint64_t k = 0; k < ob.only_del.size(); ++k) {
            uint64_t fill = UINT64_MAX;
            for (int side = 0; side < 2; ++side) {
                const auto& sd = ob.side[side];
                if (sd.node_off[k + 1] == sd.node_off[k]) fill = 0;
                else fill = std::min<uint64_t>(fill, (uint64_t)min_source_sink(sd, k));
            }
            total_length += fill;
        }
        scale = total_weight / (double)total_length;
        lap("scale estimate: extraction", t);
    }
    out->scale = scale;
    if (scale_only && !keep_scale_chain) return CL_OK;

    // ---- the affine chain (or, for cl_leaf_calibrate, the estimate's own chain)
    std::vector<HAnchor> ch;
    if (scale_only) ch.swap(sc);
    else if ((rc = run(sparse_algo, scale, ch))) return rc;
    const size_t na = ch.size();
    uint64_t total_walk = 0;
    for (const HAnchor& a : ch) total_walk += a.w1.size();
    out->n_anchors = na;
    out->n_sets = cur.size();
    const size_t na1 = na ? na : 1;
    out->anchors = (uint64_t*)malloc(na1 * 3 * sizeof(uint64_t));
    out->gap_before = (int64_t*)calloc(na1, sizeof(int64_t));
    out->gap_after = (int64_t*)calloc(na1, sizeof(int64_t));
    out->gap_score_before = (double*)calloc(na1, sizeof(double));
    out->gap_score_after = (double*)calloc(na1, sizeof(double));
    out->score = (double*)calloc(na1, sizeof(double));
    out->count1 = (uint64_t*)calloc(na1, sizeof(uint64_t));
    out->count2 = (uint64_t*)calloc(na1, sizeof(uint64_t));
    out->full_length = (uint64_t*)calloc(na1, sizeof(uint64_t));
    out->walk_off = (uint64_t*)calloc(na + 1, sizeof(uint64_t));
    out->walk1 = (uint32_t*)malloc((total_walk ? total_walk : 1) * sizeof(uint32_t));
    out->walk2 = (uint32_t*)malloc((total_walk ? total_walk : 1) * sizeof(uint32_t));
    out->set_order = (uint64_t*)malloc((cur.size() ? cur.size() : 1) * sizeof(uint64_t));
    if (!out->anchors || !out->gap_before || !out->gap_after || !out->gap_score_before || !out->gap_score_after || !out->score ||
        !out->count1 || !out->count2 || !out->full_length || !out->walk_off || !out->walk1 || !out->walk2 || !out->set_order) {
        cl_anchor_chain_result_free(out);
        return CL_ERR_OUT_OF_MEMORY;
    }
    memcpy(out->set_order, cur.data(), cur.size() * sizeof(uint64_t));
    uint64_t wpos = 0;
    for (size_t i = 0; i < na; ++i) {
        const HAnchor& a = ch[i];
        out->anchors[3 * i] = a.match_set; out->anchors[3 * i + 1] = a.idx1; out->anchors[3 * i + 2] = a.idx2;
        out->gap_before[i] = a.gb; out->gap_after[i] = a.ga;
        out->gap_score_before[i] = a.gsb; out->gap_score_after[i] = a.gsa;
        out->score[i] = a.score;
        out->count1[i] = a.count1; out->count2[i] = a.count2; out->full_length[i] = a.full_length;
        memcpy(out->walk1 + wpos, a.w1.data(), a.w1.size() * sizeof(uint32_t));
        memcpy(out->walk2 + wpos, a.w2.data(), a.w2.size() * sizeof(uint32_t));
        wpos += a.w1.size();
        out->walk_off[i + 1] = wpos;
    }
    return CL_OK;
}

extern "C" {

int cl_anchor_chain(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                    const cl_anchor_params* ap, cl_anchor_chain_result* out) {
    return anchor_chain_impl(ctx, g1, g2, ms, ap, out, false);
}

int cl_anchor_chain_masked(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms, const cl_anchor_params* ap,
                           const uint64_t* masked, uint64_t n_masked, const double* override_scale, cl_anchor_chain_result* out) {
    cl_bind_device(ctx);
    if (n_masked && !masked) { cl_set_error(ctx, "null mask"); return CL_ERR_INVALID_ARGUMENT; }
    MaskSet mask;
    mask.v.resize(n_masked);
    for (uint64_t i = 0; i < n_masked; ++i) mask.v[i] = {masked[3 * i], masked[3 * i + 1], masked[3 * i + 2]};
    std::sort(mask.v.begin(), mask.v.end());
    mask.v.erase(std::unique(mask.v.begin(), mask.v.end()), mask.v.end());
    return anchor_chain_impl(ctx, g1, g2, ms, ap, out, false, n_masked ? &mask : nullptr, override_scale);
}

// Core::generate_diagonal_mask (src/core.cpp:301-321): in a self-comparison, the pairs of a set whose two walks start at the same node
int cl_generate_diagonal_mask(const cl_match_sets* ms, uint64_t** masked_out, uint64_t* n_masked_out) {
    if (!ms || !masked_out || !n_masked_out) return CL_ERR_INVALID_ARGUMENT;
    std::vector<uint64_t> m;
    std::unordered_map<uint32_t, uint64_t> start_to_idx;
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        start_to_idx.clear();
        for (uint64_t j = ms->set_off1[s]; j < ms->set_off1[s + 1]; ++j) start_to_idx[ms->nodes1[ms->walk_off1[j]]] = j - ms->set_off1[s];   // the last walk wins (:309-311)
        for (uint64_t k = ms->set_off2[s]; k < ms->set_off2[s + 1]; ++k) {
            auto it = start_to_idx.find(ms->nodes2[ms->walk_off2[k]]);
            if (it != start_to_idx.end()) { m.push_back(s); m.push_back(it->second); m.push_back(k - ms->set_off2[s]); }
        }
    }
    *n_masked_out = m.size() / 3;
    *masked_out = (uint64_t*)malloc((m.size() ? m.size() : 1) * sizeof(uint64_t));
    if (!*masked_out) return CL_ERR_OUT_OF_MEMORY;
    if (!m.empty()) memcpy(*masked_out, m.data(), m.size() * sizeof(uint64_t));
    return CL_OK;
}

// Core::update_mask (src/core.cpp:323-372): every pair (set, j, k) in which some position of walk1 j and the same position of walk2 k are a
// node pair of the chain (or its mirror image) is added to the mask.  The result is the union, sorted, without duplicates.
int cl_update_mask(const cl_match_sets* ms, uint64_t n_chain_pairs, const uint32_t* chain_walk1, const uint32_t* chain_walk2, int mask_reciprocal,
                   const uint64_t* masked, uint64_t n_masked, uint64_t** masked_out, uint64_t* n_masked_out) {
    if (!ms || !masked_out || !n_masked_out || (n_chain_pairs && (!chain_walk1 || !chain_walk2)) || (n_masked && !masked)) return CL_ERR_INVALID_ARGUMENT;
    std::unordered_map<uint32_t, uint32_t> paired;   // later pairs overwrite earlier ones (:330-337)
    for (uint64_t i = 0; i < n_chain_pairs; ++i) {
        paired[chain_walk1[i]] = chain_walk2[i];
        if (mask_reciprocal) paired[chain_walk2[i]] = chain_walk1[i];
    }
    std::vector<std::array<uint64_t, 3>> m(n_masked);
    for (uint64_t i = 0; i < n_masked; ++i) m[i] = {masked[3 * i], masked[3 * i + 1], masked[3 * i + 2]};
    std::unordered_map<uint64_t, std::vector<uint32_t>> walk2_node;   // (position << 32 | node) -> walk2 indexes
    for (uint64_t s = 0; s < ms->n_sets; ++s) {
        walk2_node.clear();
        const uint64_t n1 = ms->set_off1[s + 1] - ms->set_off1[s], n2 = ms->set_off2[s + 1] - ms->set_off2[s];
        for (uint64_t k = 0; k < n2; ++k) {
            const uint64_t w = ms->set_off2[s] + k;
            for (uint64_t l = ms->walk_off2[w]; l < ms->walk_off2[w + 1]; ++l) walk2_node[((l - ms->walk_off2[w]) << 32) | ms->nodes2[l]].push_back((uint32_t)k);
        }
        for (uint64_t j = 0; j < n1; ++j) {
            const uint64_t w = ms->set_off1[s] + j;
            for (uint64_t l = ms->walk_off1[w]; l < ms->walk_off1[w + 1]; ++l) {
                auto it = paired.find(ms->nodes1[l]);
                if (it == paired.end()) continue;
                auto it2 = walk2_node.find(((l - ms->walk_off1[w]) << 32) | it->second);
                if (it2 == walk2_node.end()) continue;
                for (uint32_t k : it2->second) m.push_back({s, j, (uint64_t)k});
            }
        }
    }
    std::sort(m.begin(), m.end());
    m.erase(std::unique(m.begin(), m.end()), m.end());
    *n_masked_out = m.size();
    *masked_out = (uint64_t*)malloc((m.size() ? m.size() : 1) * 3 * sizeof(uint64_t));
    if (!*masked_out) return CL_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < m.size(); ++i) { (*masked_out)[3 * i] = m[i][0]; (*masked_out)[3 * i + 1] = m[i][1]; (*masked_out)[3 * i + 2] = m[i][2]; }
    return CL_OK;
}

// Anchorer::estimate_score_scale with its chain_out (anchorer.hpp:998-1047): out->scale and the chain the estimate was made on
int cl_estimate_score_scale_chain(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                                  const cl_anchor_params* ap, cl_anchor_chain_result* out) {
    cl_bind_device(ctx);
    if (!ap || !out) return CL_ERR_INVALID_ARGUMENT;
    cl_anchor_params p = *ap;
    p.autocalibrate_gap_penalties = 1;
    return anchor_chain_impl(ctx, g1, g2, ms, &p, out, true, nullptr, nullptr, true);
}

int cl_estimate_score_scale(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms,
                            const cl_anchor_params* ap, double* scale_out) {
    cl_bind_device(ctx);
    if (!ap || !scale_out) return CL_ERR_INVALID_ARGUMENT;
    cl_anchor_params p = *ap;
    p.autocalibrate_gap_penalties = 1;
    cl_anchor_chain_result r;
    const int rc = anchor_chain_impl(ctx, g1, g2, ms, &p, &r, true);
    if (rc) return rc;
    *scale_out = r.scale;
    return CL_OK;
}

}  // extern "C"
