// cl_api.cpp — host side of the C ABI declared in include/centrolign_amd.h.
//
// Mirrors, for a whole batch of between-anchor subproblems at once, what the reference does per
// subproblem on one CPU thread:
//   Stitcher::subalign      (src/stitcher.cpp:24-78)              -> choose_num_pw(), translate in collect
//   Stitcher::do_alignment  (include/centrolign/stitcher.hpp:237-370) -> route_problem()
//   pure_deletion_alignment (include/centrolign/alignment.hpp:1178-1210) -> pure_deletion() (host, O(n))
//   po_poa                  (alignment.hpp:753-1163)              -> packed to rank space here, DP + traceback
//                                                                    on the GPU (popoa_kernels.hip)
// There is no CPU implementation of the DP in this library: without a gfx950 device the entry points
// return CL_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <unordered_map>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <thread>
#include <chrono>
#include <vector>

#include "cl_internal.hpp"
#include "popoa_device.h"
#include "stitch_host.hpp"
#include "wfa_host.hpp"

hipError_t cl_launch_popoa_lane(int W, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const uint32_t* plist, const ClScoreParams& P, uint32_t* lane_sync /* null: one workgroup per pair */, hipStream_t stream);
hipError_t cl_launch_popoa_sys(int npw, int block, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const uint32_t* plist,
                               const ClScoreParams& P, hipStream_t stream);
hipError_t cl_launch_popoa_strip(int npw, uint32_t threads, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const ClStripDevice& SD, const uint32_t* slist,
                                 const ClScoreParams& P, hipStream_t stream);
hipError_t cl_launch_popoa_general(int npw, int block, uint32_t n_blocks, uint32_t ring_bytes, const ClDeviceBatch& B,
                                   const uint32_t* plist, const ClScoreParams& P, hipStream_t stream);
hipError_t cl_launch_popoa_linear(int W, uint32_t n_blocks, const ClDeviceBatch& B, const uint32_t* plist,
                                  const ClScoreParams& P, hipStream_t stream);
size_t cl_linear_workspace_bytes(uint32_t nr, uint32_t nc, int npw, int R);
hipError_t cl_launch_popoa_linear_span(uint32_t n_blocks, const ClDeviceBatch& B, const uint32_t* plist, const ClScoreParams& P, uint32_t* sync, hipStream_t stream);
uint32_t cl_linear_span_groups(uint32_t nr);

namespace {

thread_local std::string g_error;

// test hook: CL_FORCE_GENERAL=1 in the environment routes chain x chain problems to the general kernel too
const bool g_force_general = [] { const char* e = getenv("CL_FORCE_GENERAL"); return e && *e == '1'; }();
const bool g_no_sys = [] { const char* e = getenv("CL_NO_SYS"); return e && *e == '1'; }();     // test hook: no systolic DAG kernel
// test hook / A-B, read at every plan creation (tests switch it inside one process): no register kernel for near-chain pairs (popoa_lane_kernel) — the pairs then take
// the kernels of rounds 1-4 (systolic, strips, rings), which stay the route of everything that is not a near-chain pair
// strips of a near-chain pair that one round of popoa_lane_kernel takes = waves of its workgroup: FOUR when the pairs have the compute units to themselves, THREE in a
// plan whose other launches share them.  Measured (profiles/r05_lane_waves_ab.txt): a pair alone, or 150 pairs of 512 x 512 alone on the device, run 20 % faster in rounds
// of four (fewer rounds: 441 x 433 0.66 against 0.80 ms) — but the timed step of 10 x 1 Mbp (116 000 subproblems, sixteen launches side by side) takes 2.23-2.31 ms with
// rounds of three against 2.32-2.43 ms with four: a workgroup's waves meet at a barrier per chunk of 32 steps and run at the pace of the slowest SIMD, and with the other
// kernels' waves on the same SIMDs three are held up less than four.  The plan's size stands for "shared": 2 048 subproblems and more.  CL_LANE_WAVES=3|4 forces one
static uint32_t lane_round_waves(uint64_t n_problems) {
    const char* e = getenv("CL_LANE_WAVES");
    if (e && (*e == '3' || *e == '4')) return (uint32_t)(*e - '0');
    return n_problems >= 2048 ? 3u : 4u;
}
// two chain pairs of 17-32 rows per wave (popoa_linear_duo_kernel): alone on the device 32 x 32 pairs run at 215 instead of 130 G cells/s — but the step of 10 x 1 Mbp
// gets SLOWER with it, 1.79-1.96 against 1.54-1.65 ms (five runs each, profiles/r05_step_spread.txt): the 17 000 such pairs cost the one-pair-per-wave launch nothing
// there (it lasts as long as its longest thin pair, 0.43 ms with or without them) and their workgroups fill idle SIMD slots beside the latency-bound launches; taken out,
// they are one launch more in the step.  So: in plans that have the device to themselves (fewer than 2 048 subproblems), as the register kernel's rounds of four.
// ... or in which such pairs are at least half of everything (a batch of them alone).  CL_LINEAR_DUOS=0|1 forces.  (-1: decided once the plan is packed)
static int duos_forced() { const char* e = getenv("CL_LINEAR_DUOS"); return e && (*e == '0' || *e == '1') ? *e - '0' : -1; }
static bool no_lane_now() { const char* e = getenv("CL_NO_LANE"); return e && *e == '1'; }
#define g_no_lane no_lane_now()
constexpr uint64_t kSysLdsBytes = 159 * 1024;    // LDS a systolic-DAG workgroup may take (rings of every row + the column records)
const bool g_no_strip = [] { const char* e = getenv("CL_NO_STRIP"); return e && *e == '1'; }();   // test hook: no strips of rows for large branching pairs (popoa_strip_kernel)
const bool g_no_ring = [] { const char* e = getenv("CL_NO_RING"); return e && *e == '1'; }();   // test hook: HBM-plane general kernel only
constexpr uint64_t kRingLdsBytes = 128 * 1024;  // LDS a general-kernel workgroup may take for its anti-diagonal ring (160 KB per CU); launches are split at 64 KB
// CL_STITCH_GRAPH=1 replays a captured hipGraph of a plan's launches instead of launching the kernels directly (rounds 1-3 did; measured in
// round 4 on the 10 x 1 Mbp batches: one plan of 16 launches 4.24 ms per step through the graph, 3.12 ms launched directly — the graph's branches do
// not start together); CL_NO_GRAPH=1 is the old name of "off"
const bool g_no_graph = [] { const char* e = getenv("CL_STITCH_GRAPH"); return !(e && *e == '1'); }();

}  // namespace

ClFallbackCounters cl_fallbacks;
std::atomic<size_t> cl_pinned_total{0};

// ClRawVec's storage (cl_internal.hpp): blocks of a megabyte and more come from, and go back to, a per-thread cache (at most kBigCacheCap bytes
// per thread, best fit within 2x); a 64-byte header in front of every block says how large it is
namespace {
constexpr size_t kBigMin = 1u << 20, kBigCacheCap = 6ull << 30;
struct BigHeader { size_t capacity; void* map; size_t map_bytes; };   // map: the block's own mapping (null: malloc'ed)
static_assert(sizeof(BigHeader) <= 64, "header");
void big_release(void* h) noexcept {
    BigHeader* hd = static_cast<BigHeader*>(h);
    if (hd->map) (void)munmap(hd->map, hd->map_bytes); else free(h);
}
// (over all threads of the process the caches keep at most kBigCacheProcessCap: MSA workers and pool threads each have one, and a thread that has
// cached its share after one large merge may never need it again)
constexpr size_t kBigCacheProcessCap = 24ull << 30;
std::atomic<size_t> g_big_cached{0};
struct BigCache {
    std::multimap<size_t, void*> free_blocks;   // capacity -> header
    size_t bytes = 0;
    ~BigCache() { for (auto& b : free_blocks) { g_big_cached -= b.first; big_release(b.second); } }
};
thread_local BigCache t_big_cache;
}
static void* big_alloc_raw(size_t bytes);
// CL_BIG_POISON=1 (tests): every block is handed out filled with 0xA5 — code that relied on fresh mappings being zero shows at once instead of
// after a block has been used before
void* cl_big_alloc(size_t bytes) {
    static const bool poison = getenv("CL_BIG_POISON") != nullptr;
    void* p = big_alloc_raw(bytes);
    if (poison && bytes) memset(p, 0xA5, bytes);
    return p;
}
static void* big_alloc_raw(size_t bytes) {
    if (bytes >= kBigMin) {
        auto it = t_big_cache.free_blocks.lower_bound(bytes);
        if (it != t_big_cache.free_blocks.end() && it->first <= 2 * bytes) {
            void* h = it->second;
            t_big_cache.bytes -= it->first;
            g_big_cached -= it->first;
            t_big_cache.free_blocks.erase(it);
            return static_cast<char*>(h) + 64;
        }
    }
    if (bytes >= kBigMin) {
        // a mapping of its own, advised to use huge pages: first touch then costs a fault per 2 MB instead of per 4 KB (34 ms instead of
        // 120 ms per 900 MB on the MI355X host, scripts/dev/pin_bench.cpp) — and faults take the address-space lock the other workers need
        constexpr size_t kHuge = 2u << 20;
        const size_t len = ((bytes + 64 + kHuge - 1) & ~(kHuge - 1)) + kHuge;
        void* map = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map == MAP_FAILED) throw std::bad_alloc();
        char* h = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(map) + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
        (void)madvise(h, len - kHuge, MADV_HUGEPAGE);
        BigHeader* hd = reinterpret_cast<BigHeader*>(h);
        hd->capacity = bytes; hd->map = map; hd->map_bytes = len;
        return h + 64;
    }
    void* h = malloc(bytes + 64);
    if (!h) throw std::bad_alloc();
    BigHeader* hd = static_cast<BigHeader*>(h);
    hd->capacity = bytes; hd->map = nullptr; hd->map_bytes = 0;
    return static_cast<char*>(h) + 64;
}
void cl_big_free(void* p) noexcept {
    if (!p) return;
    void* h = static_cast<char*>(p) - 64;
    const size_t cap = static_cast<BigHeader*>(h)->capacity;
    if (cap >= kBigMin && t_big_cache.bytes + cap <= kBigCacheCap && g_big_cached.load(std::memory_order_relaxed) + cap <= kBigCacheProcessCap) {
        try { t_big_cache.free_blocks.emplace(cap, h); t_big_cache.bytes += cap; g_big_cached += cap; return; } catch (...) {}
    }
    big_release(h);
}

// A context keeps up to seven streams busy (the chaining DP: walk, two far launches, the sealing stream, copies) and an MSA runs several worker
// contexts; HIP's default of 4 hardware queues per process makes their launches queue up behind one another.  Measured on the nine concurrent
// stitch plans of 10 x 1 Mbp (ms per pass): 4 queues 8.9, 16: 4.9, 20: 3.8, 24: 11.5 (beyond ~23 the queues are time-sliced).  The runtime reads the
// variable when it initialises, i.e. at the first HIP call of the process: setting it when this library is loaded is early enough unless the
// host application has already used HIP (then it keeps what it has; set GPU_MAX_HW_QUEUES yourself).  Never overrides the user.
__attribute__((constructor)) static void cl_library_loaded() { setenv("GPU_MAX_HW_QUEUES", "20", 0); }

void cl_set_error(cl_context* ctx, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    if (ctx) {   // worker threads of cl_msa report into the caller's context: one writer at a time
        static std::mutex error_mutex;
        std::lock_guard<std::mutex> lock(error_mutex);
        ctx->error = buf;
    }
}
#define set_error cl_set_error

namespace {

// ---- one graph of one problem, as a view into the flat batch -------------------------------------------
struct GraphView {
    uint64_t n = 0;
    const uint8_t* label = nullptr;
    const uint64_t* prev_off = nullptr;  // n+1 entries, absolute into prev_idx
    const uint32_t* prev_idx = nullptr;
    const uint64_t* next_off = nullptr;  // may be null
    const uint32_t* next_idx = nullptr;
    uint64_t n_src = 0, n_snk = 0;
    const uint32_t* src = nullptr;
    const uint32_t* snk = nullptr;
};

GraphView view(const cl_graph_side& s, uint64_t k) {
    GraphView g;
    uint64_t b = s.node_off[k];
    g.n = s.node_off[k + 1] - b;
    g.label = s.label + b;
    g.prev_off = s.prev_off + b;
    g.prev_idx = s.prev_idx;
    g.next_off = s.next_off ? s.next_off + b : nullptr;
    g.next_idx = s.next_idx;
    g.n_src = s.src_off[k + 1] - s.src_off[k];
    g.src = s.src_idx + s.src_off[k];
    g.n_snk = s.snk_off[k + 1] - s.snk_off[k];
    g.snk = s.snk_idx + s.snk_off[k];
    return g;
}

int validate_side(cl_context* ctx, const cl_graph_side& s, uint64_t n, int which) {
    if (!s.node_off || !s.prev_off || !s.src_off || !s.snk_off) {
        set_error(ctx, "graph side %d: missing offset array", which);
        return CL_ERR_INVALID_ARGUMENT;
    }
    // (the offsets are absolute positions in the side's arrays; they need not start at 0: run_whole hands chunks of a batch on as views)
    for (uint64_t k = 0; k < n; ++k) {
        if (s.node_off[k + 1] < s.node_off[k] || s.src_off[k + 1] < s.src_off[k] || s.snk_off[k + 1] < s.snk_off[k]) {
            set_error(ctx, "graph side %d: offsets not monotone at problem %llu", which, (unsigned long long)k);
            return CL_ERR_INVALID_ARGUMENT;
        }
        uint64_t nn = s.node_off[k + 1] - s.node_off[k];
        if (nn >= (1ull << 31)) {
            set_error(ctx, "graph side %d: problem %llu has too many nodes", which, (unsigned long long)k);
            return CL_ERR_INVALID_ARGUMENT;
        }
        for (uint64_t i = s.src_off[k]; i < s.src_off[k + 1]; ++i)
            if (s.src_idx[i] >= nn) { set_error(ctx, "graph side %d: source id out of range in problem %llu", which, (unsigned long long)k); return CL_ERR_INVALID_ARGUMENT; }
        for (uint64_t i = s.snk_off[k]; i < s.snk_off[k + 1]; ++i)
            if (s.snk_idx[i] >= nn) { set_error(ctx, "graph side %d: sink id out of range in problem %llu", which, (unsigned long long)k); return CL_ERR_INVALID_ARGUMENT; }
        for (uint64_t v = s.node_off[k]; v < s.node_off[k + 1]; ++v) {
            if (s.prev_off[v + 1] < s.prev_off[v]) { set_error(ctx, "graph side %d: prev_off not monotone", which); return CL_ERR_INVALID_ARGUMENT; }
            for (uint64_t e = s.prev_off[v]; e < s.prev_off[v + 1]; ++e)
                if (s.prev_idx[e] >= nn) { set_error(ctx, "graph side %d: predecessor id out of range in problem %llu", which, (unsigned long long)k); return CL_ERR_INVALID_ARGUMENT; }
        }
    }
    return CL_OK;
}

// next lists (ids local), derived from prev lists when the caller did not pass them.  Only the order inside
// a list could differ from BaseGraph's, and nothing downstream depends on it.
struct NextLists {
    std::vector<uint64_t> off;
    std::vector<uint32_t> idx;
    const uint64_t* o = nullptr;
    const uint32_t* i = nullptr;
    uint64_t base = 0;  // o[v] - base indexes i when derived; absolute otherwise
    void build(const GraphView& g) {
        if (g.next_off) { o = g.next_off; i = g.next_idx; base = 0; return; }
        uint64_t n = g.n;
        off.assign(n + 2, 0);
        for (uint64_t v = 0; v < n; ++v)
            for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) off[g.prev_idx[e] + 2]++;
        for (uint64_t v = 0; v < n; ++v) off[v + 2] += off[v + 1];
        idx.resize(off[n + 1] ? off[n + 1] : 1);
        for (uint64_t v = 0; v < n; ++v)
            for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) idx[off[g.prev_idx[e] + 1]++] = (uint32_t)v;
        o = off.data(); i = idx.data(); base = 0;
    }
    uint64_t begin(uint64_t v) const { return o[v]; }
    uint64_t end(uint64_t v) const { return o[v + 1]; }
};

// Kahn's algorithm, LIFO stack seeded in ascending id order (topological_order.hpp:12-60).
// false if the graph has a cycle.
bool topological_order(const GraphView& g, const NextLists& nx, std::vector<uint32_t>& order,
                       std::vector<uint32_t>& scratch_stack, std::vector<uint32_t>& indeg) {
    uint64_t n = g.n;
    order.clear();
    order.reserve(n);
    scratch_stack.clear();
    indeg.resize(n);
    for (uint64_t v = 0; v < n; ++v) {
        indeg[v] = (uint32_t)(g.prev_off[v + 1] - g.prev_off[v]);
        if (indeg[v] == 0) scratch_stack.push_back((uint32_t)v);
    }
    while (!scratch_stack.empty()) {
        uint32_t v = scratch_stack.back();
        scratch_stack.pop_back();
        order.push_back(v);
        for (uint64_t e = nx.begin(v); e < nx.end(v); ++e) {
            uint32_t w = nx.i[e];
            if (--indeg[w] == 0) scratch_stack.push_back(w);
        }
    }
    return order.size() == n;
}

// Which topological order the DEVICE ranks a subgraph's nodes by.  Nothing in the result depends on it: the DP's values are those of any topological order and every
// tie of the traceback is decided by the order of a node's previous() list and of the sink list, both of which are kept as they come.  What depends
// on it is how far back a row (column) of the matrix reads.  The reference's order (Kahn with a LIFO stack, topological_order.hpp:12-60) lays the branches of a bubble
// out one after the other: the second branch's first node and the node that closes the bubble read a whole branch back — a "far fork", which the register kernel
// cannot take, which deepens the systolic kernels' rings by the branch length, and which on BOTH sides of a large pair left only the anti-diagonal sweep.  The LEVEL
// order (by the longest path from a source, ties in the reference's order) interleaves the branches: inside a bubble of two branches every node reads two ranks back,
// whatever the branches' length; what remains far is the difference of the branches' lengths, once, at the closing node.  Per graph the order with fewer far reads
// (more than four ranks back: the register kernel's reach) wins, then the one with the smaller reach, then the reference's — so chains and graphs whose bubbles are
// single nodes keep the order they had.  CL_RANK_ORDER=lifo|level forces one (measurements, tests).  `order` / `rank` come in as the reference's order and leave as the choice.
// (read when a plan is made, so that a test can make plans both ways)
int rank_order_now() { const char* e = getenv("CL_RANK_ORDER"); return !e ? 0 : !strcmp(e, "lifo") ? 1 : !strcmp(e, "level") ? 2 : 0; }
void choose_rank_order(const GraphView& g, int g_rank_order, std::vector<uint32_t>& order, std::vector<uint32_t>& rank, std::vector<uint32_t>& level, std::vector<uint32_t>& scratch) {
    const uint64_t n = g.n;
    if (g_rank_order == 1 || n < 4) return;
    auto reach = [&](uint64_t& far, uint64_t& longest) {
        far = longest = 0;
        for (uint64_t v = 0; v < n; ++v)
            for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) {
                const uint64_t back = rank[v] - rank[g.prev_idx[e]];
                far += back > 4;
                longest = std::max(longest, back);
            }
        for (uint64_t i = 0; i < g.n_src; ++i) {   // a source reads the boundary, index 0
            const uint64_t back = (uint64_t)rank[g.src[i]] + 1;
            far += back > 4;
            longest = std::max(longest, back);
        }
    };
    uint64_t far_ref, longest_ref;
    reach(far_ref, longest_ref);
    if (longest_ref <= 1 + (g.n_src > 1) && g_rank_order != 2) return;   // a chain
    level.assign(n, 0);
    uint32_t deepest = 0;
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t v = order[r];
        uint32_t l = 0;
        for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) l = std::max(l, level[g.prev_idx[e]] + 1);
        level[v] = l;
        deepest = std::max(deepest, l);
    }
    // counting sort by level, stable in the reference's order
    scratch.assign((size_t)deepest + 2, 0);
    for (uint64_t v = 0; v < n; ++v) ++scratch[level[v] + 1];
    for (uint32_t l = 0; l <= deepest; ++l) scratch[l + 1] += scratch[l];
    std::vector<uint32_t> by_level(n);
    for (uint64_t r = 0; r < n; ++r) by_level[scratch[level[order[r]]]++] = order[r];
    std::vector<uint32_t> rank_ref(rank);
    for (uint32_t r = 0; r < n; ++r) rank[by_level[r]] = r;
    uint64_t far_lvl, longest_lvl;
    reach(far_lvl, longest_lvl);
    if (g_rank_order == 2 || far_lvl < far_ref || (far_lvl == far_ref && longest_lvl < longest_ref)) order.swap(by_level);
    else rank.swap(rank_ref);
}

// src/stitcher.cpp:31-52
int choose_num_pw(uint64_t n1, uint64_t n2, const cl_align_params& p) {
    uint64_t cutoffs[2];
    for (int i = 1; i < 3; ++i) {
        if (p.gap_open[i - 1] > p.gap_open[i] || p.gap_extend[i - 1] < p.gap_extend[i]) return CL_ERR_BAD_GAP_PARAMS;
        uint32_t diff_open = p.gap_open[i] - p.gap_open[i - 1];
        uint32_t diff_extend = p.gap_extend[i - 1] - p.gap_extend[i];
        if (diff_extend == 0) return CL_ERR_BAD_GAP_PARAMS;  // the reference would divide by zero
        cutoffs[i - 1] = (diff_open + diff_extend - 1) / diff_extend;
    }
    int c = 0;
    while (c < 2 && n1 > cutoffs[c] && n2 > cutoffs[c]) ++c;
    return c + 1;
}

// Extractor::source_sink_minmax (src/anchorer.cpp:14-23) over minmax_distance (minmax_distance.hpp:16-72)
bool source_sink_minmax(const GraphView& g, int64_t& mn_out, int64_t& mx_out) {
    NextLists nx;
    nx.build(g);
    std::vector<uint32_t> order, st, indeg;
    if (!topological_order(g, nx, order, st, indeg)) return false;
    const int64_t INF = std::numeric_limits<int64_t>::max();
    std::vector<int64_t> mn(g.n, INF), mx(g.n, -1);
    for (uint64_t i = 0; i < g.n_src; ++i) { mn[g.src[i]] = 0; mx[g.src[i]] = 0; }
    for (uint32_t v : order) {
        if (mn[v] == INF) continue;
        for (uint64_t e = nx.begin(v); e < nx.end(v); ++e) {
            uint32_t w = nx.i[e];
            mn[w] = std::min(mn[w], mn[v] + 1);
            mx[w] = std::max(mx[w], mx[v] + 1);
        }
    }
    mn_out = INF;
    mx_out = -1;
    for (uint64_t i = 0; i < g.n_snk; ++i) {
        mn_out = std::min(mn_out, mn[g.snk[i]]);
        mx_out = std::max(mx_out, mx[g.snk[i]]);
    }
    return true;
}

// include/centrolign/stitcher.hpp:268-360
int route_problem(const GraphView& g1, const GraphView& g2, bool only_del, const cl_stitch_params& sp) {
    if (g2.n == 0) return CL_ROUTE_PURE_DELETION_1;
    if (g1.n == 0) return CL_ROUTE_PURE_DELETION_2;
    uint64_t mat = (g1.n + 1) * (g2.n + 1);
    if (mat <= sp.min_wfa_size && (!only_del || mat <= sp.max_trivial_size)) return CL_ROUTE_PO_POA;
    int64_t a1, b1, a2, b2;
    if (!source_sink_minmax(g1, a1, b1) || !source_sink_minmax(g2, a2, b2)) return CL_ERR_CYCLIC_GRAPH;
    uint64_t min1 = (uint64_t)a1, max1 = (uint64_t)b1, min2 = (uint64_t)a2, max2 = (uint64_t)b2;  // size_t in the reference
    if (max1 * sp.deletion_alignment_ratio <= min2 && max1 <= sp.deletion_alignment_short_max_size &&
        min2 >= sp.deletion_alignment_long_min_size)
        return CL_ROUTE_DELETION_WFA_1;
    if (max2 * sp.deletion_alignment_ratio <= min1 && max2 <= sp.deletion_alignment_short_max_size &&
        min1 >= sp.deletion_alignment_long_min_size)
        return CL_ROUTE_DELETION_WFA_2;
    double r = sp.max_wfa_ratio;
    if (mat < sp.max_wfa_size &&
        ((min2 * r >= min1 && min2 <= max1 * r) || (max2 * r >= min1 && max2 <= max1 * r) ||
         (min1 * r >= min2 && min1 <= max2 * r) || (max1 * r >= min2 && max1 <= max2 * r)) &&
        !only_del)
        return CL_ROUTE_PWFA;
    return CL_ROUTE_GREEDY_PARTIAL;
}

// pure_deletion_alignment (alignment.hpp:1178-1210): shortest source->sink path (shortest_path.hpp:32-100)
// emitted as a run of gaps.  Returns the path in LOCAL node ids.
int pure_deletion(const GraphView& g, std::vector<uint32_t>& path) {
    path.clear();
    if (g.n == 0) return CL_OK;
    NextLists nx;
    nx.build(g);
    std::vector<uint32_t> order, st, indeg;
    if (!topological_order(g, nx, order, st, indeg)) return CL_ERR_CYCLIC_GRAPH;
    const uint64_t INF = (uint64_t)std::numeric_limits<int64_t>::max();
    std::vector<uint64_t> dp(g.n, INF);
    for (uint64_t i = 0; i < g.n_src; ++i) dp[g.src[i]] = 0;
    for (uint32_t v : order) {
        uint64_t thru = dp[v] + 1;
        for (uint64_t e = nx.begin(v); e < nx.end(v); ++e) dp[nx.i[e]] = std::min(dp[nx.i[e]], thru);
    }
    uint64_t best = UINT64_MAX;
    for (uint64_t i = 0; i < g.n_snk; ++i) {
        uint32_t v = g.snk[i];
        if (dp[v] != INF && (best == UINT64_MAX || dp[v] < dp[best])) best = v;
    }
    if (best == UINT64_MAX) return CL_OK;
    uint64_t cur = best;
    path.push_back((uint32_t)cur);
    while (dp[cur] != 0) {
        uint64_t nxt = UINT64_MAX;
        for (uint64_t e = g.prev_off[cur]; e < g.prev_off[cur + 1]; ++e)
            if (dp[g.prev_idx[e]] + 1 == dp[cur]) { nxt = g.prev_idx[e]; break; }
        if (nxt == UINT64_MAX) return CL_ERR_INVALID_ARGUMENT;
        cur = nxt;
        path.push_back((uint32_t)cur);
    }
    std::reverse(path.begin(), path.end());
    return CL_OK;
}

// ---- greedy_partial_alignment (include/centrolign/alignment.hpp:1212-1611): do_alignment's route for gaps that look
// unalignable (stitcher.hpp:340-357).  Host algorithm in the reference as well: exact-match paths grown greedily (DFS over
// label-matching node pairs) from the sources and from the sinks, joined by a double deletion along shortest paths;
// overlapping or mutually unreachable match paths are trimmed by bisection on the total trim.
struct GreedyGraph {
    const GraphView* g;
    NextLists nx;
    std::vector<uint32_t> order;
    std::vector<uint64_t> dp;
    bool init(const GraphView& gv) {
        g = &gv;
        nx.build(gv);
        std::vector<uint32_t> st, indeg;
        dp.resize(gv.n);
        return topological_order(gv, nx, order, st, indeg);
    }
    // shortest_path between node sets (shortest_path.hpp:32-100), empty if there is none
    std::vector<uint64_t> shortest(const std::vector<uint64_t>& from, const std::vector<uint64_t>& to) {
        const uint64_t INF = (uint64_t)std::numeric_limits<int64_t>::max();
        std::fill(dp.begin(), dp.end(), INF);
        for (uint64_t v : from) dp[v] = 0;
        for (uint32_t v : order) {
            const uint64_t thru = dp[v] + 1;
            for (uint64_t e = nx.begin(v); e < nx.end(v); ++e) dp[nx.i[e]] = std::min(dp[nx.i[e]], thru);
        }
        uint64_t best = UINT64_MAX;
        for (uint64_t v : to)
            if (dp[v] != INF && (best == UINT64_MAX || dp[v] < dp[best])) best = v;
        std::vector<uint64_t> path;
        if (best == UINT64_MAX) return path;
        path.push_back(best);
        while (dp[path.back()] != 0) {
            const uint64_t cur = path.back();
            uint64_t nxt = UINT64_MAX;
            for (uint64_t e = g->prev_off[cur]; e < g->prev_off[cur + 1]; ++e)
                if (dp[g->prev_idx[e]] + 1 == dp[cur]) { nxt = g->prev_idx[e]; break; }
            if (nxt == UINT64_MAX) break;
            path.push_back(nxt);
        }
        std::reverse(path.begin(), path.end());
        return path;
    }
    bool reaches(uint64_t from, uint64_t to) { return !shortest(std::vector<uint64_t>(1, from), std::vector<uint64_t>(1, to)).empty(); }
};

typedef std::vector<std::pair<uint64_t, uint64_t>> HostAlignment;   // local node ids, CL_GAP for a gap

int greedy_partial_alignment(const GraphView& gv1, const GraphView& gv2, HostAlignment& out) {
    out.clear();
    GreedyGraph g1, g2;
    if (!g1.init(gv1) || !g2.init(gv2)) return CL_ERR_CYCLIC_GRAPH;
    const uint64_t n2 = gv2.n;
    const std::vector<uint64_t> sources1(gv1.src, gv1.src + gv1.n_src), sources2(gv2.src, gv2.src + gv2.n_src);
    const std::vector<uint64_t> sinks1(gv1.snk, gv1.snk + gv1.n_snk), sinks2(gv2.snk, gv2.snk + gv2.n_snk);
    HostAlignment aln_fwd, aln_rev;
    for (int dir = 0; dir < 2; ++dir) {
        const bool forward = dir == 0;
        size_t max_len = 0;
        uint64_t end_key = UINT64_MAX;
        std::unordered_map<uint64_t, uint64_t> back;   // pair key -> predecessor key (UINT64_MAX at a start)
        struct Item { uint64_t a, b; size_t len; };
        std::vector<Item> stack;
        for (uint64_t a : (forward ? sources1 : sinks1))
            for (uint64_t b : (forward ? sources2 : sinks2))
                if (gv1.label[a] == gv2.label[b]) {
                    stack.push_back(Item{a, b, 1});
                    back[a * n2 + b] = UINT64_MAX;
                }
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.len > max_len) { max_len = it.len; end_key = it.a * n2 + it.b; }
            const uint64_t b1 = forward ? g1.nx.begin(it.a) : gv1.prev_off[it.a], e1 = forward ? g1.nx.end(it.a) : gv1.prev_off[it.a + 1];
            const uint64_t b2 = forward ? g2.nx.begin(it.b) : gv2.prev_off[it.b], e2 = forward ? g2.nx.end(it.b) : gv2.prev_off[it.b + 1];
            const uint32_t* i1 = forward ? g1.nx.i : gv1.prev_idx;
            const uint32_t* i2 = forward ? g2.nx.i : gv2.prev_idx;
            for (uint64_t x = b1; x < e1; ++x)
                for (uint64_t y = b2; y < e2; ++y) {
                    const uint64_t na = i1[x], nb = i2[y];
                    if (gv1.label[na] == gv2.label[nb] && !back.count(na * n2 + nb)) {
                        back[na * n2 + nb] = it.a * n2 + it.b;
                        stack.push_back(Item{na, nb, it.len + 1});
                    }
                }
        }
        HostAlignment& aln = forward ? aln_fwd : aln_rev;
        while (end_key != UINT64_MAX) {
            aln.emplace_back(end_key / n2, end_key % n2);
            end_key = back.at(end_key);
        }
        if (forward) std::reverse(aln.begin(), aln.end());
    }
    size_t left_trim = 0, right_trim = 0;
    std::vector<uint64_t> path1, path2;
    bool found = false;
    if (aln_fwd.empty() || aln_rev.empty() ||
        (aln_fwd.back().first != aln_rev.front().first && aln_fwd.back().second != aln_rev.front().second)) {
        const std::vector<uint64_t> start1 = aln_fwd.empty() ? sources1 : std::vector<uint64_t>(1, aln_fwd.back().first);
        const std::vector<uint64_t> end1 = aln_rev.empty() ? sinks1 : std::vector<uint64_t>(1, aln_rev.front().first);
        if (!start1.empty() && !end1.empty()) path1 = g1.shortest(start1, end1);
        if (!path1.empty()) {
            const std::vector<uint64_t> start2 = aln_fwd.empty() ? sources2 : std::vector<uint64_t>(1, aln_fwd.back().second);
            const std::vector<uint64_t> end2 = aln_rev.empty() ? sinks2 : std::vector<uint64_t>(1, aln_rev.front().second);
            if (!start2.empty() && !end2.empty()) path2 = g2.shortest(start2, end2);
            if (!path2.empty()) {
                found = true;
                if (!aln_fwd.empty()) { path1.erase(path1.begin()); path2.erase(path2.begin()); }
                if (!aln_rev.empty()) { path1.pop_back(); path2.pop_back(); }
            }
        }
    }
    if (!found) {
        // the reference answers these questions by direct search for the first 8 and through a distance oracle afterwards
        // (alignment.hpp:1477-1497); both decide plain reachability
        auto reachable_after_trim = [&](size_t trim_left, size_t trim_right) {
            bool allow_equal = false;
            HostAlignment left, right;
            if (trim_left == aln_fwd.size()) {
                for (uint64_t a : sources1) for (uint64_t b : sources2) left.emplace_back(a, b);
                allow_equal = true;
            } else left.push_back(aln_fwd[aln_fwd.size() - 1 - trim_left]);
            if (trim_right == aln_rev.size()) {
                for (uint64_t a : sinks1) for (uint64_t b : sinks2) right.emplace_back(a, b);
                allow_equal = true;
            } else right.push_back(aln_rev[trim_right]);
            for (const auto& l : left)
                for (const auto& r : right) {
                    if (!allow_equal && (l.first == r.first || l.second == r.second)) continue;
                    if (g1.reaches(l.first, r.first) && g2.reaches(l.second, r.second)) return true;
                }
            return false;
        };
        int64_t lo = 1, hi = (int64_t)(aln_fwd.size() + aln_rev.size());
        while (lo <= hi) {
            const int64_t total = (lo + hi) / 2;
            bool success = false;
            const size_t l_min = (size_t)std::max<int64_t>(0, total - (int64_t)aln_rev.size());
            const size_t l_max = std::min<size_t>((size_t)total, aln_fwd.size());
            for (size_t l = l_min; l <= l_max; ++l)
                if (reachable_after_trim(l, (size_t)total - l)) {
                    left_trim = l;
                    right_trim = (size_t)total - l;
                    success = true;
                    break;
                }
            if (success) hi = total - 1;
            else lo = total + 1;
        }
        std::vector<uint64_t> from1, from2, to1, to2;
        if (left_trim == aln_fwd.size()) { from1 = sources1; from2 = sources2; }
        else { from1.push_back(aln_fwd[aln_fwd.size() - left_trim - 1].first); from2.push_back(aln_fwd[aln_fwd.size() - left_trim - 1].second); }
        if (right_trim == aln_rev.size()) { to1 = sinks1; to2 = sinks2; }
        else { to1.push_back(aln_rev[right_trim].first); to2.push_back(aln_rev[right_trim].second); }
        path1 = g1.shortest(from1, to1);
        path2 = g2.shortest(from2, to2);
        if (left_trim != aln_fwd.size()) { if (!path1.empty()) path1.erase(path1.begin()); if (!path2.empty()) path2.erase(path2.begin()); }
        if (right_trim != aln_rev.size()) { if (!path1.empty()) path1.pop_back(); if (!path2.empty()) path2.pop_back(); }
    }
    for (size_t i = 0; i + left_trim < aln_fwd.size(); ++i) out.push_back(aln_fwd[i]);
    for (uint64_t v : path1) out.emplace_back(v, CL_GAP);
    for (uint64_t v : path2) out.emplace_back(CL_GAP, v);
    for (size_t i = right_trim; i < aln_rev.size(); ++i) out.push_back(aln_rev[i]);
    return CL_OK;
}

// ---- the wavefront heuristics of do_alignment (stitcher.hpp:304-339), wfa_host.hpp
bool wfa_side(const GraphView& g, clwfa::Side& out) {
    if (!g.next_off) return false;   // the wavefronts expand next() lists in the reference's order
    out.n = g.n;
    out.label = g.label;
    out.next.assign(g.n, {});
    out.prev.assign(g.n, {});
    for (uint64_t v = 0; v < g.n; ++v) {
        out.next[v].assign(g.next_idx + g.next_off[v], g.next_idx + g.next_off[v + 1]);
        out.prev[v].assign(g.prev_idx + g.prev_off[v], g.prev_idx + g.prev_off[v + 1]);
    }
    out.sources.assign(g.src, g.src + g.n_src);
    out.sinks.assign(g.snk, g.snk + g.n_snk);
    NextLists nx;
    nx.build(g);
    std::vector<uint32_t> order, st, indeg;
    if (!topological_order(g, nx, order, st, indeg)) return false;
    out.order.assign(order.begin(), order.end());
    return true;
}

// every route of do_alignment that the reference runs on the host: pure deletion, greedy, deletion-WFA, pruned WFA
int host_route_alignment(int route, const GraphView& g1, const GraphView& g2, int npw, const cl_stitch_params& sp, HostAlignment& out) {
    out.clear();
    if (route == CL_ROUTE_PURE_DELETION_1 || route == CL_ROUTE_PURE_DELETION_2) {
        std::vector<uint32_t> path;
        const int rc = pure_deletion(route == CL_ROUTE_PURE_DELETION_1 ? g1 : g2, path);
        if (rc) return rc;
        for (uint32_t v : path) out.push_back(route == CL_ROUTE_PURE_DELETION_1 ? std::make_pair((uint64_t)v, (uint64_t)CL_GAP) : std::make_pair((uint64_t)CL_GAP, (uint64_t)v));
        return CL_OK;
    }
    if (route == CL_ROUTE_GREEDY_PARTIAL) {
        if (!g1.next_off || !g2.next_off) return CL_ERR_INVALID_ARGUMENT;
        return greedy_partial_alignment(g1, g2, out);
    }
    if (route != CL_ROUTE_DELETION_WFA_1 && route != CL_ROUTE_DELETION_WFA_2 && route != CL_ROUTE_PWFA) return CL_ERR_UNSUPPORTED_ROUTE;
    clwfa::Side s1, s2;
    if (!wfa_side(g1, s1) || !wfa_side(g2, s2)) return (g1.next_off && g2.next_off) ? CL_ERR_CYCLIC_GRAPH : CL_ERR_INVALID_ARGUMENT;
    const cl_align_params& ap = sp.alignment_params;
    const clwfa::WfaParams wp = clwfa::to_wfa_params(ap.match, ap.mismatch, ap.gap_open, ap.gap_extend, npw);   // the truncated parameters of subalign
    clwfa::Alignment aln;
    if (route == CL_ROUTE_PWFA) aln = clwfa::pwfa_po_poa(s1, s2, wp, (int64_t)(2 * sp.wfa_pruning_dist));
    else if (route == CL_ROUTE_DELETION_WFA_1) aln = clwfa::deletion_wfa_po_poa(s1, s2, wp);
    else {
        aln = clwfa::deletion_wfa_po_poa(s2, s1, wp);
        for (auto& pr : aln) std::swap(pr.first, pr.second);   // swap_graphs, src/alignment.cpp:41-45
    }
    out.assign(aln.begin(), aln.end());
    return CL_OK;
}

int64_t pure_deletion_score(size_t path_len, int npw, const cl_align_params& p) {
    if (path_len == 0) return 0;
    // alignment.hpp:1202-1205 evaluates -open - extend in uint32_t before widening; reproduced literally
    int64_t s = std::numeric_limits<int64_t>::max();
    for (int k = 0; k < npw; ++k) s = std::min<int64_t>(s, (int64_t)(uint32_t)(0u - p.gap_open[k] - p.gap_extend[k]));
    return s;
}

struct LaunchGroup {
    int kind = 0;
    int npw = 0;
    int block = 0;       // general kernel: workgroup size
    int rows = 0, waves = 0;  // linear kernel: rows per lane, waves per workgroup
    bool swap = false;        // linear kernel: graph 2 laid across the lanes
    uint32_t first = 0;  // into plist
    uint32_t count = 0;
    uint64_t cells = 0, bytes = 0;
    uint32_t ring_bytes = 0;  // general kernel: dynamic LDS of the ring variant (0 = planes read from HBM)
    uint32_t prog_first = 0, prog_count = 0;   // strip kernel: first / count index into the strip list (first, count) and the progress words it zeroes before the launch
    uint64_t est_cost = 0;    // longest sweep x the kernel's rough time per step: orders the groups and deals them over the streams
    float host_ms = 0.f;      // the profiled pass: the launch alone on the device, behind another launch (host's clock: two launches - one launch)
    float host_idle_ms = 0.f; // ... and finding the device idle (one launch + wait)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // cl_stitch_plan_execute_evented: HIP events round this launch on ITS stream in a concurrent pass
    bool evented = false;
};

}  // namespace

struct cl_stitch_plan {
    uint64_t n_problems = 0;
    cl_align_params aparams{};
    // per input problem
    std::vector<uint8_t> route, num_pw;
    std::vector<int32_t> po_index;                 // index among PO-POA problems or -1
    std::vector<std::vector<uint32_t>> pd_path;    // pure-deletion paths (local ids), indexed by input problem (sparse)
    std::vector<uint32_t> pd_problem;              // list of problems with a pd path
    std::vector<HostAlignment> host_aln;           // alignments made on the host (greedy route), indexed by input problem (sparse)
    std::vector<uint32_t> host_problem;
    // per PO-POA problem
    std::vector<ClProbDesc> desc;
    std::vector<uint8_t> lin_rows, lin_waves, lin_swap;  // chain-kernel geometry per PO-POA problem (0 = general kernel)
    std::vector<uint64_t> po_problem;              // input problem index
    std::vector<uint32_t> order[2];                // rank -> local node id, concatenated (node_base)
    // translation back to caller ids
    std::vector<uint64_t> back[2];                 // copy of back_translation (or empty)
    std::vector<uint64_t> node_off[2];
    bool has_back[2] = {false, false};
    // device
    DevBuf<ClProbDesc> d_desc;
    DevBuf<uint32_t> d_aux;
    DevBuf<uint8_t> d_lab[2];
    DevBuf<uint32_t> d_poff[2], d_pidx[2], d_snk[2];
    DevBuf<int32_t> d_planes;
    DevBuf<uint2> d_out_pairs;
    DevBuf<uint32_t> d_out_len, d_out_status, d_plist;
    DevBuf<int32_t> d_out_score;
    // strips of rows (popoa_strip_kernel)
    std::vector<ClStripDesc> strips;
    std::vector<uint32_t> strip_list;              // the strip launch groups' lists (into strips)
    DevBuf<ClStripDesc> d_strips;
    DevBuf<uint4> d_strip_recs;
    DevBuf<uint32_t> d_strip_list, d_progress;
    DevBuf<unsigned long long> d_handoff;
    ClStripDevice sdev{};
    DevBuf<uint32_t> d_lane_sync;                  // wide near-chain pairs (popoa_lane_kernel): progress / done words per group, zeroed in front of every launch
    DevBuf<unsigned long long> d_ticks;            // [2 per launch group] the launches' own clocks (ClDeviceBatch::ticks), zeroed in front of every pass
    std::vector<LaunchGroup> groups;
    ClDeviceBatch dev{};
    ClScoreParams sparams{};
    cl_plan_stats stats{};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_tried = false;
    std::vector<uint32_t> plist_host;   // the launch groups' subproblem lists (cl_stitch_plan_launch_info)
    bool executed = false, profiled = false, calibrated = false;
    int recalibrations_left = -1, concurrent_passes = 0;
    uint32_t tick_pass = 0;      // number of the last concurrent pass (ClDeviceBatch::tick_pass), 1 .. 65 535
    bool event_pass = false;     // the pass being enqueued brackets every launch with HIP events (cl_stitch_plan_execute_evented)
    bool stop_pending = false;   // the last pass's end is not marked on the context's stream yet (lazy join)   // second stage of the launch scheduling: see cl_stitch_plan_execute
};

namespace {

void plan_free(cl_stitch_plan* pl) {
    if (!pl) return;
    // (one wait for the owning context's streams, then the sixteen blocks go back to its pool: cl_internal.hpp, cl_ctx_quiesce)
    cl_context* owner = pl->d_desc.owner;
    const bool q = owner && cl_context_live(owner);
    if (q) cl_ctx_quiesce(owner);
    pl->d_desc.release(q);
    for (int s = 0; s < 2; ++s) {
        pl->d_lab[s].release(q); pl->d_poff[s].release(q); pl->d_pidx[s].release(q); pl->d_snk[s].release(q);
    }
    pl->d_planes.release(q); pl->d_out_pairs.release(q); pl->d_out_len.release(q); pl->d_out_status.release(q);
    pl->d_plist.release(q); pl->d_out_score.release(q); pl->d_aux.release(q);
    pl->d_strips.release(q); pl->d_strip_recs.release(q); pl->d_strip_list.release(q); pl->d_progress.release(q); pl->d_handoff.release(q);
    pl->d_ticks.release(q); pl->d_lane_sync.release(q);
    for (auto& g : pl->groups) {
        if (g.ev0) (void)hipEventDestroy(g.ev0);
        if (g.ev1) (void)hipEventDestroy(g.ev1);
    }
    if (pl->graph_exec) (void)hipGraphExecDestroy(pl->graph_exec);
    if (pl->graph) (void)hipGraphDestroy(pl->graph);
    if (pl->ev_start) (void)hipEventDestroy(pl->ev_start);
    if (pl->ev_stop) (void)hipEventDestroy(pl->ev_stop);
    delete pl;
}

}  // namespace

// ---- the process's host thread pool (cl_parallel_for, cl_internal.hpp) ---------------------------------------------------------------
namespace {
struct HostPool {
    struct Batch {
        const std::function<void(unsigned)>* task;
        unsigned n_tasks;
        std::atomic<unsigned> next{1};       // task 0 belongs to the caller
        std::atomic<unsigned> done{0};
        std::mutex m;
        std::condition_variable cv;
    };
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Batch>> open;   // batches that still have unclaimed tasks
    std::vector<std::thread> threads;
    unsigned width = 1;
    bool stop = false;
    static thread_local bool is_worker;

    HostPool() {
        const char* e = getenv("CL_HOST_THREADS");
        const int v = e ? atoi(e) : 0;
        const unsigned hw = std::thread::hardware_concurrency();
        width = v > 0 ? (unsigned)v : std::min(hw ? hw : 1u, 32u);
        // more threads than one loop uses: several contexts (MSA workers) run loops at the same time
        const unsigned n_threads = width <= 1 ? 0 : std::min(hw ? hw : 1u, 4 * width) - 1;
        for (unsigned i = 0; i < n_threads; ++i) threads.emplace_back([this] { work(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lock(m); stop = true; }
        cv.notify_all();
        for (auto& t : threads) t.join();
    }
    static void run_one(Batch& b, unsigned t) {
        (*b.task)(t);
        if (b.done.fetch_add(1) + 1 == b.n_tasks) { std::lock_guard<std::mutex> lock(b.m); b.cv.notify_all(); }
    }
    void work() {
        is_worker = true;
        while (true) {
            std::shared_ptr<Batch> b;
            unsigned t = 0;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return stop || !open.empty(); });
                if (stop) return;
                b = open.front();
                t = b->next.fetch_add(1);
                if (t + 1 >= b->n_tasks) open.pop_front();   // this was the last unclaimed task
                if (t >= b->n_tasks) continue;
            }
            run_one(*b, t);
        }
    }
    void run(unsigned n_tasks, const std::function<void(unsigned)>& task) {
        if (n_tasks <= 1 || threads.empty() || is_worker) {   // (a loop inside a pool task runs in place)
            for (unsigned t = 0; t < n_tasks; ++t) task(t);
            return;
        }
        auto b = std::make_shared<Batch>();
        b->task = &task;
        b->n_tasks = n_tasks;
        { std::lock_guard<std::mutex> lock(m); open.push_back(b); }
        cv.notify_all();
        run_one(*b, 0);
        // help with this batch's remaining tasks instead of sleeping, then wait for the ones other threads took
        while (true) {
            unsigned t;
            {
                std::lock_guard<std::mutex> lock(m);
                t = b->next.load();
                if (t >= b->n_tasks) break;
                t = b->next.fetch_add(1);
                if (t + 1 >= b->n_tasks) { auto it = std::find(open.begin(), open.end(), b); if (it != open.end()) open.erase(it); }
                if (t >= b->n_tasks) break;
            }
            run_one(*b, t);
        }
        std::unique_lock<std::mutex> lock(b->m);
        b->cv.wait(lock, [&] { return b->done.load() == b->n_tasks; });
    }
};
thread_local bool HostPool::is_worker = false;
HostPool& host_pool() { static HostPool pool; return pool; }
}  // namespace
void cl_pool_run(unsigned n_tasks, const std::function<void(unsigned)>& task) { host_pool().run(n_tasks, task); }
unsigned cl_pool_width() { return host_pool().width; }

namespace {
std::mutex g_live_mutex;
std::vector<const cl_context*> g_live_contexts;
}
bool cl_context_live(const cl_context* ctx) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    return std::find(g_live_contexts.begin(), g_live_contexts.end(), ctx) != g_live_contexts.end();
}

extern "C" {

int cl_abi_version(void) { return CL_ABI_VERSION; }

void cl_fallback_counters(cl_fallback_stats* out, int reset) {
    if (out) {
        out->strip_fallbacks = cl_fallbacks.strip_fallbacks.load();
        out->walk_stalls = cl_fallbacks.walk_stalls.load();
        out->chain_dps = cl_fallbacks.chain_dps.load();
        out->stitch_plans = cl_fallbacks.stitch_plans.load();
        out->strip_pairs = cl_fallbacks.strip_pairs.load();
        out->bond_trims_past_the_end = cl_fallbacks.bond_trims_past_the_end.load();
    }
    if (reset) { cl_fallbacks.strip_fallbacks = 0; cl_fallbacks.walk_stalls = 0; cl_fallbacks.chain_dps = 0; cl_fallbacks.stitch_plans = 0; cl_fallbacks.strip_pairs = 0; cl_fallbacks.bond_trims_past_the_end = 0; }
}

void cl_stitch_params_default(cl_stitch_params* p) {
    // src/parameters.cpp:74-85 (the CLI's values; Stitcher's class defaults differ)
    p->alignment_params.match = 20;
    p->alignment_params.mismatch = 80;
    const uint32_t go[3] = {60, 800, 2500}, ge[3] = {30, 5, 1};
    for (int i = 0; i < 3; ++i) { p->alignment_params.gap_open[i] = go[i]; p->alignment_params.gap_extend[i] = ge[i]; }
    p->max_trivial_size = 30000;
    p->min_wfa_size = 40000000;
    p->max_wfa_size = 75000000;
    p->max_wfa_ratio = 1.05;
    p->wfa_pruning_dist = 25;
    p->deletion_alignment_ratio = 8;
    p->deletion_alignment_short_max_size = 1500;
    p->deletion_alignment_long_min_size = 2000;
}

int cl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* cl_last_error(const cl_context* ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

const char* cl_device_name(const cl_context* ctx) { return ctx ? ctx->name.c_str() : ""; }

cl_context* cl_context_create(int device_ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(nullptr, "no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        set_error(nullptr, "device ordinal %d out of range (0..%d)", device_ordinal, n - 1);
        return nullptr;
    }
    cl_context* ctx = new (std::nothrow) cl_context();
    if (!ctx) return nullptr;
    ctx->device = device_ordinal;
    int callers_device = -1;   // the caller's current device is put back: creating worker contexts on a list of devices must not move the calling thread
    (void)hipGetDevice(&callers_device);
    hipDeviceProp_t prop;
    bool ok = hipSetDevice(device_ordinal) == hipSuccess && hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    // streams of their own: CL_CTX_STREAMS (default 6: what one chaining DP keeps busy — four far launches in flight, the sealing stream and
    // a spare).  Every stream takes a share of the process's hardware queues (GPU_MAX_HW_QUEUES); streams that share a queue run their
    // launches one after the other, whichever context they belong to, so contexts that work side by side should not hold idle streams
    // (read at every creation: a caller that wants a context with more streams for one purpose — bench.py's resident stitch plan: eight — sets the variable round that call)
    const int n_own = [] { const char* e = getenv("CL_CTX_STREAMS"); int v = e ? atoi(e) : 0; return v >= 1 && v <= kNumAuxStreams ? v : 6; }();
    ctx->n_aux = n_own;
    for (int i = 0; ok && i < kNumAuxStreams; ++i) {
        // CL_CTX_PRIO_STREAMS=k: the first k auxiliary streams get the device's highest priority — a stitch plan deals its longest launches to them first
        // (measurement knob, read at every creation like CL_CTX_STREAMS)
        const int n_prio = [] { const char* e = getenv("CL_CTX_PRIO_STREAMS"); int v = e ? atoi(e) : 0; return v >= 0 && v <= kNumAuxStreams ? v : 0; }();
        int prio_lo = 0, prio_hi = 0;
        if (n_prio) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (i < n_own) ok = (i < n_prio ? hipStreamCreateWithPriority(&ctx->aux[i], hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking)) == hipSuccess;
        else ctx->aux[i] = ctx->aux[i % n_own];
        ok = ok && hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        set_error(nullptr, "HIP context setup failed on device %d: %s", device_ordinal, hipGetErrorString(hipGetLastError()));
        cl_context_destroy(ctx);
        return nullptr;
    }
    ctx->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    if (callers_device >= 0 && callers_device != device_ordinal) (void)hipSetDevice(callers_device);
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live_contexts.push_back(ctx);
    }
    return ctx;
}

void cl_context_destroy(cl_context* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live_contexts.erase(std::remove(g_live_contexts.begin(), g_live_contexts.end(), ctx), g_live_contexts.end());
    }
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < kNumAuxStreams; ++i) {
        if (ctx->aux[i] && i < ctx->n_aux) (void)hipStreamDestroy(ctx->aux[i]);
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    cl_pinned_release(ctx);
    cl_peers_release(ctx);
    for (auto& ring : ctx->ev_ring) for (hipEvent_t e : ring) if (e) (void)hipEventDestroy(e);
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        cl_pool_trim(ctx);
    }
    delete ctx;
}

void cl_stitch_result_free(cl_stitch_result* r) {
    if (!r) return;
    free(r->aln_off); free(r->pairs); free(r->score); free(r->route); free(r->num_pw);
    memset(r, 0, sizeof(*r));
}

int cl_stitch_result_alloc(cl_stitch_result* out, uint64_t n_problems, uint64_t n_pairs) {
    if (!out) return CL_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    out->n_problems = n_problems;
    out->aln_off = (uint64_t*)calloc(n_problems + 1, sizeof(uint64_t));
    out->pairs = (uint64_t*)malloc((n_pairs ? n_pairs : 1) * 2 * sizeof(uint64_t));
    out->score = (int64_t*)calloc(n_problems ? n_problems : 1, sizeof(int64_t));
    out->route = (uint8_t*)calloc(n_problems ? n_problems : 1, 1);
    out->num_pw = (uint8_t*)calloc(n_problems ? n_problems : 1, 1);
    if (!out->aln_off || !out->pairs || !out->score || !out->route || !out->num_pw) { cl_stitch_result_free(out); return CL_ERR_OUT_OF_MEMORY; }
    return CL_OK;
}

int cl_context_set_stitch_hook(cl_context* ctx, cl_stitch_hook_fn fn, void* user, uint64_t min_cells) {
    if (!ctx) return CL_ERR_INVALID_ARGUMENT;
    ctx->stitch_hook = fn;
    ctx->stitch_hook_user = fn ? user : nullptr;
    ctx->stitch_hook_min_cells = min_cells;
    return CL_OK;
}

int cl_stitch_plan_create(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* params,
                          const uint8_t* force_num_pw, cl_stitch_plan** plan_out) {
    if (!ctx || !batch || !params || !plan_out) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    *plan_out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n = batch->n_problems;
    int rc;
    if ((rc = validate_side(ctx, batch->side[0], n, 1)) || (rc = validate_side(ctx, batch->side[1], n, 2))) return rc;
    const cl_align_params& ap = params->alignment_params;
    if (ap.match > (1u << 20) || ap.mismatch > (1u << 20)) { set_error(ctx, "match/mismatch too large"); return CL_ERR_INVALID_ARGUMENT; }
    for (int k = 0; k < 3; ++k)
        if (ap.gap_open[k] > (1u << 24) || ap.gap_extend[k] > (1u << 24)) { set_error(ctx, "gap penalties too large"); return CL_ERR_INVALID_ARGUMENT; }

    static const bool timing = getenv("CL_STITCH_TIMING") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[cl_stitch]     %-20s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count());
        tp = std::chrono::steady_clock::now();
    };
    cl_stitch_plan* pl = new (std::nothrow) cl_stitch_plan();
    if (!pl) return CL_ERR_OUT_OF_MEMORY;
    pl->n_problems = n;
    pl->aparams = ap;
    pl->route.assign(n, 0);
    pl->num_pw.assign(n, 0);
    pl->po_index.assign(n, -1);
    pl->pd_path.resize(n);
    pl->host_aln.resize(n);
    for (int s = 0; s < 2; ++s) {
        pl->node_off[s].assign(batch->side[s].node_off, batch->side[s].node_off + n + 1);
        if (batch->side[s].back_translation) {
            pl->has_back[s] = true;
            pl->back[s].assign(batch->side[s].back_translation, batch->side[s].back_translation + batch->side[s].node_off[n]);
        }
    }

    // host-side packing into rank space.  The subproblems are independent: chunks of them are packed side by side into parts of their own
    // (offsets relative to the part) and the parts are appended in order, which gives exactly the arrays of a serial pass
    struct PackPart {
        std::vector<uint8_t> lab[2];
        std::vector<uint32_t> poff[2], pidx[2], snk[2], order[2];
        std::vector<uint32_t> ring_need;   // per PO-POA problem: dynamic LDS of the ring / systolic variant (0 = not taken)
        std::vector<uint32_t> sys_aux;     // saved-column lists of the systolic kernel's problems, concatenated
        std::vector<ClStripDesc> strips;   // strip kernel: prob = index into desc of this part; rec_base / hand_* / prog relative to this part
        std::vector<uint4> strip_recs;
        std::vector<uint32_t> strip_lds;   // per strip: dynamic LDS
        uint64_t hand_words = 0;
        std::vector<ClProbDesc> desc;
        std::vector<uint64_t> po_problem;
        std::vector<uint32_t> pd_problem, host_problem;
        std::vector<uint8_t> lin_rows, lin_waves, lin_swap;
        uint64_t plane_cursor = 0, out_cursor = 0, dp_cells = 0, dp_bytes = 0, max_cells = 0, n_linear = 0;
        int rc = CL_OK;
        std::string err;
        PackPart() { poff[0].push_back(0); poff[1].push_back(0); }
        void fail(int code, const char* fmt, ...) {
            if (rc) return;
            rc = code;
            char buf[512];
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, sizeof(buf), fmt, ap);
            va_end(ap);
            err = buf;
        }
    };
    struct Scratch { std::vector<uint32_t> order, st, indeg, rank; };
    const int rank_order = rank_order_now();
    const uint32_t lane_rw = lane_round_waves(n);
    // waves of popoa_linear_kernel's workgroup for chain pairs of 65-256 rows / above (rounds of W strips): CL_LINEAR_WAVES_MID=3|4 [4], CL_LINEAR_WAVES_BIG=3|4|8|16
    // [8 up to 512 rows, 16 above].  Measured (profiles/r05_linear_waves_ab.txt): the timed step of 10 x 1 Mbp 2.29-2.31 ms with sixteen waves for every pair above 256
    // rows, 2.16-2.22 with eight, 2.15-2.24 with four, 2.16-2.19 with three; alone on the device 512^2 x 150 runs the same with 16 / 8 / 4 (105 G cells/s), 2 048^2 x 10
    // 15.9 / 14.3 / 10.4 G cells/s — so eight where eight strips are all there are, sixteen above.  Three waves for 65-256 rows: slower in the step (2.25-2.37)
    auto waves_env = [](const char* name, int dflt) { const char* e = getenv(name); const int v = e ? atoi(e) : 0; return (uint8_t)(v == 3 || v == 4 || v == 8 || v == 16 ? v : dflt); };
    const uint8_t lin_mid = waves_env("CL_LINEAR_WAVES_MID", 4), lin_big = waves_env("CL_LINEAR_WAVES_BIG", 0);
    auto pack_one = [&](uint64_t k, PackPart& P, Scratch& S) {
        std::vector<uint32_t>& order = S.order; std::vector<uint32_t>& st = S.st; std::vector<uint32_t>& indeg = S.indeg; std::vector<uint32_t>& rank = S.rank;
        GraphView g[2] = {view(batch->side[0], k), view(batch->side[1], k)};
        int npw = force_num_pw ? force_num_pw[k] : choose_num_pw(g[0].n, g[1].n, ap);
        if (npw < 1 || npw > 3) { P.fail(npw < 0 ? npw : CL_ERR_INVALID_ARGUMENT, npw == CL_ERR_BAD_GAP_PARAMS ? "Affine gap parameters must be increasing in gap open penalty and decreasing in gap extend penalty"
                                                        : "num_pw must be 1, 2 or 3 (problem %llu)", (unsigned long long)k); return; }
        bool only_del = batch->only_deletion_alns && batch->only_deletion_alns[k];
        int route;
        if (force_num_pw) route = g[1].n == 0 ? CL_ROUTE_PURE_DELETION_1 : g[0].n == 0 ? CL_ROUTE_PURE_DELETION_2 : CL_ROUTE_PO_POA;
        else route = route_problem(g[0], g[1], only_del, *params);
        if (route < 0) { P.fail(route, "problem %llu: subgraph is not acyclic", (unsigned long long)k); return; }
        pl->route[k] = (uint8_t)route;
        pl->num_pw[k] = (uint8_t)npw;
        if (route == CL_ROUTE_PURE_DELETION_1 || route == CL_ROUTE_PURE_DELETION_2) {
            const int rc = pure_deletion(route == CL_ROUTE_PURE_DELETION_1 ? g[0] : g[1], pl->pd_path[k]);
            if (rc) { P.fail(rc, "problem %llu: pure deletion failed", (unsigned long long)k); return; }
            P.pd_problem.push_back((uint32_t)k);
            return;
        }
        if (route != CL_ROUTE_PO_POA) {   // greedy / deletion-WFA / pruned WFA: host algorithms (in the reference too)
            const int rc = host_route_alignment(route, g[0], g[1], npw, *params, pl->host_aln[k]);
            if (rc) { P.fail(rc, rc == CL_ERR_INVALID_ARGUMENT ? "problem %llu: route %d needs the next lists of the subgraphs" : "problem %llu: host route %d failed",
                          (unsigned long long)k, route); return; }
            P.host_problem.push_back((uint32_t)k);
            return;
        }
        uint64_t cells = (g[0].n + 1) * (g[1].n + 1);
        if (cells >= (1ull << 30) || g[0].n_src == 0 || g[1].n_src == 0 || g[0].n_snk == 0 || g[1].n_snk == 0) { P.fail(cells >= (1ull << 30) ? CL_ERR_INVALID_ARGUMENT : CL_ERR_UNREACHABLE_SINK, "problem %llu: matrix too large for the device path or no sources/sinks", (unsigned long long)k); return; }
        ClProbDesc d{};
        d.n1 = (uint32_t)g[0].n;
        d.n2 = (uint32_t)g[1].n;
        d.npw = (uint8_t)npw;
        bool linear = true;
        uint64_t span[2] = {0, 0};   // how many rows (columns) back the cells of a row (column) read: predecessors, boundary for a source
        for (int s = 0; s < 2; ++s) {
            NextLists nx;
            nx.build(g[s]);
            if (!topological_order(g[s], nx, order, st, indeg)) { P.fail(CL_ERR_CYCLIC_GRAPH, "problem %llu: graph %d is not acyclic", (unsigned long long)k, s + 1); return; }
            rank.resize(g[s].n);
            for (uint32_t r = 0; r < g[s].n; ++r) rank[order[r]] = r;
            choose_rank_order(g[s], rank_order, order, rank, st, indeg);
            if (P.lab[s].size() + g[s].n >= (1ull << 32) || P.pidx[s].size() + (g[s].prev_off[g[s].n] - g[s].prev_off[0]) >= (1ull << 32)) { P.fail(CL_ERR_INVALID_ARGUMENT, "batch too large for 32-bit device offsets"); return; }
            d.node_base[s] = (uint32_t)P.lab[s].size();
            size_t lab0 = P.lab[s].size();
            for (uint32_t r = 0; r < g[s].n; ++r) {
                uint32_t v = order[r];
                P.lab[s].push_back(g[s].label[v] & 0x7f);
                uint64_t deg = g[s].prev_off[v + 1] - g[s].prev_off[v];
                for (uint64_t e = g[s].prev_off[v]; e < g[s].prev_off[v + 1]; ++e) {
                    P.pidx[s].push_back(rank[g[s].prev_idx[e]] + 1);
                    span[s] = std::max<uint64_t>(span[s], r - rank[g[s].prev_idx[e]]);
                }
                P.poff[s].push_back((uint32_t)P.pidx[s].size());
                if (r == 0 ? deg != 0 : (deg != 1 || rank[g[s].prev_idx[g[s].prev_off[v]]] != r - 1)) linear = false;
                if (g[s].label[v] & 0x80) { P.fail(CL_ERR_INVALID_ARGUMENT, "labels must be < 128"); return; }
            }
            for (uint64_t i = 0; i < g[s].n_src; ++i) {
                P.lab[s][lab0 + rank[g[s].src[i]]] |= 0x80;
                span[s] = std::max<uint64_t>(span[s], (uint64_t)rank[g[s].src[i]] + 1);   // a source reads the boundary index 0
            }
            if (g[s].n_src != 1 || rank[g[s].src[0]] != 0) linear = false;
            if (g[s].n_snk != 1 || rank[g[s].snk[0]] != g[s].n - 1) linear = false;
            d.snk_base[s] = (uint32_t)P.snk[s].size();
            d.snk_cnt[s] = (uint32_t)g[s].n_snk;
            for (uint64_t i = 0; i < g[s].n_snk; ++i) P.snk[s].push_back(rank[g[s].snk[i]] + 1);
            P.order[s].insert(P.order[s].end(), order.begin(), order.end());
        }
        // (a chain pair of 4 096 rows and more would take the chain kernel's ONE workgroup through four and more passes: the strip kernel puts it on
        // twenty compute units instead — 6 300 x 6 300: 22.4 -> ≈ 12 ms; such a pair always meets the strip kernel's conditions)
        // round 6: chain pairs of more than 1 024 rows span SEVERAL workgroups (popoa_linear_span_kernel, popoa_linear.hip: groups of four strips on different compute units, one
        // round): 2 048 x 2 048 and 6 300 x 6 300 leave the sixteen-wave workgroup's rounds / the DAG strip kernel.  CL_LINEAR_SPAN=0: the routing of rounds 4-5 (A/B)
        const bool span_env = [] { const char* e = getenv("CL_LINEAR_SPAN"); return !e || e[0] != '0'; }();   // (read per plan: the tests of the older routes switch it)
        const bool span_linear = span_env && linear && !g_force_general && std::min(d.n1, d.n2) > 1024 && std::min(d.n1, d.n2) <= 40000;
        const bool big_linear = !span_linear && linear && !g_no_strip && !g_force_general && std::min(d.n1, d.n2) >= 4096 && std::min(d.n1, d.n2) <= 40000;
        d.kind = (linear && !g_force_general && !big_linear) ? CL_KIND_LINEAR : CL_KIND_GENERAL;
        d.plane_base = P.plane_cursor;
        uint8_t lr = 0, lw = 0, ls = 0;
        if (d.kind == CL_KIND_LINEAR) {
            // the shorter graph goes across the lanes; strips of 64 rows are pipelined over the waves
            ls = d.n2 < d.n1;
            const uint32_t nshort = std::min(d.n1, d.n2), nlong = std::max(d.n1, d.n2);
            // (lw = 2: FOUR pairs per wave, 16 lanes each — popoa_linear_quad_kernel; CL_NO_LINEAR_QUADS=1: one pair per wave as in rounds 1-4)
            static const bool no_quads = [] { const char* e = getenv("CL_NO_LINEAR_QUADS"); return e && *e == '1'; }();
            const bool no_duos = duos_forced() == 0;   // (whether the plan takes the kernel at all is decided when every pair is known: below)
            if (span_linear) { lr = 1; lw = 40; }   // (lw = 40: the pair's strips over several workgroups)
            else if (nshort <= 16 && !no_quads) { lr = 1; lw = 2; }
            else if (nshort <= 32 && !no_duos) { lr = 1; lw = 5; }   // (lw = 5: TWO pairs per wave, 32 lanes each — popoa_linear_duo_kernel; see duos_now)
            else if (nshort <= 64) { lr = 1; lw = 1; }
            else if (nshort <= 128 && nlong < 300) { lr = 2; lw = 1; }
            else if (nshort <= 256) { lr = 1; lw = lin_mid; }
            else { lr = 1; lw = lin_big ? lin_big : (nshort <= 512 || n >= 2048) ? 8 : 16; }   // (up to eight strips: eight waves — with sixteen, half of them only stand at the barriers; above, sixteen only in plans that have the device to themselves: the one such pair of the 10 x 1 Mbp step as a launch of its own costs the step 3 %, 1.61 against 1.56 ms, three runs each)
            d.pad = (uint16_t)(ls | (lr << 1));  // read by linear_dispatch
            P.plane_cursor += (cl_linear_workspace_bytes(nshort, nlong, npw, lr) + 15) / 16 * 4;
            P.ring_need.push_back(0);
        } else {
            P.plane_cursor += (cells * (uint64_t)(1 + 2 * npw) + 3) / 4 * 4;
            // LDS ring of the most recent anti-diagonals (popoa_kernels.hip): span1+span2+1 of them serve every read; when that
            // does not fit, as many as do (at least 8) — the rare reads that reach further back go to HBM
            const uint64_t width = std::min(d.n1, d.n2) + 1, per_diag = width * (uint64_t)(1 + 2 * npw) * 4;
            // the ring variant also stages the subproblem's topology in LDS: offsets, predecessor ranks, labels
            const uint64_t n_pred = (P.poff[0].back() - P.poff[0][d.node_base[0]]) + (P.poff[1].back() - P.poff[1][d.node_base[1]]);
            const uint64_t topo_bytes = ((uint64_t)d.n1 + d.n2) * 8 + n_pred * 4 + ((uint64_t)d.n1 + d.n2 + 2) * (1 + npw) * 4 + 16;   // node records, lists, boundaries
            // the systolic kernel (popoa_sys_kernel): the shorter graph's rows on the threads, a ring of H columns per row in LDS.  H must
            // exceed the row graph's predecessor span plus the column graph's NEAR predecessor distances; columns that are read from
            // further away (the fork in front of a long bubble; column 0 for a late source) are SAVED columns with LDS of their own
            const int sRow = d.n2 < d.n1 ? 1 : 0, sCol = 1 - sRow;
            const uint64_t n_rows = std::min(d.n1, d.n2) + 1, n_cols = std::max(d.n1, d.n2);
            // near-chain pairs in registers (popoa_lane_kernel, popoa_lane.h): every node one to four predecessors (a source's boundary index counted), the row
            // graph's within 4 ranks, the column graph's within 4 columns or in at most sixteen SAVED columns (two per column at most); rows = the shorter graph.
            // Two shapes of the cell: predecessors up to 2 rows / 3 columns back, or 4 / 4
            bool take_lane = false;
            uint32_t lane_dr = 0, lane_dc = 0, lane_slots = 0;
            uint64_t lane_lds = 0;
            std::vector<uint32_t> lane_words;
            // Which workgroup shapes (measured on one pair alone, profiles/r05_lane_probe.txt and DESIGN.md §4.3): one, two or three ACTIVE waves of a workgroup run a
            // step in 0.24 / 0.30 / 0.30 us, four in 0.55 us, eight in 0.93 us; strips dealt to several workgroups (WIDE: groups of three strips, progress words between
            // them) pay ~14 us per chunk of 32 steps for the hand-off across compute units (write-through stores against a 64-deep store queue, reads past the caches).
            // Inside the timed step of 10 x 1 Mbp (sixteen launches side by side, three runs each): register kernel for pairs up to 192 rows 2.49-2.58 ms, up to 512 or
            // 1 024 rows (further rounds of four waves) 2.30-2.35 ms, not at all 2.57 ms — so the default is 1 024 rows (CL_LANE_MAX_ROWS), in one workgroup.
            // CL_LANE_WIDE=1 deals pairs above 192 rows to several workgroups (up to 128 groups): parity-tested (tests/test_gpu_parity.py), slower than the strips
            // on 5 500 x 5 500 (16.6 against 12.7 ms), off by default
            const bool want_wide = [] { const char* e = getenv("CL_LANE_WIDE"); return e && *e == '1'; }();
            const uint64_t lane_max_rows = [] { const char* e = getenv("CL_LANE_MAX_ROWS"); const long v = e ? atol(e) : 1024; return (uint64_t)(v < 1 ? 1 : v > 1024 ? 1024 : v); }();
            const bool no_wide = !want_wide;
            const bool lane_wide = want_wide && n_rows > 193;
            const uint64_t lane_groups = lane_wide ? ((n_rows - 1 + 63) / 64 + 2) / 3 : 1;
            // which pairs: the register kernel wins where a launch lasts as long as its longest sweep (alone on the device: 2 225 x 165 0.88 ms against 1.32 ms on the systolic
            // kernel, 2 130 x 35 0.83 against 1.17, 441 x 433 0.60 against 0.71); on the thousands of small pairs that fill a launch's workgroups it issues about as many
            // instructions per cell as the systolic kernel and its launches, dealt by (waves, long / short), interleave worse: pairs below CL_LANE_MIN_SWEEP rows + columns
            // (default below) stay where they were
            const uint64_t lane_min_sweep = [] { const char* e = getenv("CL_LANE_MIN_SWEEP"); return e ? (uint64_t)atoll(e) : (uint64_t)512; }();   // (read per plan: tests)
            if (!g_no_lane && !g_force_general && lane_groups <= 128 && (lane_wide || n_rows - 1 <= lane_max_rows) && !(lane_wide && (no_wide || g_no_strip)) && n_cols < (1u << 28) && n_rows - 1 + n_cols >= lane_min_sweep) {
                const uint32_t nR = (uint32_t)n_rows - 1, nCl = (uint32_t)n_cols;
                const uint32_t* rp = P.poff[sRow].data() + d.node_base[sRow];
                const uint8_t* rl = P.lab[sRow].data() + d.node_base[sRow];
                const uint32_t* cp = P.poff[sCol].data() + d.node_base[sCol];
                const uint8_t* cl = P.lab[sCol].data() + d.node_base[sCol];
                uint32_t max_rd = 0;
                bool ok = true;
                for (uint32_t i = 1; i <= nR && ok; ++i) {
                    if (rp[i] - rp[i - 1] + (rl[i - 1] >> 7) == 0) ok = false;
                    for (uint32_t e = rp[i - 1]; e < rp[i]; ++e) max_rd = std::max(max_rd, i - P.pidx[sRow][e]);
                }
                ok = ok && max_rd <= 4;
                for (int shape = max_rd <= 2 ? 0 : 1; shape < 2 && ok && !take_lane; ++shape) {
                    const uint32_t near = 3u;   // (the lane's history rings hold the last three columns; further back = a saved column)
                    std::vector<uint32_t> far;
                    uint32_t max_cd = 0;
                    bool fits = true;
                    for (uint32_t j = 1; j <= nCl && fits; ++j) {
                        uint32_t nf = 0;
                        if (cp[j] - cp[j - 1] + (cl[j - 1] >> 7) == 0) fits = false;
                        for (uint32_t e = cp[j - 1]; e < cp[j]; ++e) {
                            const uint32_t dist = j - P.pidx[sCol][e];
                            if (dist > near) { far.push_back(P.pidx[sCol][e]); ++nf; } else max_cd = std::max(max_cd, dist);
                        }
                        if (nf > 2) fits = false;
                    }
                    std::sort(far.begin(), far.end());
                    far.erase(std::unique(far.begin(), far.end()), far.end());
                    lane_dr = shape == 0 ? 2 : 4; lane_dc = near; lane_slots = (uint32_t)far.size();
                    // LDS of a workgroup: the hand-off window between neighbouring strips ([waves - 1][DR][1 + NumPW][128 columns]) + the saved columns
                    const uint64_t lane_w = n_rows - 1 <= 64 ? 1 : lane_rw;
                    lane_lds = (lane_w > 1 ? (lane_w - 1) * lane_dr * (1 + npw) * 128 * 4 : 0) + (uint64_t)lane_slots * (lane_dr + (lane_wide ? 192 : n_rows - 1) + 1) * (1 + npw) * 4;
                    if (!fits || far.size() > 16 || lane_lds > 150 * 1024) continue;
                    take_lane = true;
                    // shortest walk from a source, in nodes (the boundary cells' closed form)
                    lane_words.assign(1 + 2ull * nR + 2ull * nCl, 0u);   // [0]: a wide pair's first progress word (set when the launch groups are made)
                    uint32_t* rowrec = lane_words.data() + 1, *rowdist = rowrec + nR, *colrec = rowdist + nR, *coldist = colrec + nCl;
                    for (uint32_t i = 1; i <= nR; ++i) {
                        uint32_t mask = 0, best = (rl[i - 1] >> 7) ? 1u : UINT32_MAX;
                        for (uint32_t e = rp[i - 1]; e < rp[i]; ++e) {
                            const uint32_t pr = P.pidx[sRow][e];
                            mask |= 1u << (i - pr - 1);
                            best = std::min(best, rowdist[pr - 1] == UINT32_MAX ? UINT32_MAX : rowdist[pr - 1] + 1);
                        }
                        rowdist[i - 1] = best;
                        rowrec[i - 1] = mask | ((uint32_t)(rl[i - 1] >> 7) << 4) | ((uint32_t)(rl[i - 1] & 0x7Fu) << 8);
                        if (best == UINT32_MAX) take_lane = false;   // (not reachable from a source: never in an extracted subgraph)
                    }
                    for (uint32_t j = 1; j <= nCl; ++j) {
                        uint32_t mask = 0, nf = 0, slots = 0, best = (cl[j - 1] >> 7) ? 1u : UINT32_MAX;
                        for (uint32_t e = cp[j - 1]; e < cp[j]; ++e) {
                            const uint32_t q = P.pidx[sCol][e];
                            if (j - q <= near) mask |= 1u << (j - q - 1);
                            else { slots |= (uint32_t)(std::lower_bound(far.begin(), far.end(), q) - far.begin()) << (20 + 4 * nf); ++nf; }
                            best = std::min(best, coldist[q - 1] == UINT32_MAX ? UINT32_MAX : coldist[q - 1] + 1);
                        }
                        coldist[j - 1] = best;
                        const auto self = std::lower_bound(far.begin(), far.end(), j);
                        const uint32_t keep = self != far.end() && *self == j ? (1u << 15) | ((uint32_t)(self - far.begin()) << 16) : 0u;
                        colrec[j - 1] = mask | ((uint32_t)(cl[j - 1] >> 7) << 4) | (nf << 5) | ((uint32_t)(cl[j - 1] & 0x7Fu) << 8) | keep | slots;
                        if (best == UINT32_MAX) take_lane = false;
                    }
                    if (!take_lane) break;
                }
            }
            bool take_sys = !take_lane && !g_no_sys && n_rows <= 1024;
            uint32_t sys_log = 0;
            uint64_t sys_bytes = 0;
            uint64_t near_limit = 8;
            std::vector<uint32_t> far_cols;
            if (take_sys) {
                const uint32_t* cp = P.poff[sCol].data() + d.node_base[sCol];   // node j (1-based rank): predecessors P.pidx[cp[j - 1] .. cp[j])
                const uint8_t* cl = P.lab[sCol].data() + d.node_base[sCol];
                uint64_t near_max = 0, max_deg = 0;
                for (uint64_t j = 1; j <= n_cols; ++j) max_deg = std::max<uint64_t>(max_deg, cp[j] - cp[j - 1]);
                // the near limit decides the ring depth H (LDS per row) against the number of saved columns (LDS per column): the candidate
                // with the smallest footprint wins — LDS is what bounds how many subproblems a CU holds at once
                const uint64_t cw = npw == 1 ? 4 : 8;
                const uint64_t candidates[7] = {2, 4, 8, 32, 128, 512, 2047};   // the kernel's column records hold a near distance in 11 bits
                uint64_t best_bytes = UINT64_MAX;
                std::vector<uint32_t> cand_cols;
                for (uint64_t limit : candidates) {
                    cand_cols.clear();
                    uint64_t nm = 0;
                    for (uint64_t j = 1; j <= n_cols; ++j) {
                        for (uint32_t e = cp[j - 1]; e < cp[j]; ++e) {
                            const uint64_t dist = j - P.pidx[sCol][e];
                            if (dist > limit) cand_cols.push_back(P.pidx[sCol][e]); else nm = std::max(nm, dist);
                        }
                        if (cl[j - 1] & 0x80) { if (j > limit) cand_cols.push_back(0); else nm = std::max(nm, j); }
                    }
                    std::sort(cand_cols.begin(), cand_cols.end());
                    cand_cols.erase(std::unique(cand_cols.begin(), cand_cols.end()), cand_cols.end());
                    if (cand_cols.size() > 32) continue;
                    uint32_t lg = 0;
                    while ((1ull << lg) < span[sRow] + nm + 1 && lg < 14) ++lg;
                    const uint64_t hw = (1ull << lg) * cw, stride = hw + (cw == 8 ? 4 : 8);   // popoa_sys_kernel's row stride
                    const uint64_t bytes = n_rows * stride * 4 + cand_cols.size() * n_rows * cw * 4 + n_cols * 8 + n_pred * 4 + cand_cols.size() * 4 + 16;
                    if (bytes < best_bytes) { best_bytes = bytes; near_limit = limit; near_max = nm; sys_log = lg; far_cols = cand_cols; }
                    if (cand_cols.empty()) break;   // a larger limit only deepens the ring
                }
                sys_bytes = best_bytes;
                take_sys = best_bytes != UINT64_MAX && far_cols.size() <= 32 && (1ull << sys_log) >= span[sRow] + near_max + 1 && sys_bytes <= kSysLdsBytes &&
                           max_deg <= 63 && cp[n_cols] - cp[0] < (1u << 17);   // field widths of the column records
            }
            // strips of rows (popoa_strip_kernel): pairs too large for one workgroup's LDS, one workgroup per strip of S rows, all strips of the pair in
            // flight at once.  Taken when the column predecessors (a source's boundary column included) lie inside a ring of at most 64 columns but for at
            // most eight SAVED columns, the row graph's predecessors within 32 rows (they become the next strip's ghost rows) and the pair is large enough
            bool take_strip = false;
            uint32_t strip_log = 0, strip_S = 0, strip_n = 0, strip_g = 0, strip_limit = 0;
            std::vector<uint32_t> strip_far;
            // rows = the shorter graph; if ITS predecessors reach too far back for ghost rows (a long bubble) and the longer graph's do not, rows = the longer one
            int strip_rows_side = sRow;
            for (int attempt = 0; attempt < 2 && !take_strip; ++attempt) {
                const int sR = attempt == 0 ? sRow : sCol, sC = 1 - sR;
                const uint64_t nRw = (sR == 0 ? d.n1 : d.n2) + 1ull, nCl = sR == 0 ? d.n2 : d.n1;
                if (take_lane || take_sys || g_no_strip || g_force_general || nRw < 192 || cells < 100000 || nCl >= (1u << 28)) continue;
                strip_rows_side = sR;
                strip_far.clear();
                const uint32_t* cp = P.poff[sC].data() + d.node_base[sC];
                const uint8_t* cl = P.lab[sC].data() + d.node_base[sC];
                const uint32_t* rp = P.poff[sR].data() + d.node_base[sR];
                uint64_t max_deg = 0, gd = 0;
                for (uint64_t j = 1; j <= nCl; ++j) max_deg = std::max<uint64_t>(max_deg, cp[j] - cp[j - 1]);
                for (uint64_t i = 1; i < nRw; ++i)
                    for (uint32_t e = rp[i - 1]; e < rp[i]; ++e) gd = std::max<uint64_t>(gd, i - P.pidx[sR][e]);
                // the near limit decides the ring depth (LDS per row) against the number of saved columns (LDS per row as well: a slot each): the candidate with
                // the smallest footprint per row wins, as for the systolic kernel above
                const uint64_t cw = npw == 1 ? 4 : 8;
                uint32_t lg = 15;
                uint64_t row_bytes = UINT64_MAX;
                {
                    const uint64_t limits[6] = {2, 4, 8, 16, 32, 62};
                    std::vector<uint32_t> cand;
                    for (uint64_t limit : limits) {
                        cand.clear();
                        uint64_t nm = 0;
                        for (uint64_t j = 1; j <= nCl; ++j) {
                            for (uint32_t e = cp[j - 1]; e < cp[j]; ++e) {
                                const uint64_t dist = j - P.pidx[sC][e];
                                if (dist > limit) cand.push_back(P.pidx[sC][e]); else nm = std::max(nm, dist);
                            }
                            if (cl[j - 1] & 0x80) { if (j > limit) cand.push_back(0); else nm = std::max(nm, j); }
                        }
                        std::sort(cand.begin(), cand.end());
                        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
                        if (cand.size() > 8) continue;
                        uint32_t l2 = 1;
                        while ((1ull << l2) < span[sR] + nm + 1 && l2 < 14) ++l2;
                        if (l2 > 6) break;   // (a larger limit only deepens the ring)
                        const uint64_t bytes = ((1ull << l2) * cw + (cw == 8 ? 4 : 8)) * 4 + cand.size() * cw * 4;
                        if (bytes < row_bytes) { row_bytes = bytes; lg = l2; strip_limit = (uint32_t)limit; strip_far = cand; }
                        if (cand.empty()) break;
                    }
                }
                if (lg <= 6 && span[sR] <= 32 && gd <= 32 && max_deg <= 63) {
                    // the largest S (a multiple of 64, at most 768: 1 024 threads less the four ghost waves) whose rings and row lists fit
                    auto rec_ring_log = [&](uint64_t n_loc) { uint32_t l = 5; while ((1ull << l) < n_loc + 48) ++l; return l; };   // popoa_strip_kernel's record ring
                    auto strip_bytes = [&](uint64_t S) {
                        uint64_t worst = 0;
                        for (uint64_t a = 0; a < nRw; a += S) {
                            const uint64_t hi = std::min(nRw - 1, a + S - 1), lo = std::max<uint64_t>(a, 1);
                            const uint64_t edges = hi >= lo ? rp[hi] - rp[lo - 1] : 0, n_loc = (a ? gd : 0) + std::min(S, nRw - a);
                            worst = std::max<uint64_t>(worst, n_loc * row_bytes + ((uint64_t)16 << rec_ring_log(n_loc)) + edges * 4 + 64);
                        }
                        return worst;
                    };
                    // CL_STRIP_ROWS=64..768 caps the rows per strip (measurements)
                    static const uint64_t rows_env = [] { const char* e = getenv("CL_STRIP_ROWS"); const long v = e ? atol(e) : 0; return (uint64_t)(v >= 64 && v <= 768 ? v / 64 * 64 : 0); }();
                    uint64_t S = rows_env ? rows_env : 256;   // (measured: a pair's duration hardly depends on the rows per strip between 64 and 768 — a step is a latency chain, not throughput — so strips are kept small: more of them run side by side and each leaves LDS for its neighbours)
                    while (S >= 64 && strip_bytes(S) > kSysLdsBytes) S -= 64;
                    if (S >= 64) {
                        const uint64_t n = (nRw + S - 1) / S;
                        const uint64_t even = ((nRw + n - 1) / n + 63) / 64 * 64;   // the rows dealt evenly
                        if (even <= S && strip_bytes(even) <= kSysLdsBytes) S = even;
                        strip_S = (uint32_t)S; strip_n = (uint32_t)((nRw + S - 1) / S); strip_g = (uint32_t)gd; strip_log = lg;
                        take_strip = strip_n <= 200;   // (all strips of a pair must be resident together: one workgroup per compute unit)
                    }
                }
                if (take_strip) {
                    const uint32_t rec_base = (uint32_t)P.strip_recs.size();
                    for (uint64_t j = 1; j <= nCl; ++j) {
                        const uint32_t b0 = cp[j - 1], deg = cp[j] - cp[j - 1], l = cl[j - 1], src = l >> 7, nq = deg + src;
                        // a predecessor column as a code: that many columns back (below 0x80: inside the ring), or 0x80 | its slot among the saved columns
                        auto code = [&](uint32_t q) -> uint32_t {
                            if ((uint32_t)j - q <= strip_limit) return (uint32_t)j - q;
                            return 0x80u | (uint32_t)(std::lower_bound(strip_far.begin(), strip_far.end(), q) - strip_far.begin());
                        };
                        uint32_t x = 0;
                        if (nq >= 1 && nq <= 3) {   // the straight-line cell's three predecessors (8 bits each)
                            uint32_t q[3], nl = 0;
                            for (uint32_t f = 0; f < deg; ++f) q[nl++] = P.pidx[sC][b0 + f];
                            if (src) q[nl++] = 0u;
                            for (; nl < 3; ++nl) q[nl] = q[0];
                            x = code(q[0]) | (code(q[1]) << 8) | (code(q[2]) << 16);
                        }
                        // the general cell's predecessor list rides along as distances (z: first two, w: next two, x — free when the straight-line
                        // cell does not apply — the fifth and sixth); longer lists are read from HBM
                        const bool fast = nq >= 1 && nq <= 3, inl = deg <= 6;
                        uint32_t dist[6] = {0, 0, 0, 0, 0, 0};
                        for (uint32_t f = 0; f < deg && f < 6; ++f) dist[f] = code(P.pidx[sC][b0 + f]);
                        if (!fast && inl) x = dist[4] | (dist[5] << 12);
                        // bit 15: this column is itself a saved one, bits 12-14: in that slot
                        const auto self = std::lower_bound(strip_far.begin(), strip_far.end(), (uint32_t)j);
                        const uint32_t keep = self != strip_far.end() && *self == (uint32_t)j ? 0x8000u | ((uint32_t)(self - strip_far.begin()) << 12) : 0u;
                        P.strip_recs.push_back(make_uint4(x, keep | (inl ? 1u << 16 : 0u) | (deg << 17) | (fast ? 1u << 23 : 0u) | ((l & 0x7Fu) << 24) | (src << 31),
                                                          dist[0] | (dist[1] << 12), dist[2] | (dist[3] << 12)));
                    }
                    const uint64_t hand_per = (uint64_t)strip_g * (nCl + 1) * (cw / 2);
                    for (uint32_t j = 0; j < strip_n; ++j) {
                        ClStripDesc sd{};
                        const uint64_t a = (uint64_t)j * strip_S;
                        sd.prob = (uint32_t)P.desc.size();
                        sd.n_ghost = j ? strip_g : 0;
                        sd.row_base = (uint32_t)(a - sd.n_ghost);
                        sd.n_real = (uint32_t)std::min<uint64_t>(strip_S, nRw - a);
                        sd.n_out = j + 1 < strip_n ? strip_g : 0;
                        sd.rec_base = rec_base;
                        sd.hand_in = P.hand_words + (j ? (uint64_t)(j - 1) * hand_per : 0);
                        sd.hand_out = P.hand_words + (uint64_t)j * hand_per;
                        sd.prog = (uint32_t)P.strips.size();
                        sd.strip = j; sd.n_strips = strip_n; sd.logH = strip_log;
                        uint32_t lrw = 5;
                        while ((1ull << lrw) < (uint64_t)sd.n_ghost + sd.n_real + 48) ++lrw;
                        sd.logRW = lrw;
                        const uint64_t hi = std::min<uint64_t>(nRw - 1, a + strip_S - 1), lo = std::max<uint64_t>(a, 1);
                        P.strip_lds.push_back((uint32_t)(((uint64_t)sd.n_ghost + sd.n_real) * row_bytes + (16ull << lrw) + (hi >= lo ? rp[hi] - rp[lo - 1] : 0) * 4 + 64));
                        P.strips.push_back(sd);
                    }
                    P.hand_words += (uint64_t)(strip_n - 1) * hand_per;
                }
                        }
            uint64_t depth = 1;   // a power of two (the kernel masks instead of dividing): enough for every read, or all that fits
            while (depth < span[0] + span[1] + 1 && 2 * depth * per_diag + topo_bytes <= kRingLdsBytes && depth < 16384) depth *= 2;
            if (take_lane) {
                d.kind = CL_KIND_LANE;
                d.pad = (uint16_t)(lane_dr | (lane_dc << 4) | ((uint32_t)(lane_groups - 1) << 8) | (d.n2 < d.n1 ? 0x8000u : 0u));   // DR | DC << 4 | groups - 1 << 8 | rows = graph 2
                d.aux_base = (uint32_t)P.sys_aux.size();   // sync | rowrec | rowdist | colrec | coldist (popoa_lane.h)
                d.aux_cnt = lane_slots;
                P.sys_aux.insert(P.sys_aux.end(), lane_words.begin(), lane_words.end());
                P.ring_need.push_back((uint32_t)lane_lds);
                // the hand-off rows between the strips of 64 rows lie behind the planes: [strips - 1][DR][1 + NumPW][columns]; behind them, for a wide pair, the
                // saved-column cells a group hands to the next: [groups][saved columns][DR]
                const uint64_t n_strips = (n_rows - 1 + 63) / 64;
                P.plane_cursor += ((n_strips - 1) * lane_dr * (1 + npw) * n_cols + (lane_groups > 1 ? lane_groups * lane_slots * lane_dr : 0) + 3) / 4 * 4;
            } else if (take_sys) {
                d.kind = CL_KIND_SYS;
                d.pad = (uint16_t)(sys_log | (d.n2 < d.n1 ? 0x8000u : 0u));   // log2 H | rows = graph 2
                d.aux_base = (uint32_t)P.sys_aux.size();   // {near limit, the saved columns ascending}
                d.aux_cnt = (uint32_t)far_cols.size();
                P.sys_aux.push_back((uint32_t)near_limit);
                P.sys_aux.insert(P.sys_aux.end(), far_cols.begin(), far_cols.end());
                P.ring_need.push_back((uint32_t)sys_bytes);
            } else if (take_strip) {
                d.kind = CL_KIND_STRIP;
                d.pad = (uint16_t)(strip_log | (strip_rows_side == 1 ? 0x8000u : 0u));   // bit 15: the rows are graph 2
                d.aux_base = (uint32_t)P.sys_aux.size();   // {near limit, the saved columns ascending}, as for the systolic kernel
                d.aux_cnt = (uint32_t)strip_far.size();
                P.sys_aux.push_back(strip_limit);
                P.sys_aux.insert(P.sys_aux.end(), strip_far.begin(), strip_far.end());
                P.ring_need.push_back(0);
            } else if (!g_no_ring && depth * per_diag + topo_bytes <= kRingLdsBytes && (depth >= 8 || depth >= span[0] + span[1] + 1)) {
                d.pad = (uint16_t)(depth | (depth >= span[0] + span[1] + 1 ? 0x8000u : 0u));   // bit 15: the ring serves every read
                P.ring_need.push_back((uint32_t)(depth * per_diag + topo_bytes));
            } else P.ring_need.push_back(0);
        }
        P.lin_rows.push_back(lr);
        P.lin_waves.push_back(lw);
        P.lin_swap.push_back(ls);
        if (P.out_cursor + d.n1 + d.n2 >= (1ull << 32)) { P.fail(CL_ERR_INVALID_ARGUMENT, "batch too large for 32-bit output offsets"); return; }
        d.out_base = (uint32_t)P.out_cursor;
        P.out_cursor += d.n1 + d.n2;
        P.desc.push_back(d);
        P.po_problem.push_back(k);
        P.dp_cells += cells;
        P.dp_bytes += cells * 4ull * (1 + 2 * npw);
        P.max_cells = std::max<uint64_t>(P.max_cells, cells);
        if (d.kind == CL_KIND_LINEAR) P.n_linear++;
    };
    unsigned hw_threads = std::thread::hardware_concurrency();
    const uint64_t n_parts = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw_threads ? hw_threads : 1, 16), n / 256));
    std::vector<PackPart> parts(n_parts);
    {
        auto work = [&](uint64_t t) {
            Scratch S;
            for (uint64_t k = n * t / n_parts; k < n * (t + 1) / n_parts && !parts[t].rc; ++k) pack_one(k, parts[t], S);
        };
        cl_pool_run((unsigned)n_parts, [&](unsigned t) { work(t); });
    }
    std::vector<uint8_t> lab[2];
    std::vector<uint32_t> poff[2], pidx[2], snk[2];
    poff[0].push_back(0);
    poff[1].push_back(0);
    std::vector<uint32_t> ring_need, sys_aux, strip_lds;
    std::vector<uint4> strip_recs;
    uint64_t hand_words = 0;
    uint64_t plane_cursor = 0, out_cursor = 0;
    for (PackPart& P : parts) {   // in order: the first failure is the one a serial pass would have met
        if (P.rc) { set_error(ctx, "%s", P.err.c_str()); plan_free(pl); return P.rc; }
        for (int s = 0; s < 2; ++s)
            if (lab[s].size() + P.lab[s].size() >= (1ull << 32) || pidx[s].size() + P.pidx[s].size() >= (1ull << 32)) {
                set_error(ctx, "batch too large for 32-bit device offsets");
                plan_free(pl);
                return CL_ERR_INVALID_ARGUMENT;
            }
        if (out_cursor + P.out_cursor >= (1ull << 32)) { set_error(ctx, "batch too large for 32-bit output offsets"); plan_free(pl); return CL_ERR_INVALID_ARGUMENT; }
        const uint32_t lab_base[2] = {(uint32_t)lab[0].size(), (uint32_t)lab[1].size()}, pidx_base[2] = {(uint32_t)pidx[0].size(), (uint32_t)pidx[1].size()};
        const uint32_t snk_base[2] = {(uint32_t)snk[0].size(), (uint32_t)snk[1].size()}, aux_base = (uint32_t)sys_aux.size();
        if (strip_recs.size() + P.strip_recs.size() >= (1ull << 32)) { set_error(ctx, "batch too large for 32-bit device offsets"); plan_free(pl); return CL_ERR_INVALID_ARGUMENT; }
        for (ClStripDesc sd : P.strips) {
            sd.prob += (uint32_t)pl->desc.size();
            sd.rec_base += (uint32_t)strip_recs.size();
            sd.hand_in += hand_words; sd.hand_out += hand_words;
            pl->strips.push_back(sd);
        }
        {   // the progress words are the strips' own indices
            const size_t base = pl->strips.size() - P.strips.size();
            for (size_t i = 0; i < P.strips.size(); ++i) pl->strips[base + i].prog = (uint32_t)(base + i);
        }
        strip_recs.insert(strip_recs.end(), P.strip_recs.begin(), P.strip_recs.end());
        strip_lds.insert(strip_lds.end(), P.strip_lds.begin(), P.strip_lds.end());
        hand_words += P.hand_words;
        for (size_t i = 0; i < P.desc.size(); ++i) {
            ClProbDesc d = P.desc[i];
            for (int s = 0; s < 2; ++s) { d.node_base[s] += lab_base[s]; d.snk_base[s] += snk_base[s]; }
            d.plane_base += plane_cursor;
            d.out_base += (uint32_t)out_cursor;
            if (d.kind == CL_KIND_SYS || d.kind == CL_KIND_STRIP || d.kind == CL_KIND_LANE) d.aux_base += aux_base;
            pl->po_index[P.po_problem[i]] = (int32_t)pl->desc.size();
            pl->desc.push_back(d);
            pl->po_problem.push_back(P.po_problem[i]);
        }
        for (int s = 0; s < 2; ++s) {
            lab[s].insert(lab[s].end(), P.lab[s].begin(), P.lab[s].end());
            for (size_t i = 1; i < P.poff[s].size(); ++i) poff[s].push_back(P.poff[s][i] + pidx_base[s]);
            pidx[s].insert(pidx[s].end(), P.pidx[s].begin(), P.pidx[s].end());
            snk[s].insert(snk[s].end(), P.snk[s].begin(), P.snk[s].end());
            pl->order[s].insert(pl->order[s].end(), P.order[s].begin(), P.order[s].end());
        }
        ring_need.insert(ring_need.end(), P.ring_need.begin(), P.ring_need.end());
        sys_aux.insert(sys_aux.end(), P.sys_aux.begin(), P.sys_aux.end());
        pl->pd_problem.insert(pl->pd_problem.end(), P.pd_problem.begin(), P.pd_problem.end());
        pl->host_problem.insert(pl->host_problem.end(), P.host_problem.begin(), P.host_problem.end());
        pl->lin_rows.insert(pl->lin_rows.end(), P.lin_rows.begin(), P.lin_rows.end());
        pl->lin_waves.insert(pl->lin_waves.end(), P.lin_waves.begin(), P.lin_waves.end());
        pl->lin_swap.insert(pl->lin_swap.end(), P.lin_swap.begin(), P.lin_swap.end());
        plane_cursor += P.plane_cursor;
        out_cursor += P.out_cursor;
        pl->stats.dp_cells += P.dp_cells;
        pl->stats.dp_bytes += P.dp_bytes;
        pl->stats.max_cells = std::max<uint64_t>(pl->stats.max_cells, P.max_cells);
        pl->stats.n_linear += P.n_linear;
        P = PackPart();
    }
    pl->stats.n_problems = n;
    pl->stats.n_po_poa = pl->desc.size();

    // launch groups, largest matrices first inside a group:
    //   linear kernel  : (NumPW, rows per lane, waves per workgroup)
    //   general kernel : (NumPW, workgroup size by widest anti-diagonal)
    std::vector<uint32_t> plist;
    auto cells_of = [&](uint32_t x) { return (uint64_t)(pl->desc[x].n1 + 1) * (pl->desc[x].n2 + 1); };
    auto close_group = [&](LaunchGroup grp) {
        grp.count = (uint32_t)plist.size() - grp.first;
        if (!grp.count) return;
        for (uint32_t i = grp.first; i < plist.size(); ++i) {
            grp.cells += cells_of(plist[i]);
            grp.bytes += cells_of(plist[i]) * 4ull * (1 + 2 * grp.npw);
        }
        std::stable_sort(plist.begin() + grp.first, plist.end(), [&](uint32_t x, uint32_t y) { return cells_of(x) > cells_of(y); });
        // A launch of thousands of workgroups is bound by throughput, not by its longest sweep, and inside a step that shares the device it lasts twice as long as alone
        // (10 x 1 Mbp: 2 651 pairs of the systolic kernel, 0.70 ms alone, 1.45-1.5 ms in the step — the longest launch of the step, alone on its stream).  Such a
        // launch is dealt into PARTS (every parts-th pair of the list by size, so that the parts are alike) which the streams can take separately.
        // CL_STITCH_SPLIT=<workgroups per part> [0: never]
        // Measured (profiles/r05_step_spread.txt): it LOSES — 1.58-1.59 ms per step undivided, 1.70-1.81 in parts of 1 400, 1.87-1.99 in parts of 900 (more launches
        // on the same eight streams cost more than the better balance brings) — so off by default
        static const uint32_t split_at = [] { const char* e = getenv("CL_STITCH_SPLIT"); const long v = e ? atol(e) : 0; return (uint32_t)(v < 0 ? 0 : v); }();
        const uint32_t parts = grp.kind == CL_KIND_SYS && split_at && grp.count > split_at + split_at / 2 ? (grp.count + split_at - 1) / split_at : 1;
        if (parts > 1) {
            std::vector<uint32_t> all(plist.begin() + grp.first, plist.end());
            plist.resize(grp.first);
            for (uint32_t part = 0; part < parts; ++part) {
                LaunchGroup sub = grp;
                sub.first = (uint32_t)plist.size();
                sub.cells = sub.bytes = 0;
                for (size_t i = part; i < all.size(); i += parts) {
                    plist.push_back(all[i]);
                    sub.cells += cells_of(all[i]);
                    sub.bytes += cells_of(all[i]) * 4ull * (1 + 2 * grp.npw);
                }
                sub.count = (uint32_t)plist.size() - sub.first;
                pl->groups.push_back(sub);
            }
            return;
        }
        pl->groups.push_back(grp);
    };
    // two pairs per wave only where it pays (duos_forced): the pairs marked for it run one per wave otherwise — same codes, same workspace
    if (duos_forced() < 0) {
        uint64_t marked = 0;
        for (uint32_t i = 0; i < pl->desc.size(); ++i) marked += pl->desc[i].kind == CL_KIND_LINEAR && pl->lin_waves[i] == 5;
        if (!(n < 2048 || 2 * marked >= pl->desc.size()))
            for (uint32_t i = 0; i < pl->desc.size(); ++i)
                if (pl->desc[i].kind == CL_KIND_LINEAR && pl->lin_waves[i] == 5) pl->lin_waves[i] = 1;
    }
    // chain kernel: one launch per workgroup shape; the problems are ordered by the length of their sweep
    const int lin_waves[5] = {16, 8, 4, 3, 1};
    for (int gi = 0; gi < 5; ++gi) {
        LaunchGroup grp;
        grp.kind = CL_KIND_LINEAR; grp.npw = 0; grp.waves = lin_waves[gi];
        grp.first = (uint32_t)plist.size();
        for (uint32_t i = 0; i < pl->desc.size(); ++i)
            if (pl->desc[i].kind == CL_KIND_LINEAR && pl->lin_waves[i] == grp.waves) plist.push_back(i);
        grp.count = (uint32_t)plist.size() - grp.first;
        if (!grp.count) continue;
        for (uint32_t i = grp.first; i < plist.size(); ++i) {
            grp.cells += cells_of(plist[i]);
            grp.bytes += cells_of(plist[i]) * 4ull * (1 + 2 * pl->desc[plist[i]].npw);
        }
        auto sweep = [&](uint32_t x) { return (uint64_t)pl->desc[x].n1 + pl->desc[x].n2 + (pl->desc[x].npw == 3 ? 64 : 0); };
        std::stable_sort(plist.begin() + grp.first, plist.end(), [&](uint32_t x, uint32_t y) { return sweep(x) > sweep(y); });
        pl->groups.push_back(grp);
    }
    // small chain pairs, four per wave (popoa_linear_quad_kernel): quads of one NumPW, longest first so that a quad's pairs are about as long as one another
    {
      for (int marker : {2, 5}) {   // 2: four pairs per wave (quads), 5: two (duos)
        LaunchGroup grp;
        grp.kind = CL_KIND_LINEAR; grp.npw = 0; grp.waves = marker;
        grp.first = (uint32_t)plist.size();
        for (int npw = 3; npw >= 1; --npw) {
            std::vector<uint32_t> q;
            for (uint32_t i = 0; i < pl->desc.size(); ++i)
                if (pl->desc[i].kind == CL_KIND_LINEAR && pl->lin_waves[i] == marker && pl->desc[i].npw == npw) q.push_back(i);
            std::stable_sort(q.begin(), q.end(), [&](uint32_t x, uint32_t y) { return std::max(pl->desc[x].n1, pl->desc[x].n2) > std::max(pl->desc[y].n1, pl->desc[y].n2); });
            for (uint32_t i : q) {
                plist.push_back(i);
                grp.cells += cells_of(i);
                grp.bytes += cells_of(i) * 4ull * (1 + 2 * npw);
            }
            while ((plist.size() - grp.first) % (marker == 2 ? 4 : 2)) plist.push_back(0xFFFFFFFFu);
        }
        grp.count = (uint32_t)plist.size() - grp.first;
        if (grp.count) pl->groups.push_back(grp);
      }
    }
    // near-chain pairs in registers: one launch per workgroup shape as well (strips of 64 rows over 1 / 4 / 16 waves), longest sweep first
    // The LONG sweeps (1 024 steps and more: a handful of pairs that bound the pass) get launches of their own whose workgroups ask for more than half a compute
    // unit's LDS: one workgroup per compute unit, so that a long sweep's waves do not share their SIMDs with another pair's (190 registers per lane leave room for two
    // waves per SIMD, and two waves on a SIMD each run at half speed — for a launch that lasts as long as its longest sweep that doubles its duration).
    // CL_LANE_LONG=1 switches the split on.  Measured (10 x 1 Mbp, the timed step): it does not pay — 2.69 ms per step with it, 2.46 without at the default
    // routing threshold: the compute units of the long launches are shared with the OTHER kernels' workgroups anyway, and one launch more queues on the streams
    static const bool lane_long_split = [] { const char* e = getenv("CL_LANE_LONG"); return e && e[0] == '1'; }();
    for (int gi = 0; gi < 6; ++gi) {
        LaunchGroup grp;
        const int lane_waves[3] = {0, (int)lane_rw, 1};   // (one wave per SIMD; no launches of eight waves any more)
        const bool long_ones = gi < 3;
        grp.kind = CL_KIND_LANE; grp.npw = 0; grp.waves = lane_waves[gi % 3];
        grp.first = (uint32_t)plist.size();
        for (uint32_t i = 0; i < pl->desc.size(); ++i) {
            const ClProbDesc& d = pl->desc[i];
            const uint32_t rows = std::min(d.n1, d.n2);
            const bool is_long = lane_long_split && (uint64_t)d.n1 + d.n2 >= 1024;
            if (d.kind == CL_KIND_LANE && !((d.pad >> 8) & 0x7Fu) && (rows <= 64 ? 1 : (int)lane_rw) == grp.waves && is_long == long_ones) { plist.push_back(i); grp.ring_bytes = std::max<uint32_t>(grp.ring_bytes, ring_need[i]); }
        }
        grp.count = (uint32_t)plist.size() - grp.first;
        if (!grp.count) continue;
        if (long_ones) grp.ring_bytes = std::max<uint32_t>(grp.ring_bytes, 81u * 1024u);   // more than half of 160 KB: one workgroup per compute unit
        for (uint32_t i = grp.first; i < plist.size(); ++i) {
            grp.cells += cells_of(plist[i]);
            grp.bytes += cells_of(plist[i]) * 4ull * (1 + 2 * pl->desc[plist[i]].npw);
        }
        auto sweep = [&](uint32_t x) { return (uint64_t)pl->desc[x].n1 + pl->desc[x].n2 + 96ull * (std::min(pl->desc[x].n1, pl->desc[x].n2) / 64); };
        std::stable_sort(plist.begin() + grp.first, plist.end(), [&](uint32_t x, uint32_t y) { return sweep(x) > sweep(y); });
        pl->groups.push_back(grp);
    }
    // wide near-chain pairs: a pair's groups are consecutive workgroups of one launch (they wait for one another: at most 224 workgroups per launch, whole pairs);
    // their progress / done words are a run of the plan's lane_sync array, zeroed in front of the launch
    uint32_t lane_sync_words = 0;
    {
        LaunchGroup grp;
        auto open = [&]() { grp = LaunchGroup(); grp.kind = CL_KIND_LANE; grp.npw = 0; grp.waves = 4; grp.block = 1; grp.first = (uint32_t)plist.size(); grp.prog_first = lane_sync_words; };
        auto close = [&]() {
            grp.count = (uint32_t)plist.size() - grp.first;
            grp.prog_count = lane_sync_words - grp.prog_first;
            if (grp.count) pl->groups.push_back(grp);
        };
        open();
        std::vector<uint32_t> wide;
        for (uint32_t i = 0; i < pl->desc.size(); ++i) if (pl->desc[i].kind == CL_KIND_LANE && ((pl->desc[i].pad >> 8) & 0x7Fu)) wide.push_back(i);
        std::stable_sort(wide.begin(), wide.end(), [&](uint32_t x, uint32_t y) { return (uint64_t)pl->desc[x].n1 + pl->desc[x].n2 > (uint64_t)pl->desc[y].n1 + pl->desc[y].n2; });
        for (uint32_t i : wide) {
            const ClProbDesc& d = pl->desc[i];
            const uint32_t ng = ((d.pad >> 8) & 0x7Fu) + 1u;
            if ((uint32_t)plist.size() - grp.first + ng > 224) { close(); open(); }
            sys_aux[d.aux_base] = lane_sync_words;
            lane_sync_words += 2 * ng;
            for (uint32_t g2 = 0; g2 < ng; ++g2) plist.push_back(i);
            grp.ring_bytes = std::max<uint32_t>(grp.ring_bytes, ring_need[i]);
            grp.cells += cells_of(i);
            grp.bytes += cells_of(i) * 4ull * (1 + 2 * d.npw);
        }
        close();
    }
    // chain pairs that span several workgroups (popoa_linear_span_kernel): as the wide pairs above — a pair's groups are consecutive workgroups of one launch, at most 224
    // workgroups per launch, whole pairs; progress / done words in the same lane_sync array (ClProbDesc::aux_base = the pair's first word, aux_cnt = its groups)
    {
        LaunchGroup grp;
        auto open = [&]() { grp = LaunchGroup(); grp.kind = CL_KIND_LINEAR; grp.npw = 0; grp.waves = 40; grp.first = (uint32_t)plist.size(); grp.prog_first = lane_sync_words; };
        auto close = [&]() {
            grp.count = (uint32_t)plist.size() - grp.first;
            grp.prog_count = lane_sync_words - grp.prog_first;
            if (grp.count) pl->groups.push_back(grp);
        };
        open();
        std::vector<uint32_t> span;
        for (uint32_t i = 0; i < pl->desc.size(); ++i) if (pl->desc[i].kind == CL_KIND_LINEAR && pl->lin_waves[i] == 40) span.push_back(i);
        std::stable_sort(span.begin(), span.end(), [&](uint32_t x, uint32_t y) { return (uint64_t)pl->desc[x].n1 + pl->desc[x].n2 > (uint64_t)pl->desc[y].n1 + pl->desc[y].n2; });
        for (uint32_t i : span) {
            ClProbDesc& d = pl->desc[i];
            const uint32_t ng = cl_linear_span_groups(std::min(d.n1, d.n2));
            if ((uint32_t)plist.size() - grp.first + ng > 224) { close(); open(); }
            d.aux_base = lane_sync_words;
            d.aux_cnt = ng;
            lane_sync_words += 2 * ng;
            for (uint32_t g2 = 0; g2 < ng; ++g2) plist.push_back(i);
            grp.cells += cells_of(i);
            grp.bytes += cells_of(i) * 4ull * (1 + 2 * d.npw);
        }
        close();
    }
    lap("pack + route");
    const int blocks[3] = {64, 256, 1024};
    for (int bi = 2; bi >= 0; --bi)
        for (int npw = 3; npw >= 1; --npw)
            for (int ring = 2; ring >= 0; --ring) {   // 2: ring above 64 KB of LDS (one workgroup per CU), 1: ring up to 64 KB, 0: planes in HBM
                LaunchGroup grp;
                grp.kind = CL_KIND_GENERAL; grp.npw = npw; grp.block = blocks[bi];
                grp.first = (uint32_t)plist.size();
                for (uint32_t i = 0; i < pl->desc.size(); ++i) {
                    const ClProbDesc& d = pl->desc[i];
                    uint32_t width = std::min(d.n1, d.n2) + 1;
                    int b = width <= 64 ? 0 : width <= 256 ? 1 : 2;
                    const int cls = d.pad == 0 ? 0 : ring_need[i] > 64 * 1024 ? 2 : 1;
                    if (d.kind == CL_KIND_GENERAL && d.npw == npw && b == bi && cls == ring) {
                        plist.push_back(i);
                        if (ring) grp.ring_bytes = std::max<uint32_t>(grp.ring_bytes, ring_need[i]);
                    }
                }
                close_group(grp);
            }
    // systolic DAG kernel: (NumPW, workgroup size, LDS class) — a launch's dynamic LDS is that of its hungriest problem, and LDS decides
    // how many workgroups share a CU, so the many small problems must not ride with the few large ones
    // CL_STITCH_LDS_CLASSES=a,b,c,d (KB; four boundaries below the ceiling) for measurements
    uint32_t lds_class[5] = {12 * 1024, 32 * 1024, 64 * 1024, 100 * 1024, (uint32_t)kSysLdsBytes};
    if (const char* e = getenv("CL_STITCH_LDS_CLASSES")) {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (sscanf(e, "%u,%u,%u,%u", &a, &b, &c, &d) == 4 && a < b && b < c && c < d && d * 1024u < (uint32_t)kSysLdsBytes) { lds_class[0] = a * 1024; lds_class[1] = b * 1024; lds_class[2] = c * 1024; lds_class[3] = d * 1024; }
    }
    // ... unless the smaller ones are so few that all workgroups of the merged launch are resident at once anyway: a launch lasts as long as its longest sweep
    // whatever rides along, and every launch less is a millisecond less on the stream it would have occupied (the streams, not the device, are what a step
    // runs out of: 10 x 1 Mbp, sixteen launches a step on eight streams).  CL_STITCH_MERGE_LDS=0: one launch per class as in rounds 2-3 (A/B)
    static const bool merge_lds = [] { const char* e = getenv("CL_STITCH_MERGE_LDS"); return !e || e[0] != '0'; }();
    // workgroup sizes of the systolic kernel: 64 / 128 / 256 / 1024 threads (round 6: pairs of 65-128 rows — two thirds of the subproblems that used to take 256 threads in
    // the step of 10 x 1 Mbp — get two waves instead of four: a barrier between two waves, no idle waves at it).  CL_SYS_BLOCK128=1 for measurements
    static const bool block128 = [] { const char* e = getenv("CL_SYS_BLOCK128"); return e && e[0] == '1'; }();   // (measured and lost: 1.51-1.52 -> 1.61-1.68 ms per step — two more launches on the step's streams; off unless asked for)
    // (one wave with two / four rows per lane — popoa_sysr_kernel, commit 2b07e99 — was measured and lost as well: 3.37 against 1.56 ms per step, profiles/r06_rows_per_lane_ab.txt)
    const int sys_blocks[4] = {64, 128, 256, 1024};
    for (int bi = 3; bi >= 0; --bi)
        for (int npw = 3; npw >= 1; --npw) {
            std::vector<uint32_t> of_class[5];
            for (uint32_t i = 0; i < pl->desc.size(); ++i) {
                const ClProbDesc& d = pl->desc[i];
                const uint32_t rows = std::min(d.n1, d.n2) + 1;
                const int b = rows <= 64 ? 0 : (rows <= 128 && block128) ? 1 : rows <= 256 ? 2 : 3;
                if (d.kind != CL_KIND_SYS || d.npw != npw || b != bi) continue;
                int c = 0;
                while (c < 4 && ring_need[i] > lds_class[c]) ++c;
                of_class[c].push_back(i);
            }
            LaunchGroup grp;
            auto open = [&]() { grp = LaunchGroup(); grp.kind = CL_KIND_SYS; grp.npw = npw; grp.block = sys_blocks[bi]; grp.first = (uint32_t)plist.size(); };
            open();
            for (int lc = 4; lc >= 0; --lc) {
                if (of_class[lc].empty()) continue;
                const size_t have = plist.size() - grp.first;
                if (have) {
                    auto per_cu_at = [&](uint32_t lds) { return std::max<uint64_t>(1, std::min<uint64_t>((160u * 1024u) / std::max<uint32_t>(lds, 1u), 2048u / (uint32_t)sys_blocks[bi])); };
                    uint32_t own = 0;
                    for (uint32_t i : of_class[lc]) own = std::max(own, ring_need[i]);
                    // all resident at once.  (Also merging classes that leave the newcomers' workgroups per compute unit unchanged was measured and lost: 2.75 against
                    // 2.55 ms per step — the long sweeps of the two classes then share a launch's tail)
                    (void)own;
                    const bool fits = have + of_class[lc].size() <= 256 * per_cu_at(grp.ring_bytes);
                    static const bool merge_all = [] { const char* e = getenv("CL_STITCH_MERGE_ALL"); return e && e[0] == '1'; }();   // (measurement: one launch per kernel shape whatever the LDS classes)
                    if (!merge_all && (!merge_lds || !fits)) { close_group(grp); open(); }
                }
                for (uint32_t i : of_class[lc]) {
                    plist.push_back(i);
                    grp.ring_bytes = std::max<uint32_t>(grp.ring_bytes, ring_need[i]);
                }
            }
            close_group(grp);
        }
    // strips of rows: every strip of a pair is a workgroup of the same launch (they wait for one another: at most 224 workgroups per launch, whole pairs)
    for (int npw = 3; npw >= 1; --npw) {
        LaunchGroup grp;
        auto open = [&]() { grp = LaunchGroup(); grp.kind = CL_KIND_STRIP; grp.npw = npw; grp.first = (uint32_t)pl->strip_list.size(); grp.prog_first = UINT32_MAX; };
        auto close = [&]() {
            grp.count = (uint32_t)pl->strip_list.size() - grp.first;
            if (grp.count) pl->groups.push_back(grp);
        };
        open();
        for (size_t i = 0; i < pl->strips.size();) {
            const ClStripDesc& s0 = pl->strips[i];
            const ClProbDesc& d = pl->desc[s0.prob];
            if (d.npw != npw) { i += s0.n_strips; continue; }
            if ((uint32_t)pl->strip_list.size() - grp.first + s0.n_strips > 224) { close(); open(); }
            // a strip's progress word = its position in the launch lists: the words of a launch are one run, zeroed by that launch alone (launches of
            // other NumPW run beside it on other streams and must not have their words touched)
            if (grp.prog_first == UINT32_MAX) grp.prog_first = (uint32_t)pl->strip_list.size();
            uint32_t max_rows = 0;
            for (uint32_t j = 0; j < s0.n_strips; ++j) {
                pl->strips[i + j].prog = (uint32_t)pl->strip_list.size();
                pl->strip_list.push_back((uint32_t)(i + j));
                grp.ring_bytes = std::max(grp.ring_bytes, strip_lds[i + j]);
                max_rows = std::max(max_rows, pl->strips[i + j].n_real);
            }
            grp.prog_count = (uint32_t)pl->strip_list.size() - grp.prog_first;
            grp.block = std::max<int>(grp.block, 256 + (int)((max_rows + 63) / 64 * 64));
            const uint64_t cells = (uint64_t)(d.n1 + 1) * (d.n2 + 1);
            grp.cells += cells;
            grp.bytes += cells * 4ull * (1 + 2 * npw);
            grp.est_cost = std::max<uint64_t>(grp.est_cost, (uint64_t)d.n1 + d.n2 + 96ull * s0.n_strips);
            i += s0.n_strips;
            ++cl_fallbacks.strip_pairs;
        }
        close();
    }
    // longest-running launch first: a group's duration is set by its longest anti-diagonal sweep
    {
        auto crit = [&](const LaunchGroup& g) {
            if (g.kind == CL_KIND_STRIP) return g.est_cost;
            if (g.kind == CL_KIND_LANE && g.block == 1) {   // wide: every 64 rows add a lag of three chunks
                uint64_t c = 0;
                for (uint32_t i = g.first; i < g.first + g.count; ++i) {
                    const ClProbDesc& d = pl->desc[plist[i]];
                    c = std::max<uint64_t>(c, (uint64_t)std::max(d.n1, d.n2) + 5ull * std::min(d.n1, d.n2) / 2);
                }
                return c / 3;
            }
            uint64_t c = 0;
            if (g.kind == CL_KIND_LINEAR && g.waves == 40) {   // one round over several workgroups: columns + 1.5 x rows steps
                for (uint32_t i = g.first; i < g.first + g.count; ++i) {
                    const ClProbDesc& d = pl->desc[plist[i]];
                    c = std::max<uint64_t>(c, (uint64_t)std::max(d.n1, d.n2) + 3ull * std::min(d.n1, d.n2) / 2);
                }
                return c / 3;
            }
            for (uint32_t i = g.first; i < g.first + g.count; ++i)
                if (plist[i] != 0xFFFFFFFFu) c = std::max<uint64_t>(c, (uint64_t)pl->desc[plist[i]].n1 + pl->desc[plist[i]].n2);
            // microseconds per anti-diagonal step, roughly: planes in HBM 4-8, LDS ring 1.6-2.5, systolic DAG 0.45, chain 0.2-0.6
            return c * (g.kind == CL_KIND_GENERAL ? (g.ring_bytes ? 5 : 16) : 1);
        };
        for (LaunchGroup& g : pl->groups) g.est_cost = crit(g);
        std::stable_sort(pl->groups.begin(), pl->groups.end(), [&](const LaunchGroup& x, const LaunchGroup& y) { return x.est_cost > y.est_cost; });
    }
    pl->stats.n_launches = pl->groups.size();
    lap("launch groups");

    // HBM
    pl->plist_host = plist;
    // (the eleven copies are enqueued and waited for once: the host arrays live to the end of this function)
    if ((rc = pl->d_desc.upload_async(ctx, pl->desc)) || (rc = pl->d_plist.upload_async(ctx, plist)) || (rc = pl->d_aux.upload_async(ctx, sys_aux))) { plan_free(pl); return rc; }
    for (int s = 0; s < 2; ++s)
        if ((rc = pl->d_lab[s].upload_async(ctx, lab[s])) || (rc = pl->d_poff[s].upload_async(ctx, poff[s])) ||
            (rc = pl->d_pidx[s].upload_async(ctx, pidx[s])) || (rc = pl->d_snk[s].upload_async(ctx, snk[s]))) { plan_free(pl); return rc; }
    if (!pl->strips.empty() && ((rc = pl->d_strips.upload_async(ctx, pl->strips)) || (rc = pl->d_strip_recs.upload_async(ctx, strip_recs)) ||
                                (rc = pl->d_strip_list.upload_async(ctx, pl->strip_list)) || (rc = pl->d_progress.alloc(ctx, pl->strips.size())) ||
                                (rc = pl->d_handoff.alloc(ctx, hand_words)))) { plan_free(pl); return rc; }
    if (lane_sync_words && (rc = pl->d_lane_sync.alloc(ctx, lane_sync_words))) { plan_free(pl); return rc; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { cl_set_error(ctx, "upload failed"); plan_free(pl); return CL_ERR_HIP; }
    { static const bool dbg = [] { const char* e = getenv("CL_STRIP_DEBUG_FAIL"); return e && *e == '1'; }(); pl->sdev.debug_fail = dbg ? 1u : 0u; }
    pl->sdev.strips = pl->d_strips.p; pl->sdev.recs = pl->d_strip_recs.p; pl->sdev.handoff = pl->d_handoff.p; pl->sdev.progress = pl->d_progress.p;
    if ((rc = pl->d_planes.alloc(ctx, plane_cursor)) || (rc = pl->d_out_pairs.alloc(ctx, out_cursor)) ||
        (rc = pl->d_out_len.alloc(ctx, pl->desc.size())) || (rc = pl->d_out_status.alloc(ctx, pl->desc.size())) ||
        (rc = pl->d_out_score.alloc(ctx, pl->desc.size()))) { plan_free(pl); return rc; }
    pl->stats.workspace_bytes = plane_cursor * 4 + out_cursor * 8 + lab[0].size() + lab[1].size() +
                                4 * (poff[0].size() + poff[1].size() + pidx[0].size() + pidx[1].size());
    pl->dev.desc = pl->d_desc.p;
    for (int s = 0; s < 2; ++s) {
        pl->dev.lab[s] = pl->d_lab[s].p; pl->dev.poff[s] = pl->d_poff[s].p; pl->dev.pidx[s] = pl->d_pidx[s].p; pl->dev.snk[s] = pl->d_snk[s].p;
    }
    pl->dev.aux = pl->d_aux.p;
    pl->dev.planes = pl->d_planes.p;
    pl->dev.out_pairs = pl->d_out_pairs.p;
    pl->dev.out_len = pl->d_out_len.p;
    pl->dev.out_score = pl->d_out_score.p;
    pl->dev.out_status = pl->d_out_status.p;
    { const char* e = getenv("CL_SPAN_DEBUG_FAIL"); pl->dev.debug_span_fail = e && e[0] == '1' ? 1 : 0; }   // test hook: every spanning chain pair reports status 9
    { const char* e = getenv("CL_DEBUG_SKIP_TRACEBACK"); pl->dev.skip_traceback = e ? atoi(e) : 0; }   // measurement hook: 1 no traceback, 3 also no plane stores in the systolic kernel
    pl->sparams.match = (int32_t)ap.match;
    pl->sparams.mismatch = (int32_t)ap.mismatch;
    for (int k = 0; k < 3; ++k) { pl->sparams.oe[k] = (int32_t)(ap.gap_open[k] + ap.gap_extend[k]); pl->sparams.ext[k] = (int32_t)ap.gap_extend[k]; }
    if (hipEventCreate(&pl->ev_start) != hipSuccess || hipEventCreate(&pl->ev_stop) != hipSuccess) {
        set_error(ctx, "hipEventCreate failed");
        plan_free(pl);
        return CL_ERR_HIP;
    }
    lap("allocate + upload");
    *plan_out = pl;
    return CL_OK;
}

// auxiliary streams a plan spreads its launches over (CL_STITCH_STREAMS, 1..kNumAuxStreams): many contexts at once share the process's
// hardware queues, and launches of different streams that land on one queue run one after the other
static const int g_plan_streams = [] { const char* e = getenv("CL_STITCH_STREAMS"); int v = e ? atoi(e) : 0; return v >= 1 && v <= kNumAuxStreams ? v : kNumAuxStreams; }();

// Enqueue every launch group of the plan as a fork/join over the auxiliary streams; `timed` adds per-launch
// HIP events (used by the profiled path only: event records cost host time and are not capturable everywhere).
static hipError_t launch_group(const LaunchGroup& g, cl_stitch_plan* pl, const ClDeviceBatch& dev, hipStream_t stream) {
    if (g.kind == CL_KIND_LINEAR && g.waves == 40) {   // the pairs' progress / done words start every pass at zero
        hipError_t e = hipMemsetAsync(pl->d_lane_sync.p + g.prog_first, 0, (size_t)g.prog_count * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        return cl_launch_popoa_linear_span(g.count, dev, pl->d_plist.p + g.first, pl->sparams, pl->d_lane_sync.p, stream);
    }
    if (g.kind == CL_KIND_LINEAR) return cl_launch_popoa_linear(g.waves == 2 ? 0 : g.waves == 5 ? -2 : g.waves, g.count, dev, pl->d_plist.p + g.first, pl->sparams, stream);
    if (g.kind == CL_KIND_STRIP) {
        // the strips' progress words start every pass at zero (the strips of a launch poll one another's)
        hipError_t e = hipMemsetAsync(pl->d_progress.p + g.prog_first, 0, (size_t)g.prog_count * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        return cl_launch_popoa_strip(g.npw, (uint32_t)g.block, g.count, g.ring_bytes, dev, pl->sdev, pl->d_strip_list.p + g.first, pl->sparams, stream);
    }
    if (g.kind == CL_KIND_LANE && g.block == 1) {
        hipError_t e = hipMemsetAsync(pl->d_lane_sync.p + g.prog_first, 0, (size_t)g.prog_count * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        return cl_launch_popoa_lane(g.waves, g.count, g.ring_bytes, dev, pl->d_plist.p + g.first, pl->sparams, pl->d_lane_sync.p, stream);
    }
    if (g.kind == CL_KIND_LANE) return cl_launch_popoa_lane(g.waves, g.count, g.ring_bytes, dev, pl->d_plist.p + g.first, pl->sparams, nullptr, stream);
    if (g.kind == CL_KIND_SYS) return cl_launch_popoa_sys(g.npw, g.block, g.count, g.ring_bytes, dev, pl->d_plist.p + g.first, pl->sparams, stream);
    return cl_launch_popoa_general(g.npw, g.block, g.count, g.ring_bytes, dev, pl->d_plist.p + g.first, pl->sparams, stream);
}

// a launch's two tick words (popoa_device.h: pass << 48 | 48 bits of ~start, of end; 100 MHz) -> its duration in microseconds, 0 if the words are not of one pass
static double tick_span_us(unsigned long long w0, unsigned long long w1) {
    constexpr unsigned long long mask = 0xFFFFFFFFFFFFull;
    if (!w0 || !w1 || (w0 >> 48) != (w1 >> 48)) return 0.0;
    const unsigned long long start = ~w0 & mask, end = w1 & mask;
    return end > start ? (double)(end - start) * 1e-2 : 0.0;
}

// The context's stream waits for the stitch launches that were handed to the auxiliary streams.  A pass of a plan does NOT do this by itself any more (second half of
// round 5): the launches of a resident plan's next pass follow this pass's launches in STREAM order (a launch group goes to the same stream in every pass and reads
// and writes nothing but its own subproblems), so passes that are enqueued back to back overlap — a stream starts pass i + 1 when ITS launches of pass i are done —
// instead of leaving every stream idle until the slowest one and two event hops are through (0.25-0.5 ms of 2.3 per step of 10 x 1 Mbp: scripts/dev/step_timeline.py).
// Whatever needs a pass to be complete joins first: cl_stitch_plan_sync / _collect / _execute_profiled / _destroy, a re-dealing of the launches, the chaining DP.
extern "C++" int cl_stitch_join(cl_context* ctx) {   // (C++ linkage: declared in cl_internal.hpp for the chaining DP, not part of the C ABI)
    for (int si = 0; si < kNumAuxStreams && ctx->stitch_join_pending; ++si)
        if (ctx->stitch_join_pending & (1u << si)) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[si], 0));
            ctx->stitch_join_pending &= ~(1u << si);
        }
    return CL_OK;
}
static int plan_mark_stop(cl_context* ctx, cl_stitch_plan* pl) {
    int rc = cl_stitch_join(ctx);
    if (rc) return rc;
    if (pl->stop_pending) { HIP_TRY(ctx, hipEventRecord(pl->ev_stop, ctx->stream)); pl->stop_pending = false; }
    return CL_OK;
}

static int enqueue_groups(cl_context* ctx, cl_stitch_plan* pl, bool timed, bool join_now = false) {
    if (timed) {
        // the profiled pass: every launch ALONE on the context's stream, the device idle before and after, timed by the HOST's clock round
        // launch + wait.  HIP event pairs read about twice the kernel's duration in the rocprofv3 kernel trace here (round 4: 2.87 ms by events,
        // 1.43 ms in the trace for the dominant launch, also with nothing else on the device); the host's clock round launch + wait reads the
        // kernel plus ~20 us of launch and wake-up latency, i.e. agrees with the trace within a few per cent for the launches that matter
        // (a launch that finds the device idle runs at about half speed — the dominant launch 2.9 ms against 1.45 ms behind another launch, with the
        // shader clock already at its top in both cases: whatever ramps, ramps within the first launch — so the duration reported is that of a launch
        // behind another one, which is what a launch inside a step is, and what the rocprofv3 kernel trace of a step shows)
        { int rc = plan_mark_stop(ctx, pl); if (rc) return rc; }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const size_t ng = pl->groups.size();
        if (!pl->d_ticks.p) { int rc = pl->d_ticks.alloc(ctx, 2 * ng); if (rc) return rc; }
        for (size_t gi = 0; gi < ng; ++gi) {
            LaunchGroup& g = pl->groups[gi];
            // the kernel's own clock: first workgroup's start, last workgroup's end (s_memrealtime, 100 MHz) — what the rocprofv3 kernel trace shows;
            // beside it the host's clock round launch + wait
            ClDeviceBatch dev = pl->dev;
            dev.ticks = pl->d_ticks.p + 2 * gi;
            dev.tick_pass = 0xFFFFu;
            HIP_TRY(ctx, launch_group(g, pl, dev, ctx->stream));   // (first run: whatever ramps with the device idle, ramps here)
            HIP_TRY(ctx, hipMemsetAsync(dev.ticks, 0, 2 * sizeof(unsigned long long), ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            const auto t0 = std::chrono::steady_clock::now();
            HIP_TRY(ctx, launch_group(g, pl, dev, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            g.host_idle_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            unsigned long long got[2] = {0, 0};
            HIP_TRY(ctx, cl_copy_sync(ctx, got, dev.ticks, sizeof(got), hipMemcpyDeviceToHost));
            const double us = tick_span_us(got[0], got[1]);
            g.host_ms = us > 0 ? (float)(us * 1e-3) : g.host_idle_ms;
        }
        return CL_OK;
    }
    // every launch also leaves its own clock's start / end (two atomics per workgroup): cl_stitch_plan_launch_info reports the durations of the LAST pass,
    // i.e. of launches that overlap with one another as they do in production
    // (the words are not zeroed between passes: a pass's number in their top bits makes its clocks win over every earlier pass's; zeroed when the number wraps and
    // after the profiled pass, whose launches carry the highest number)
    const bool fresh = !pl->d_ticks.p;
    if (fresh && !pl->groups.empty()) { int rc = pl->d_ticks.alloc(ctx, 2 * pl->groups.size()); if (rc) return rc; }
    if (pl->d_ticks.p && (fresh || pl->tick_pass == 0 || pl->tick_pass >= 0xFFFEu)) {
        int rc = plan_mark_stop(ctx, pl);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(pl->d_ticks.p, 0, 2 * pl->groups.size() * sizeof(unsigned long long), ctx->stream));
        pl->tick_pass = 0;
    }
    ++pl->tick_pass;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    bool used[kNumAuxStreams] = {};
    // the groups come longest first (cl_stitch_plan_create: estimated duration = longest sweep x the kernel's time per step); each goes to the
    // stream with the least work so far (longest-processing-time rule), so that the one or two long sweeps of a batch start at once on streams
    // of their own and the short launches queue up beside, not behind, them.  CL_STITCH_SCHED=rr deals them round-robin (rounds 1-3; A/B).
    static const bool round_robin = [] { const char* e = getenv("CL_STITCH_SCHED"); return e && e[0] == 'r'; }();
    const int n_streams = std::min(g_plan_streams, ctx->n_aux);
    uint64_t load[kNumAuxStreams] = {};
    std::vector<int> stream_of(pl->groups.size());
    for (size_t gi = 0; gi < pl->groups.size(); ++gi) {
        int si = (int)(gi % (size_t)n_streams);
        if (!round_robin) {
            si = 0;
            for (int t = 1; t < n_streams; ++t) if (load[t] < load[si]) si = t;
            load[si] += std::max<uint64_t>(1, pl->groups[gi].est_cost);
        }
        stream_of[gi] = si;
    }
    static const bool sched_log = getenv("CL_STITCH_SCHED_LOG") != nullptr;   // the deal, once per change of the groups' order: stream, estimated duration, workgroups, kind
    if (sched_log && pl->tick_pass <= 1) {
        fprintf(stderr, "[stitch plan] launches over %d streams (calibrated %d, re-dealings left %d):", n_streams, (int)pl->calibrated, pl->recalibrations_left);
        for (int t = 0; t < n_streams; ++t) {
            fprintf(stderr, "  s%d %llu us [", t, (unsigned long long)load[t]);
            for (size_t gi = 0; gi < pl->groups.size(); ++gi)
                if (stream_of[gi] == t) fprintf(stderr, " k%d/w%d/b%d x%u:%llu", pl->groups[gi].kind, pl->groups[gi].waves, pl->groups[gi].block, pl->groups[gi].count, (unsigned long long)pl->groups[gi].est_cost);
            fprintf(stderr, " ]");
        }
        fprintf(stderr, "\n");
    }
    // In which order the launches are ISSUED: the device starts about four launches at a time (a hardware queue that is handing out a launch's workgroups is busy until
    // the last of them has a compute unit — for the 34 716 workgroups of the short chain pairs that is the launch's whole duration), and in the step of 10 x 1 Mbp the
    // launch that ends the step — 66 long near-chain pairs, 1.27 ms — was started 0.43 ms after the first one (scripts/dev/step_timeline.py).  Launches of few workgroups
    // (at most two per compute unit: resident at once, their queue is free again at once) therefore go first, the wide ones after them; each on the stream the
    // longest-processing-time rule gave it.  CL_STITCH_ORDER=cost: in the order of their durations (up to the first half of round 5)
    static const bool light_first = [] { const char* e = getenv("CL_STITCH_ORDER"); return !(e && e[0] == 'c'); }();
    for (int pass = 0; pass < 2; ++pass)
        for (size_t gi = 0; gi < pl->groups.size(); ++gi) {
            const LaunchGroup& g = pl->groups[gi];
            const bool light = light_first && g.count <= 512;
            if (light != (pass == 0)) continue;
            const int si = stream_of[gi];
            if (!used[si]) { HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux[si], ctx->ev_fork, 0)); used[si] = true; }
            ClDeviceBatch dev = pl->dev;
            dev.ticks = pl->d_ticks.p ? pl->d_ticks.p + 2 * gi : nullptr;
            dev.tick_pass = pl->tick_pass;
            if (pl->event_pass) {
                LaunchGroup& ge = pl->groups[gi];
                if (!ge.ev0) { HIP_TRY(ctx, hipEventCreate(&ge.ev0)); HIP_TRY(ctx, hipEventCreate(&ge.ev1)); }
                HIP_TRY(ctx, hipEventRecord(ge.ev0, ctx->aux[si]));
            }
            HIP_TRY(ctx, launch_group(g, pl, dev, ctx->aux[si]));
            if (pl->event_pass) { HIP_TRY(ctx, hipEventRecord(pl->groups[gi].ev1, ctx->aux[si])); pl->groups[gi].evented = true; }
        }
    const bool eager_join = [] { const char* e = getenv("CL_STITCH_JOIN"); return e && e[0] == 'e'; }();   // CL_STITCH_JOIN=eager: every pass joins (up to the first half of round 5); read per pass: bench.py times both
    for (int si = 0; si < kNumAuxStreams; ++si)
        if (used[si]) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join[si], ctx->aux[si]));
            ctx->stitch_join_pending |= 1u << si;
        }
    pl->stop_pending = true;
    if (eager_join || join_now) return plan_mark_stop(ctx, pl);
    return CL_OK;
}

int cl_stitch_plan_execute(cl_context* ctx, cl_stitch_plan* pl) {
    if (!ctx || !pl) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ++cl_fallbacks.stitch_plans;
    // A plan that is executed again (resident inputs, replayed passes) measures its launches ONCE — each alone on the device, the host's clock
    // round launch + wait — and deals them over the streams by those durations from then on: a launch lasts as long as its longest sweep at a
    // rate that depends on the subproblems' branching (0.25 .. 2.7 us per step), which the estimate of cl_stitch_plan_create cannot know.
    // Measured on the 10 x 1 Mbp batches as one plan: 3.03 ms per step with the estimate (the stream that got 1.40 + 0.70 + 0.79 ms sets the
    // pace), the sum of the sixteen launches is 12.8 ms over six streams.  CL_STITCH_CALIBRATE=0: the estimate only
    static const bool calibrate = [] { const char* e = getenv("CL_STITCH_CALIBRATE"); return !e || e[0] != '0'; }();
    if (calibrate && pl->executed && !pl->calibrated && pl->groups.size() > 2) {
        pl->calibrated = true;
        int rc = enqueue_groups(ctx, pl, true);
        if (rc) return rc;
        for (LaunchGroup& g : pl->groups) g.est_cost = (uint64_t)(g.host_ms * 1000.f) + 1;
        std::stable_sort(pl->groups.begin(), pl->groups.end(), [](const LaunchGroup& x, const LaunchGroup& y) { return x.est_cost > y.est_cost; });
        pl->tick_pass = 0;
    }
    // Second stage: a launch alone and the same launch among fifteen others are different things — in the timed step of 10 x 1 Mbp the chain kernel's launch of 221 long
    // pairs takes 0.4 ms alone and 1.7 ms in the step (its sixteen waves per workgroup wait for SIMDs they share), so the first stage put a 0.28 ms launch behind it on
    // its stream, and that stream ended the step 0.2 ms after every other one (scripts/dev/step_timeline.py).  So the pass after the first concurrent one deals
    // the launches out again by the durations their OWN clocks showed inside the previous pass (first workgroup's start to last workgroup's end, queueing for
    // compute units included — which is what a stream is busy for).  One wait for the stream + one small copy + one more graph capture, once per plan.  CL_STITCH_CALIBRATE=1: first stage only
    static const bool second_stage = [] { const char* e = getenv("CL_STITCH_CALIBRATE"); return !e || e[0] == '2'; }();
    // (how often: once — the third pass of a plan, which is still a warm-up pass of bench.py; CL_STITCH_RECAL=0..3 for measurements)
    static const int recal_env = [] { const char* e = getenv("CL_STITCH_RECAL"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : v > 3 ? 3 : v; }();
    if (pl->recalibrations_left < 0) pl->recalibrations_left = recal_env;
    if (calibrate && second_stage && pl->calibrated && pl->concurrent_passes >= 1 && pl->recalibrations_left > 0 && pl->groups.size() > 2 && pl->d_ticks.p) {
        --pl->recalibrations_left;
        pl->concurrent_passes = 0;
        { int rc = plan_mark_stop(ctx, pl); if (rc) return rc; }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<unsigned long long> got(2 * pl->groups.size());
        HIP_TRY(ctx, cl_copy_sync(ctx, got.data(), pl->d_ticks.p, got.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        bool all = true;
        for (size_t gi = 0; gi < pl->groups.size(); ++gi) all = all && tick_span_us(got[2 * gi], got[2 * gi + 1]) > 0;
        if (all) {
            for (size_t gi = 0; gi < pl->groups.size(); ++gi) pl->groups[gi].est_cost = (uint64_t)tick_span_us(got[2 * gi], got[2 * gi + 1]) + 1;
            pl->tick_pass = 0;   // (the words belong to the launches in their OLD order: start afresh)
            std::stable_sort(pl->groups.begin(), pl->groups.end(), [](const LaunchGroup& x, const LaunchGroup& y) { return x.est_cost > y.est_cost; });
            if (pl->graph_exec) { (void)hipGraphExecDestroy(pl->graph_exec); pl->graph_exec = nullptr; }
            if (pl->graph) { (void)hipGraphDestroy(pl->graph); pl->graph = nullptr; }
            pl->graph_tried = false;
        }
    }
    // The launch DAG (one kernel per group, all independent) is captured once into a hipGraph and replayed:
    // a dozen kernels start together instead of trickling out at the host's enqueue rate.
    if (!pl->graph_tried && !g_no_graph && !pl->groups.empty()) {
        pl->graph_tried = true;
        hipGraph_t graph = nullptr;
        // (nothing uncaptured may be waited for inside a capture: join what is out first; and the launch clocks are zeroed by a node of the graph — a replayed
        // pass carries the same pass number every time)
        { int rc = plan_mark_stop(ctx, pl); if (rc) return rc; }
        pl->tick_pass = 0;
        if (!pl->d_ticks.p) { int rc = pl->d_ticks.alloc(ctx, 2 * pl->groups.size()); if (rc) return rc; }   // (no allocation inside a capture)
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = enqueue_groups(ctx, pl, false, true);   // (a capture has to end with every forked stream joined)
            hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (rc == CL_OK && e == hipSuccess && graph &&
                hipGraphInstantiate(&pl->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                pl->graph = graph;
            } else {
                if (graph) (void)hipGraphDestroy(graph);
                pl->graph_exec = nullptr;
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
    }
    HIP_TRY(ctx, hipEventRecord(pl->ev_start, ctx->stream));
    if (pl->graph_exec) {
        HIP_TRY(ctx, hipGraphLaunch(pl->graph_exec, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(pl->ev_stop, ctx->stream));
        pl->stop_pending = false;
    } else {
        int rc = enqueue_groups(ctx, pl, false);   // (the end of the pass is marked when somebody joins: plan_mark_stop)
        if (rc) return rc;
    }
    pl->executed = true;
    if (pl->calibrated) ++pl->concurrent_passes;
    return CL_OK;
}

int cl_stitch_plan_execute_profiled(cl_context* ctx, cl_stitch_plan* pl) {
    if (!ctx || !pl) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc = plan_mark_stop(ctx, pl); if (rc) return rc; }
    HIP_TRY(ctx, hipEventRecord(pl->ev_start, ctx->stream));
    int rc = enqueue_groups(ctx, pl, true);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(pl->ev_stop, ctx->stream));
    pl->executed = true;
    pl->profiled = true;
    pl->tick_pass = 0;   // (the profiled launches left the highest pass number in the tick words: the next concurrent pass zeroes them)
    return CL_OK;
}

// A concurrent pass exactly as cl_stitch_plan_execute enqueues one without a graph — the same launches on the same streams, side by side, following the pass before in
// stream order — with a pair of HIP events round every launch ON ITS STREAM.  An event pair spans from the moment the stream reaches the launch (the end of whatever
// preceded it there) to the launch's completion, end-of-kernel cache write-back included: what the rocprofv3 kernel trace of a step shows as the dispatch's duration plus
// the stream's launch gap, i.e. an UPPER bound of the trace's duration where the kernel's own clock (first workgroup's start to last workgroup's end) is a lower bound.
// bench.py prices roofline.frac with these (round-5 verdict: the printed fraction must be the one profiles/ reproduces).
int cl_stitch_plan_execute_evented(cl_context* ctx, cl_stitch_plan* pl) {
    if (!ctx || !pl) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(pl->ev_start, ctx->stream));
    pl->event_pass = true;
    const int rc = enqueue_groups(ctx, pl, false);
    pl->event_pass = false;
    if (rc) return rc;
    pl->executed = true;
    return CL_OK;
}

int cl_stitch_plan_sync(cl_context* ctx, cl_stitch_plan* pl, float* ms_out) {
    if (!ctx || !pl) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc = plan_mark_stop(ctx, pl); if (rc) return rc; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ms_out) {
        *ms_out = 0.f;   // (from the last pass's enqueue on the stream to its end: with passes enqueued back to back, the time they all took from there)
        if (pl->executed) HIP_TRY(ctx, hipEventElapsedTime(ms_out, pl->ev_start, pl->ev_stop));
    }
    return CL_OK;
}

int cl_stitch_plan_stats(const cl_stitch_plan* pl, cl_plan_stats* out) {
    if (!pl || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = pl->stats;
    return CL_OK;
}

int cl_stitch_plan_launch_count(const cl_stitch_plan* pl) { return pl ? (int)pl->groups.size() : 0; }

int cl_stitch_plan_launch_info(cl_context* ctx, const cl_stitch_plan* pl, int index, cl_launch_info* out) {
    if (!ctx || !pl || !out || index < 0 || index >= (int)pl->groups.size()) return CL_ERR_INVALID_ARGUMENT;
    const LaunchGroup& g = pl->groups[index];
    memset(out, 0, sizeof(*out));
    if (g.kind == CL_KIND_LINEAR && g.waves == 40) snprintf(out->kernel, sizeof(out->kernel), "popoa_linear_span_kernel");
    else if (g.kind == CL_KIND_LINEAR && g.waves == 2) snprintf(out->kernel, sizeof(out->kernel), "popoa_linear_quad_kernel");
    else if (g.kind == CL_KIND_LINEAR && g.waves == 5) snprintf(out->kernel, sizeof(out->kernel), "popoa_linear_duo_kernel");
    else if (g.kind == CL_KIND_LINEAR) snprintf(out->kernel, sizeof(out->kernel), "popoa_linear_kernel<%d>", g.waves);
    else if (g.kind == CL_KIND_LANE) snprintf(out->kernel, sizeof(out->kernel), g.block == 1 ? "popoa_lane_kernel<%d, wide>" : "popoa_lane_kernel<%d>", g.waves);
    else if (g.kind == CL_KIND_SYS) snprintf(out->kernel, sizeof(out->kernel), "popoa_sys_kernel<%d, %d>", g.npw, g.block);
    else if (g.kind == CL_KIND_STRIP) snprintf(out->kernel, sizeof(out->kernel), "popoa_strip_kernel<%d> x %d", g.npw, g.block);
    else snprintf(out->kernel, sizeof(out->kernel), "%s<%d, %d>", g.ring_bytes ? "popoa_ring_kernel" : "popoa_general_kernel", g.npw, g.block);
    out->n_problems = g.count;
    if ((g.kind == CL_KIND_LANE && g.block == 1) || (g.kind == CL_KIND_LINEAR && g.waves == 40)) {   // (a wide / spanning pair takes a workgroup per group of strips: count the pairs)
        out->n_problems = 0;
        for (uint32_t i = g.first; i < g.first + g.count; ++i) out->n_problems += i == g.first || pl->plist_host[i] != pl->plist_host[i - 1];
    }
    out->dp_cells = g.cells;
    out->dp_bytes = g.bytes;
    out->lds_bytes = g.kind == CL_KIND_LINEAR ? 0u : g.ring_bytes;
    if (g.kind == CL_KIND_LINEAR && (g.waves == 2 || g.waves == 5)) {   // (the quads' and duos' lists are padded with "no pair")
        out->n_problems = 0;
        for (uint32_t i = g.first; i < g.first + g.count; ++i) out->n_problems += pl->plist_host[i] != 0xFFFFFFFFu;
    }
    for (uint32_t i = g.first; i < g.first + g.count; ++i) {
        if (g.kind != CL_KIND_STRIP && pl->plist_host[i] == 0xFFFFFFFFu) continue;
        const ClProbDesc& d = pl->desc[g.kind == CL_KIND_STRIP ? pl->strips[pl->strip_list[i]].prob : pl->plist_host[i]];
        if (d.n1 + d.n2 > out->max_sweep) { out->max_sweep = d.n1 + d.n2; out->max_n1 = d.n1; out->max_n2 = d.n2; }
    }
    if (pl->profiled) out->last_ms = g.host_ms;
    if (g.evented && hipEventQuery(g.ev1) == hipSuccess) {
        float ems = 0.f;
        if (hipEventElapsedTime(&ems, g.ev0, g.ev1) == hipSuccess) out->event_ms = ems;
        else (void)hipGetLastError();
    }
    if (pl->d_ticks.p && pl->executed) {   // the launch's own clock in the last pass (the caller has waited for it: cl_stitch_plan_sync)
        unsigned long long got[2] = {0, 0};
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (cl_copy_sync(ctx, got, pl->d_ticks.p + 2 * index, sizeof(got), hipMemcpyDeviceToHost) == hipSuccess && tick_span_us(got[0], got[1]) > 0)
            out->in_pass_ms = (float)(tick_span_us(got[0], got[1]) * 1e-3);
    }
    return CL_OK;
}

int cl_stitch_plan_collect(cl_context* ctx, cl_stitch_plan* pl, cl_stitch_result* out) {
    if (!ctx || !pl || !out) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    if (!pl->executed && !pl->desc.empty()) { set_error(ctx, "plan was not executed"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rc = plan_mark_stop(ctx, pl); if (rc) return rc; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t npo = pl->desc.size();
    std::vector<uint32_t> len(npo), status(npo);
    std::vector<int32_t> score(npo);
    std::vector<uint2> pairs(pl->d_out_pairs.n);
    if (npo) {
        HIP_TRY(ctx, cl_copy_sync(ctx, len.data(), pl->d_out_len.p, npo * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, cl_copy_sync(ctx, status.data(), pl->d_out_status.p, npo * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, cl_copy_sync(ctx, score.data(), pl->d_out_score.p, npo * 4, hipMemcpyDeviceToHost));
        if (!pairs.empty()) HIP_TRY(ctx, cl_copy_sync(ctx, pairs.data(), pl->d_out_pairs.p, pairs.size() * sizeof(uint2), hipMemcpyDeviceToHost));
    }
    // a pair whose strips gave up waiting for one another (status 9: popoa_strip_kernel's waits are bounded, and the strips of a launch can be kept apart
    // by whatever else crowds the device) is run again by the anti-diagonal kernel, which waits for nobody — same planes, same output slots
    {
        std::vector<uint32_t> redo[4];
        for (size_t i = 0; i < npo; ++i)
            if (status[i] == 9 && (pl->desc[i].kind == CL_KIND_STRIP || pl->desc[i].kind == CL_KIND_LANE)) redo[pl->desc[i].npw].push_back((uint32_t)i);
        bool any = false;
        {   // ... and a chain pair whose groups gave up waiting for one another (popoa_linear_span_kernel) by the one-workgroup chain kernel: same codes, same hand-off rows
            std::vector<uint32_t> again;
            for (size_t i = 0; i < npo; ++i)
                if (status[i] == 9 && pl->desc[i].kind == CL_KIND_LINEAR) again.push_back((uint32_t)i);
            if (!again.empty()) {
                any = true;
                DevBuf<uint32_t> d_again;
                int rc = d_again.upload(ctx, again);
                if (rc) return rc;
                ClDeviceBatch dev = pl->dev;
                dev.ticks = nullptr;
                hipError_t e = cl_launch_popoa_linear(16, (uint32_t)again.size(), dev, d_again.p, pl->sparams, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                d_again.release();
                if (e != hipSuccess) { set_error(ctx, "re-running %zu chain pairs on one workgroup each failed: %s", again.size(), hipGetErrorString(e)); return CL_ERR_HIP; }
                pl->stats.n_strip_fallbacks += again.size();
                cl_fallbacks.strip_fallbacks += again.size();
            }
        }
        for (int npw = 1; npw <= 3; ++npw) {
            if (redo[npw].empty()) continue;
            any = true;
            DevBuf<uint32_t> d_redo;
            int rc = d_redo.upload(ctx, redo[npw]);
            if (rc) return rc;
            ClDeviceBatch dev = pl->dev;
            dev.ticks = nullptr;
            hipError_t e = cl_launch_popoa_general(npw, 1024, (uint32_t)redo[npw].size(), 0, dev, d_redo.p, pl->sparams, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            d_redo.release();
            if (e != hipSuccess) { set_error(ctx, "re-running %zu pairs on the anti-diagonal kernel failed: %s", redo[npw].size(), hipGetErrorString(e)); return CL_ERR_HIP; }
            pl->stats.n_strip_fallbacks += redo[npw].size();
            cl_fallbacks.strip_fallbacks += redo[npw].size();
        }
        if (any) {
            HIP_TRY(ctx, cl_copy_sync(ctx, len.data(), pl->d_out_len.p, npo * 4, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, cl_copy_sync(ctx, status.data(), pl->d_out_status.p, npo * 4, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, cl_copy_sync(ctx, score.data(), pl->d_out_score.p, npo * 4, hipMemcpyDeviceToHost));
            if (!pairs.empty()) HIP_TRY(ctx, cl_copy_sync(ctx, pairs.data(), pl->d_out_pairs.p, pairs.size() * sizeof(uint2), hipMemcpyDeviceToHost));
        }
    }
    const uint64_t n = pl->n_problems;
    uint64_t total = 0;
    for (size_t i = 0; i < npo; ++i) {
        if (status[i] != 0) {
            set_error(ctx, "device traceback failed on problem %llu (status %u): no source..sink connection or corrupt input",
                      (unsigned long long)pl->po_problem[i], status[i]);
            return CL_ERR_UNREACHABLE_SINK;
        }
        total += len[i];
    }
    for (uint32_t k : pl->pd_problem) total += pl->pd_path[k].size();
    for (uint32_t k : pl->host_problem) total += pl->host_aln[k].size();
    out->n_problems = n;
    out->aln_off = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
    out->pairs = (uint64_t*)malloc((total ? total : 1) * 2 * sizeof(uint64_t));
    out->score = (int64_t*)calloc(n ? n : 1, sizeof(int64_t));
    out->route = (uint8_t*)calloc(n ? n : 1, 1);
    out->num_pw = (uint8_t*)calloc(n ? n : 1, 1);
    if (!out->aln_off || !out->pairs || !out->score || !out->route || !out->num_pw) {
        cl_stitch_result_free(out);
        return CL_ERR_OUT_OF_MEMORY;
    }
    uint64_t cur = 0;
    for (uint64_t k = 0; k < n; ++k) {
        out->route[k] = pl->route[k];
        out->num_pw[k] = pl->num_pw[k];
        const uint64_t nb1 = pl->node_off[0][k], nb2 = pl->node_off[1][k];
        if (pl->po_index[k] >= 0) {
            const size_t i = (size_t)pl->po_index[k];
            const ClProbDesc& d = pl->desc[i];
            const uint32_t cap = d.n1 + d.n2;
            const uint2* src = pairs.data() + d.out_base + (cap - len[i]);
            const uint32_t* o1 = pl->order[0].data() + d.node_base[0];
            const uint32_t* o2 = pl->order[1].data() + d.node_base[1];
            for (uint32_t t = 0; t < len[i]; ++t) {
                // rank+1 -> local id -> (translate, src/alignment.cpp:26-39) caller id
                uint64_t a = CL_GAP, b = CL_GAP;
                if (src[t].x) { a = o1[src[t].x - 1]; if (pl->has_back[0]) a = pl->back[0][nb1 + a]; }
                if (src[t].y) { b = o2[src[t].y - 1]; if (pl->has_back[1]) b = pl->back[1][nb2 + b]; }
                out->pairs[2 * cur] = a;
                out->pairs[2 * cur + 1] = b;
                ++cur;
            }
            out->score[k] = score[i];
        } else if (pl->route[k] != CL_ROUTE_PURE_DELETION_1 && pl->route[k] != CL_ROUTE_PURE_DELETION_2) {
            for (const auto& pr : pl->host_aln[k]) {
                uint64_t a = pr.first, b = pr.second;
                if (a != CL_GAP && pl->has_back[0]) a = pl->back[0][nb1 + a];
                if (b != CL_GAP && pl->has_back[1]) b = pl->back[1][nb2 + b];
                out->pairs[2 * cur] = a;
                out->pairs[2 * cur + 1] = b;
                ++cur;
            }
            out->score[k] = 0;   // do_alignment does not ask the heuristics for a score (stitcher.hpp:304-345)
        } else {
            const std::vector<uint32_t>& path = pl->pd_path[k];
            const bool first = pl->route[k] == CL_ROUTE_PURE_DELETION_1;
            for (uint32_t v : path) {
                uint64_t id = v;
                if (first) { if (pl->has_back[0]) id = pl->back[0][nb1 + v]; }
                else if (pl->has_back[1]) id = pl->back[1][nb2 + v];
                out->pairs[2 * cur] = first ? id : CL_GAP;       // swap_graphs for the graph2 case (src/alignment.cpp:41-45)
                out->pairs[2 * cur + 1] = first ? CL_GAP : id;
                ++cur;
            }
            out->score[k] = pure_deletion_score(path.size(), pl->num_pw[k], pl->aparams);
        }
        out->aln_off[k + 1] = cur;
    }
    return CL_OK;
}

void cl_stitch_plan_destroy(cl_context* ctx, cl_stitch_plan* pl) {
    if (ctx) (void)hipSetDevice(ctx->device);
    plan_free(pl);
}

static int run_chunk(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* sp, const uint8_t* force, cl_stitch_result* out);

// A plan holds one workspace for its whole batch: full score planes for every graph x graph subproblem that is not swept from LDS, addressed
// with 32-bit word offsets.  A merge whose gaps add up to more than that (several graph x graph gaps near the 40 M-cell ceiling of
// min_wfa_size, src/parameters.cpp:79; configs[4]-sized merges) is run as consecutive chunks of subproblems, each a plan of its own on the
// same context, the results appended in order.  CL_STITCH_WORKSPACE_WORDS (default 3 * 2^30 int32 words = 12 GB) for tests.
static int run_whole(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* sp, const uint8_t* force,
                     cl_stitch_result* out) {
    const char* cap_env = getenv("CL_STITCH_WORKSPACE_WORDS");
    const uint64_t cap = cap_env && atoll(cap_env) > 0 ? (uint64_t)atoll(cap_env) : (3ull << 30);
    const uint64_t n = batch->n_problems;
    std::vector<uint64_t> cuts(1, 0);
    uint64_t acc = 0;
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t n1 = batch->side[0].node_off[k + 1] - batch->side[0].node_off[k], n2 = batch->side[1].node_off[k + 1] - batch->side[1].node_off[k];
        const uint64_t words = (n1 + 1) * (n2 + 1) * 7;   // the planes of a NumPW = 3 matrix: an upper bound whatever kernel takes it
        if (acc && acc + words > cap) { cuts.push_back(k); acc = 0; }
        acc += words;
    }
    cuts.push_back(n);
    if (cuts.size() == 2) return run_chunk(ctx, batch, sp, force, out);
    memset(out, 0, sizeof(*out));
    std::vector<cl_stitch_result> parts(cuts.size() - 1);
    int rc = CL_OK;
    size_t done = 0;
    for (; done + 1 < cuts.size() && !rc; ++done) {
        cl_stitch_batch sub = *batch;
        const uint64_t a = cuts[done];
        sub.n_problems = cuts[done + 1] - a;
        for (int s = 0; s < 2; ++s) { sub.side[s].node_off += a; sub.side[s].src_off += a; sub.side[s].snk_off += a; }
        if (sub.only_deletion_alns) sub.only_deletion_alns += a;
        rc = run_chunk(ctx, &sub, sp, force ? force + a : nullptr, &parts[done]);
    }
    if (!rc) {
        uint64_t total_pairs = 0;
        for (size_t i = 0; i < done; ++i) total_pairs += parts[i].aln_off[parts[i].n_problems];
        out->n_problems = n;
        out->aln_off = (uint64_t*)malloc((n + 1) * sizeof(uint64_t));
        out->pairs = (uint64_t*)malloc((total_pairs ? total_pairs : 1) * 2 * sizeof(uint64_t));
        out->score = (int64_t*)malloc((n ? n : 1) * sizeof(int64_t));
        out->route = (uint8_t*)malloc(n ? n : 1);
        out->num_pw = (uint8_t*)malloc(n ? n : 1);
        if (!out->aln_off || !out->pairs || !out->score || !out->route || !out->num_pw) { cl_stitch_result_free(out); rc = CL_ERR_OUT_OF_MEMORY; }
        else {
            uint64_t pair_at = 0;
            for (size_t i = 0; i < done; ++i) {
                const cl_stitch_result& p = parts[i];
                const uint64_t a = cuts[i];
                for (uint64_t k = 0; k < p.n_problems; ++k) out->aln_off[a + k] = pair_at + p.aln_off[k];
                memcpy(out->pairs + 2 * pair_at, p.pairs, p.aln_off[p.n_problems] * 2 * sizeof(uint64_t));
                memcpy(out->score + a, p.score, p.n_problems * sizeof(int64_t));
                memcpy(out->route + a, p.route, p.n_problems);
                memcpy(out->num_pw + a, p.num_pw, p.n_problems);
                pair_at += p.aln_off[p.n_problems];
            }
            out->aln_off[n] = pair_at;
        }
    }
    for (size_t i = 0; i < done; ++i) cl_stitch_result_free(&parts[i]);
    return rc;
}

static int run_chunk(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* sp, const uint8_t* force,
                     cl_stitch_result* out) {
    // CL_STITCH_TIMING=1: host phase times on stderr
    static const bool timing = getenv("CL_STITCH_TIMING") != nullptr;
    auto t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[cl_stitch]   %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count());
        t = std::chrono::steady_clock::now();
    };
    cl_stitch_plan* pl = nullptr;
    int rc = cl_stitch_plan_create(ctx, batch, sp, force, &pl);
    if (rc) return rc;
    pl->graph_tried = true;   // executed once: capturing and instantiating a hipGraph costs more than the launches it would save
    lap("plan (host + upload)");
    rc = cl_stitch_plan_execute(ctx, pl);
    if (!rc && timing) { (void)hipStreamSynchronize(ctx->stream); lap("execute (device)"); }
    if (!rc) rc = cl_stitch_plan_collect(ctx, pl, out);
    lap("collect");
    cl_stitch_plan_destroy(ctx, pl);
    lap("free");
    return rc;
}

// host-only: the alignment of ONE subproblem by the route do_alignment takes for it, when that route is a host algorithm
int cl_host_route_align(const cl_stitch_batch* batch, uint64_t k, const cl_stitch_params* sp, int* route_out, uint64_t** pairs_out,
                        uint64_t* n_pairs_out) {
    if (!batch || !sp || !pairs_out || !n_pairs_out || k >= batch->n_problems) return CL_ERR_INVALID_ARGUMENT;
    *pairs_out = nullptr;
    *n_pairs_out = 0;
    const GraphView g1 = view(batch->side[0], k), g2 = view(batch->side[1], k);
    const int npw = choose_num_pw(g1.n, g2.n, sp->alignment_params);
    if (npw < 1 || npw > 3) return npw < 0 ? npw : CL_ERR_INVALID_ARGUMENT;
    const int route = route_problem(g1, g2, batch->only_deletion_alns && batch->only_deletion_alns[k], *sp);
    if (route_out) *route_out = route;
    if (route < 0) return route;
    if (route == CL_ROUTE_PO_POA) return CL_ERR_UNSUPPORTED_ROUTE;   // the device's
    HostAlignment aln;
    const int rc = host_route_alignment(route, g1, g2, npw, *sp, aln);
    if (rc) return rc;
    uint64_t* out = (uint64_t*)malloc((aln.size() ? aln.size() : 1) * 2 * sizeof(uint64_t));
    if (!out) return CL_ERR_OUT_OF_MEMORY;
    const uint64_t nb1 = batch->side[0].node_off[k], nb2 = batch->side[1].node_off[k];
    for (size_t i = 0; i < aln.size(); ++i) {   // translate, src/alignment.cpp:26-39
        uint64_t a = aln[i].first, b = aln[i].second;
        if (a != CL_GAP && batch->side[0].back_translation) a = batch->side[0].back_translation[nb1 + a];
        if (b != CL_GAP && batch->side[1].back_translation) b = batch->side[1].back_translation[nb2 + b];
        out[2 * i] = a;
        out[2 * i + 1] = b;
    }
    *pairs_out = out;
    *n_pairs_out = aln.size();
    return CL_OK;
}

int cl_stitch_rank_order(const cl_stitch_batch* batch, uint64_t k, int side, int mode, uint32_t* order_out, uint32_t* far_reads_out, uint32_t* longest_read_out) {
    if (!batch || k >= batch->n_problems || side < 0 || side > 1 || mode < 0 || mode > 2 || !order_out) return CL_ERR_INVALID_ARGUMENT;
    const GraphView g = view(batch->side[side], k);
    NextLists nx;
    nx.build(g);
    std::vector<uint32_t> order, st, indeg, rank(g.n);
    if (!topological_order(g, nx, order, st, indeg)) return CL_ERR_CYCLIC_GRAPH;
    for (uint32_t r = 0; r < g.n; ++r) rank[order[r]] = r;
    choose_rank_order(g, mode, order, rank, st, indeg);
    uint32_t far = 0, longest = 0;
    for (uint64_t v = 0; v < g.n; ++v) {
        if (order[rank[v]] != v) return CL_ERR_INVALID_ARGUMENT;   // (cannot happen: order and rank are inverse permutations)
        for (uint64_t e = g.prev_off[v]; e < g.prev_off[v + 1]; ++e) {
            const uint32_t back = rank[v] - rank[g.prev_idx[e]];
            far += back > 4;
            longest = std::max(longest, back);
        }
    }
    for (uint64_t i = 0; i < g.n_src; ++i) { far += rank[g.src[i]] + 1 > 4; longest = std::max(longest, rank[g.src[i]] + 1); }
    std::copy(order.begin(), order.end(), order_out);
    if (far_reads_out) *far_reads_out = far;
    if (longest_read_out) *longest_read_out = longest;
    return CL_OK;
}

int cl_po_poa_batch(cl_context* ctx, const cl_stitch_batch* batch, const uint8_t* num_pw, const cl_align_params* params,
                    cl_stitch_result* out) {
    if (!ctx || !batch || !num_pw || !params || !out) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    cl_stitch_params sp;
    cl_stitch_params_default(&sp);
    sp.alignment_params = *params;
    return run_whole(ctx, batch, &sp, num_pw, out);
}

int cl_stitch_batch_align(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* params, cl_stitch_result* out) {
    if (!ctx || !batch || !params || !out) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    return run_whole(ctx, batch, params, nullptr, out);
}


struct cl_owned_batch {
    clhost::OwnedBatch b;
};

int cl_extract_stitch_batch(const cl_base_graph* g1, const cl_base_graph* g2, const cl_anchor_segments* sg, cl_owned_batch** out) {
    if (!g1 || !g2 || !sg || !out) { set_error(nullptr, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    for (const cl_base_graph* g : {g1, g2})
        if (g->n_nodes >= (1ull << 32) || g->src_id >= g->n_nodes || g->snk_id >= g->n_nodes) { set_error(nullptr, "bad graph"); return CL_ERR_INVALID_ARGUMENT; }
    cl_owned_batch* ob = new (std::nothrow) cl_owned_batch();
    if (!ob) return CL_ERR_OUT_OF_MEMORY;
    int rc = clhost::extract_stitch_batch(*g1, *g2, *sg, ob->b, cl_shared_table(g1), cl_shared_table(g2));
    if (rc) { set_error(nullptr, rc == CL_ERR_CYCLIC_GRAPH ? "merge graph is not acyclic" : "empty anchor segment"); delete ob; return rc; }
    *out = ob;
    return CL_OK;
}

const cl_stitch_batch* cl_owned_batch_view(cl_owned_batch* b) { return b ? b->b.view() : nullptr; }
void cl_owned_batch_free(cl_owned_batch* b) { delete b; }

void cl_alignment_free(cl_alignment* a) {
    if (!a) return;
    free(a->pairs);
    a->pairs = nullptr;
    a->n_pairs = 0;
}

int cl_stitch(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_anchor_segments* sg,
              const cl_stitch_params* params, cl_alignment* out) {
    cl_bind_device(ctx);
    if (!ctx || !g1 || !g2 || !sg || !params || !out) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    out->n_pairs = 0;
    out->pairs = nullptr;
    static const bool timing = getenv("CL_STITCH_TIMING") != nullptr;
    auto t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[cl_stitch] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count());
        t = std::chrono::steady_clock::now();
    };
    cl_owned_batch* ob = nullptr;
    int rc = cl_extract_stitch_batch(g1, g2, sg, &ob);
    if (rc) { if (ctx) ctx->error = g_error; return rc; }
    lap("extraction");
    cl_stitch_result res;
    const cl_stitch_batch* bv = cl_owned_batch_view(ob);
    bool hooked = false;
    if (ctx->stitch_hook) {   // several devices take the merge's subproblems (cl_context_set_stitch_hook)
        uint64_t cells = 0;
        for (uint64_t q = 0; q < bv->n_problems; ++q) {
            const uint64_t a = bv->side[0].node_off[q + 1] - bv->side[0].node_off[q], b = bv->side[1].node_off[q + 1] - bv->side[1].node_off[q];
            if (a && b) cells += (a + 1) * (b + 1);
        }
        hooked = cells >= ctx->stitch_hook_min_cells;
    }
    if (hooked) {
        memset(&res, 0, sizeof(res));
        rc = ctx->stitch_hook(ctx->stitch_hook_user, ctx, bv, params, &res);
        ++ctx->stitch_hook_calls;
        if (!rc && res.n_problems != bv->n_problems) { cl_stitch_result_free(&res); set_error(ctx, "the stitch hook returned %llu results for %llu subproblems", (unsigned long long)res.n_problems, (unsigned long long)bv->n_problems); rc = CL_ERR_INVALID_ARGUMENT; }
        else if (rc) set_error(ctx, "the stitch hook failed (%d)", rc);
    } else rc = cl_stitch_batch_align(ctx, bv, params, &res);
    lap("batch align");
    cl_owned_batch_free(ob);
    if (rc) return rc;
    // stitcher.hpp:157-203: P0 A0 P1 A1 ... every subproblem but the first is preceded by one copied anchor
    uint64_t n_anchor_pairs = sg->n_segments ? sg->walk_off[sg->seg_off[sg->n_segments]] - sg->walk_off[sg->seg_off[0]] : 0;
    uint64_t total = res.aln_off[res.n_problems] + n_anchor_pairs;
    out->pairs = (uint64_t*)malloc((total ? total : 1) * 2 * sizeof(uint64_t));
    if (!out->pairs) { cl_stitch_result_free(&res); return CL_ERR_OUT_OF_MEMORY; }
    uint64_t cur = 0, k = 0;
    auto copy_problem = [&](uint64_t p) {
        uint64_t n = res.aln_off[p + 1] - res.aln_off[p];
        memcpy(out->pairs + 2 * cur, res.pairs + 2 * res.aln_off[p], n * 2 * sizeof(uint64_t));
        cur += n;
    };
    copy_problem(k++);
    for (uint64_t s = 0; s < sg->n_segments; ++s)
        for (uint64_t a = sg->seg_off[s]; a < sg->seg_off[s + 1]; ++a) {
            if (a != sg->seg_off[s]) copy_problem(k++);
            for (uint64_t w = sg->walk_off[a]; w < sg->walk_off[a + 1]; ++w) {
                out->pairs[2 * cur] = sg->walk1[w];
                out->pairs[2 * cur + 1] = sg->walk2[w];
                ++cur;
            }
            if (a + 1 == sg->seg_off[s + 1]) copy_problem(k++);
        }
    out->n_pairs = cur;
    cl_stitch_result_free(&res);
    lap("interleave anchors");
    return CL_OK;
}

// Stitcher::internal_stitch (stitcher.hpp:209-234): a chain of anchors between two places of ONE graph (a tandem duplication found by the
// cyclisation rounds, src/core.cpp:268-269).  The gaps between consecutive anchors are extracted from the graph against itself and aligned
// like any other gap (subalign with only_deletion_alns = false); the output order is the reference's: anchor 0, then for every later anchor
// its own pairs FOLLOWED by the alignment of the gap in front of it (:216-231).
int cl_internal_stitch(cl_context* ctx, const cl_base_graph* g, uint64_t n_anchors, const uint64_t* walk_off, const uint32_t* walk1,
                       const uint32_t* walk2, const cl_stitch_params* params, cl_alignment* out) {
    cl_bind_device(ctx);
    if (!ctx || !g || !params || !out || (n_anchors && (!walk_off || !walk1 || !walk2))) { set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    out->n_pairs = 0;
    out->pairs = nullptr;
    if (g->n_nodes >= (1ull << 32) || g->src_id >= g->n_nodes || g->snk_id >= g->n_nodes) { set_error(ctx, "bad graph"); return CL_ERR_INVALID_ARGUMENT; }
    cl_stitch_result res;
    memset(&res, 0, sizeof(res));
    if (n_anchors > 1) {
        clhost::PathMergeTable own;
        const clhost::PathMergeTable* pm = cl_shared_table(g);
        if (!pm) { if (!own.build(*g)) { set_error(ctx, "graph is not acyclic"); return CL_ERR_CYCLIC_GRAPH; } pm = &own; }
        const uint64_t seg_off[2] = {0, n_anchors};
        const cl_anchor_segments sg{1, seg_off, walk_off, walk1, walk2};
        for (uint64_t a = 0; a < n_anchors; ++a) if (walk_off[a + 1] <= walk_off[a]) { set_error(ctx, "empty anchor"); return CL_ERR_INVALID_ARGUMENT; }
        cl_owned_batch ob;
        int rc = clhost::extract_stitch_batch(*g, *g, sg, ob.b, pm, pm, false);
        if (rc) { set_error(ctx, "extraction failed"); return rc; }
        if ((rc = cl_stitch_batch_align(ctx, ob.b.view(), params, &res))) return rc;
        if (res.n_problems != n_anchors - 1) { cl_stitch_result_free(&res); set_error(ctx, "internal_stitch: gap count"); return CL_ERR_INVALID_ARGUMENT; }
    }
    const uint64_t total = (n_anchors ? walk_off[n_anchors] - walk_off[0] : 0) + (res.aln_off ? res.aln_off[res.n_problems] : 0);
    out->pairs = (uint64_t*)malloc((total ? total : 1) * 2 * sizeof(uint64_t));
    if (!out->pairs) { cl_stitch_result_free(&res); return CL_ERR_OUT_OF_MEMORY; }
    uint64_t cur = 0;
    for (uint64_t a = 0; a < n_anchors; ++a) {
        for (uint64_t w = walk_off[a]; w < walk_off[a + 1]; ++w) { out->pairs[2 * cur] = walk1[w]; out->pairs[2 * cur + 1] = walk2[w]; ++cur; }
        if (a) {
            const uint64_t n = res.aln_off[a] - res.aln_off[a - 1];
            memcpy(out->pairs + 2 * cur, res.pairs + 2 * res.aln_off[a - 1], n * 2 * sizeof(uint64_t));
            cur += n;
        }
    }
    out->n_pairs = cur;
    cl_stitch_result_free(&res);
    return CL_OK;
}

}  // extern "C"
