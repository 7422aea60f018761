// cl_internal.hpp — shared by the translation units of libcentrolign_amd.so (not part of the public ABI)
#ifndef CL_INTERNAL_HPP
#define CL_INTERNAL_HPP

#include <thread>
#include <functional>
#include <sys/mman.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/centrolign_amd.h"

constexpr int kNumAuxStreams = 12;

struct cl_context {
    int device = 0;
    bool poisoned = false;   // a merge-group wait expired on this context: its serial stream is stuck behind a wait nobody will satisfy.  Calls fail at once, releases do not wait for it
    hipStream_t stream = nullptr;
    hipStream_t aux[kNumAuxStreams] = {};   // the first n_aux are streams of their own, the rest aliases of them (aux[i] = aux[i % n_aux])
    int n_aux = kNumAuxStreams;
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[kNumAuxStreams] = {};
    int far_last_choice = 0;            // what the last affine chaining DP of this context with a far pass decided at its checkpoint: 1 branch-and-bound, 2 all-pairs sweep (0: none yet)
    uint32_t stitch_join_pending = 0;   // aux streams whose last stitch launches the context's stream has not been made to wait for yet (cl_stitch_join, cl_api.cpp)
    std::string error;
    std::string name;
    // page-locked host staging area, grown on demand and kept for the context's lifetime (cl_pinned): device-to-host copies into it run
    // at the link's rate, copies into pageable memory at a tenth of it, and locking pages is too slow to do per call
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    void* pinned_map = nullptr;      // the mapping `pinned` lies in (cl_pinned)
    size_t pinned_map_bytes = 0;
    std::mutex pinned_mutex;   // cl_merge grows the area on a helper thread beside the match finding: every access goes through cl_pinned
    // device memory pool (cl_dev_alloc / cl_dev_free).  hipFree waits for the WHOLE device and hipMalloc takes a process-wide lock: with
    // several contexts at work (the worker threads of an MSA) every release in one of them stalled on the others' kernels.  Released
    // blocks are kept here and handed out again, largest-fit within 2x; they go back to the driver when the context is destroyed or the
    // pool holds more than kPoolCap bytes.
    // events the chaining DP records once per macro-block (walk done / far done / sealed): three rings, made once and reused — an event wait
    // refers to the record that precedes it, so a slot can be recorded again as soon as its waits have been enqueued (a few blocks later)
    hipEvent_t ev_ring[4][32] = {};
    // the merge group this context shares the far pass of its chaining DPs with (cl_peer_api.cpp); n <= 1: none
    struct Peers {
        uint32_t n = 0, me = 0;
        int* inbox = nullptr;          // [kPeerRing][kPeerSlotInts] what the others found, then [kPeerMaxMembers][kPeerRing] arrival words (one allocation, exported)
        uint32_t* flags = nullptr;
        int* peer_inbox[8] = {};       // the other members' inboxes and arrival words as this process sees them (index = member)
        uint32_t* peer_flags[8] = {};
        uint32_t epoch = 0;            // one per shared DP, the same on every member; arrival words hold epoch << 20 | macro-block + 1
        uint32_t last_shared_epoch = 0; // the shared DP before this one (its "done" words are what the next one's peer stores wait for)
        uint32_t epoch_mark = 0;       // the highest epoch this context's inbox has ever been used with: the arrival words are never reset, so a group's
                                       // epoch base must not lie below it (cl_context_peer_group refuses), and
        uint32_t test_mark = 0;        // the highest token of cl_context_peer_selftest (a token at or below it would find its words already there)
        uint64_t shared_dps = 0, shared_far_launches = 0, merged_blocks = 0, steals = 0;
        uint32_t steal_job = 0, steal_next = 0;   // cl_context_peer_steal without a group: a local counter
        std::vector<std::pair<std::string, void*>> opened;   // IPC handles this context has opened (kept until it is destroyed)
    } peers;
    // cl_context_set_stitch_hook: the subproblems of a merge aligned by several devices (cl_api.cpp, cl_stitch)
    int (*stitch_hook)(void*, cl_context*, const cl_stitch_batch*, const cl_stitch_params*, cl_stitch_result*) = nullptr;
    void* stitch_hook_user = nullptr;
    uint64_t stitch_hook_min_cells = 0, stitch_hook_calls = 0;
    std::mutex pool_mutex;
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_size;
    size_t pool_free_bytes = 0;
    size_t dev_live_bytes = 0, dev_peak_bytes = 0;   // device bytes handed out by cl_dev_alloc and not yet given back; the high-water mark (cl_context_memory)
};
constexpr size_t kPoolCap = 24ull << 30;

inline void cl_pool_trim(cl_context* ctx) {   // under pool_mutex
    for (auto& b : ctx->pool_free) { ctx->pool_size.erase(b.second); (void)hipFree(b.second); }
    ctx->pool_free.clear();
    ctx->pool_free_bytes = 0;
}

// The HIP runtime's "current device" is per thread, and the library is driven from threads it did not make (a host's pool, cl_msa's workers):
// every public entry point that takes a context binds the calling thread to the context's device first, and so does every allocation (kernels
// go to the context's streams, but hipMalloc, hipFuncSetAttribute and event creation follow the current device).
inline void cl_bind_device(const cl_context* ctx) { if (ctx) (void)hipSetDevice(ctx->device); }

inline hipError_t cl_dev_alloc(cl_context* ctx, size_t bytes, void** out) {
    cl_bind_device(ctx);
    static const bool no_pool = getenv("CL_NO_POOL") != nullptr;
    if (no_pool) return hipMalloc(out, bytes);
    bytes = (bytes + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    auto it = ctx->pool_free.lower_bound(bytes);
    // a cached block serves a request of at least half its size — small and medium blocks; from 256 MB on the slack is an eighth: a 34 GB request that
    // took a 60 GB block made a context "hold" 87 GB where its arrays needed 61 (50 x 100 kbp root, round 5), which is what the memory model is checked against
    const size_t slack = bytes >= ((size_t)256 << 20) ? bytes / 8 : bytes + (1u << 20);
    static const bool alloc_log = getenv("CL_ALLOC_LOG") != nullptr;   // measurements (scripts/memory_model.py): every request of 64 MB and more
    if (alloc_log && bytes >= ((size_t)64 << 20))
        fprintf(stderr, "[cl_dev_alloc] %.1f MB asked, %s (live before: %.1f MB)\n", bytes / 1048576.0,
                it != ctx->pool_free.end() && it->first <= bytes + slack ? "a cached block serves it" : "fresh hipMalloc", ctx->dev_live_bytes / 1048576.0);
    if (it != ctx->pool_free.end() && it->first <= bytes + slack) {
        *out = it->second;
        ctx->pool_free_bytes -= it->first;
        ctx->dev_live_bytes += it->first;
        ctx->dev_peak_bytes = std::max(ctx->dev_peak_bytes, ctx->dev_live_bytes);
        ctx->pool_free.erase(it);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        cl_pool_trim(ctx);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) {
        ctx->pool_size[*out] = bytes;
        ctx->dev_live_bytes += bytes;
        ctx->dev_peak_bytes = std::max(ctx->dev_peak_bytes, ctx->dev_live_bytes);
    }
    return e;
}

inline hipError_t cl_ring_event(cl_context* ctx, int kind, uint32_t k, hipEvent_t* out) {
    hipEvent_t& e = ctx->ev_ring[kind][k & 31u];
    if (!e) { hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (rc != hipSuccess) return rc; }
    *out = e;
    return hipSuccess;
}

// cl_fallback_counters (process-wide, cl_api.cpp)
struct ClFallbackCounters { std::atomic<uint64_t> strip_fallbacks{0}, walk_stalls{0}, chain_dps{0}, stitch_plans{0}, strip_pairs{0}, bond_trims_past_the_end{0}; };
extern ClFallbackCounters cl_fallbacks;

void cl_peers_release(cl_context* ctx);         // cl_peer_api.cpp
struct cl_polish_params;
int cl_polish_cyclized_graph_workers(cl_context* const* ctxs, unsigned n_ctx, const cl_base_graph* graph, const char* const* path_names, const char* newick,
                                     const char* const* sequence_names, uint64_t n_sequences, const cl_merge_params* mp, const cl_polish_params* pp,
                                     cl_owned_base_graph** out, uint64_t* n_regions_out);   // cl_polish_api.cpp
bool cl_context_live(const cl_context* ctx);   // cl_api.cpp: created and not yet destroyed

// every stream of the context has run dry: what cl_dev_free makes sure of before a block goes back to the pool.  A caller that releases dozens of
// blocks at once (the end of a chaining DP: ~40 of them, each of which used to wait for all seven streams again — most of a small DP's time,
// and a polishing step runs tens of thousands of small DPs) waits once and releases with quiesced = true.
int cl_stitch_join(cl_context* ctx);   // cl_api.cpp: the context's stream waits for the stitch launches still out on the auxiliary streams
inline void cl_ctx_quiesce(cl_context* ctx) {
    if (ctx->poisoned) return;   // (its streams never run dry)
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < ctx->n_aux; ++i) if (ctx->aux[i]) (void)hipStreamSynchronize(ctx->aux[i]);
}
inline void cl_dev_free(cl_context* ctx, void* p, bool quiesced = false) {
    if (!p) return;
    static const bool no_pool = getenv("CL_NO_POOL") != nullptr;
    if (no_pool || !ctx || !cl_context_live(ctx)) { (void)hipFree(p); return; }   // (a plan may outlive the context it was made on)
    // the block may be handed out again at once: wait for THIS context's streams (hipFree used to wait for the whole device)
    if (!quiesced) cl_ctx_quiesce(ctx);
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    auto it = ctx->pool_size.find(p);
    if (it == ctx->pool_size.end()) { (void)hipFree(p); return; }
    ctx->dev_live_bytes -= std::min(ctx->dev_live_bytes, it->second);
    if (ctx->pool_free_bytes + it->second > kPoolCap) { ctx->pool_size.erase(it); (void)hipFree(p); return; }
    ctx->pool_free.emplace(it->second, p);
    ctx->pool_free_bytes += it->second;
}

// at least `bytes` of page-locked host memory owned by the context (contents undefined; one user at a time); nullptr when it cannot be had.
// Serialised per context (cl_merge pre-pins on a helper thread), and capped over the whole process: page-locked memory is taken from
// every other process of the host, and an MSA may run up to sixteen worker contexts (CL_PINNED_CAP_GB, default 16; callers fall back
// to pageable copies when the cap is reached).
extern std::atomic<size_t> cl_pinned_total;   // cl_api.cpp
inline size_t cl_pinned_cap() {
    static const size_t cap = [] { const char* e = getenv("CL_PINNED_CAP_GB"); const long v = e ? atol(e) : 16; return (size_t)(v < 0 ? 0 : v) << 30; }();
    return cap;
}
// The area is an anonymous mapping advised to use huge pages, touched and then registered with the runtime: locking 900 MB that way takes
// 36 ms on the MI355X host (touch 34 + hipHostRegister 2.4) against 170-190 ms for hipHostMalloc of the same size, releasing it 33 ms
// against 80-100 — and while pages are being locked every HIP call and every page fault of the process waits (scripts/dev/pin_bench.cpp).
// Device-to-host copies into it run at the same 54 GB/s.
inline void cl_pinned_release_locked(cl_context* ctx) {
    if (ctx->pinned) {
        (void)hipHostUnregister(ctx->pinned);
        (void)munmap(ctx->pinned_map, ctx->pinned_map_bytes);
        cl_pinned_total -= ctx->pinned_bytes;
    }
    ctx->pinned = nullptr;
    ctx->pinned_map = nullptr;
    ctx->pinned_bytes = ctx->pinned_map_bytes = 0;
}
inline void* cl_pinned(cl_context* ctx, size_t bytes) {
    std::lock_guard<std::mutex> lock(ctx->pinned_mutex);
    if (bytes <= ctx->pinned_bytes) return ctx->pinned;
    cl_pinned_release_locked(ctx);
    constexpr size_t kHuge = 2u << 20;
    const size_t want = (bytes + bytes / 8 + kHuge - 1) & ~(kHuge - 1);
    if (cl_pinned_total.fetch_add(want) + want > cl_pinned_cap()) { cl_pinned_total -= want; return nullptr; }
    void* map = mmap(nullptr, want + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (map == MAP_FAILED) { cl_pinned_total -= want; return nullptr; }
    char* area = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(map) + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
    (void)madvise(area, want, MADV_HUGEPAGE);
    for (size_t i = 0; i < want; i += 4096) area[i] = 0;
    if (hipHostRegister(area, want, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        (void)munmap(map, want + kHuge);
        cl_pinned_total -= want;
        return nullptr;
    }
    ctx->pinned = area;
    ctx->pinned_map = map;
    ctx->pinned_map_bytes = want + kHuge;
    ctx->pinned_bytes = want;
    return ctx->pinned;
}
// give the area back (after the largest merge of a run; the next user locks what it needs)
inline void cl_pinned_release(cl_context* ctx) {
    std::lock_guard<std::mutex> lock(ctx->pinned_mutex);
    cl_pinned_release_locked(ctx);
}

// defined in cl_api.cpp
void cl_set_error(cl_context* ctx, const char* fmt, ...);

#define HIP_TRY(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            cl_set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? CL_ERR_OUT_OF_MEMORY : CL_ERR_HIP;                 \
        }                                                                                         \
    } while (0)

// Blocking copy WITHOUT the legacy default stream: a hipMemcpy on the legacy stream fails ("operation would make the legacy
// stream depend on a capturing blocking stream") while ANOTHER thread of the process captures its stitch plan into a hipGraph, so
// every copy of this library goes through the context's own (non-blocking) stream and waits for it.
inline hipError_t cl_copy_sync(cl_context* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    cl_context* owner = nullptr;
    bool is_view = false;   // p lies inside another DevBuf's block (small chaining DPs pack all their arrays into one): nothing to release
    void view(T* at, size_t count) { release(); p = at; n = count; is_view = true; }
    int alloc(cl_context* ctx, size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        HIP_TRY(ctx, cl_dev_alloc(ctx, count * sizeof(T), (void**)&p));
        owner = ctx;
        return CL_OK;
    }
    template <class Vec>
    int upload(cl_context* ctx, const Vec& h) {
        int rc = alloc(ctx, h.size());
        if (rc) return rc;
        if (!h.empty()) HIP_TRY(ctx, cl_copy_sync(ctx, p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
        return CL_OK;
    }
    // the copy is only enqueued: `h` must stay as it is until the caller has synchronised the context's stream
    template <class Vec>
    int upload_async(cl_context* ctx, const Vec& h) {
        int rc = alloc(ctx, h.size());
        if (rc) return rc;
        if (!h.empty()) HIP_TRY(ctx, hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        return CL_OK;
    }
    void release(bool quiesced = false) {
        if (p && !is_view) cl_dev_free(owner, p, quiesced);
        p = nullptr;
        n = 0;
        is_view = false;
    }
};


// Host arrays of tens to hundreds of megabytes that are filled completely right after they are made (the record and query tables of a
// chaining DP: 1.2 GB at the root of a ten-sequence tree): std::vector would zero them first, single-threaded, and malloc hands such blocks
// out as fresh mappings whose every page faults on first touch — and takes the process's address-space lock, under which the other MSA
// workers' faults wait.  ClRawVec leaves new elements uninitialised and keeps large blocks in a per-thread cache for the next DP.
void* cl_big_alloc(size_t bytes);            // cl_api.cpp
void  cl_big_free(void* p) noexcept;
template <class T>
struct ClRawAlloc {
    using value_type = T;
    ClRawAlloc() = default;
    template <class U> ClRawAlloc(const ClRawAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(cl_big_alloc(n * sizeof(T))); }
    void deallocate(T* p, size_t) noexcept { cl_big_free(p); }
    template <class U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }   // default-, not value-initialised
    template <class U, class A0, class... A> void construct(U* p, A0&& a0, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...); }
    template <class U> bool operator==(const ClRawAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const ClRawAlloc<U>&) const { return false; }
};
template <class T> using ClRawVec = std::vector<T, ClRawAlloc<T>>;

// std::vector<match_set_t> owned by the library (cl_split_branching_matches, cl_find_matches); layout of cl_match_sets
#include <vector>
struct cl_owned_match_sets {
    std::vector<uint64_t> set_off1{0}, walk_off1{0}, set_off2{0}, walk_off2{0}, count1, count2, full_length;
    std::vector<uint32_t> nodes1, nodes2;
};

// a BaseGraph + SentinelTableau owned by the library (cl_fuse, cl_merge, cl_leaf_graph); layout of cl_base_graph
struct cl_owned_base_graph {
    std::vector<uint8_t> label;
    std::vector<uint64_t> next_off, prev_off, path_off;
    std::vector<uint32_t> next_idx, prev_idx, path_nodes;
    uint64_t src_id = 0, snk_id = 0;
};

// The guide tree as Execution keeps it (src/execution.cpp:12-92: pruned to the sequences, compacted, binarised), for Core::make_copy_expanded_tree
// (cl_polish_api.cpp); defined in cl_plan_api.cpp next to the parser
#include <string>
struct ClGuideTreeView {
    std::vector<std::vector<uint32_t>> kids;   // Tree::get_children, in order
    std::vector<std::string> label;
    uint32_t root = 0;
    std::vector<uint32_t> postorder;           // Tree::postorder
};
int cl_processed_guide_tree(const char* newick, const char* const* names, uint64_t n_names, ClGuideTreeView& out, std::string& error);
// purge_uncovered_nodes (src/modify_graph.cpp:89-163), cl_cyclize_api.cpp
bool cl_purge_uncovered(cl_owned_base_graph& g);

// defined in cl_anchor_api.cpp
int cl_find_matches_hooked(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_params* prm, cl_owned_match_sets** out,
                           cl_match_stats* stats, const std::function<void()>* after_device_half);   // cl_match_api.cpp
bool cl_split_is_identity(const cl_base_graph* g1, const cl_base_graph* g2, const cl_split_params* sp);
int cl_split_branching_matches_unless_identity(const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* ms, const cl_split_params* sp,
                                               cl_owned_match_sets** out);   // *out stays null when the split changes nothing

// The PathMerge tables of the two graphs of a merge are needed by the chaining, the partitioner's gap measurement and the stitcher's
// extraction (the reference builds them once in Core::align, core.hpp:186-193, and hands them down).  cl_core_align builds them once and
// registers them for the duration of the call, per thread; the stages look them up by graph pointer and build their own when called alone.
namespace clhost { class PathMergeTable; }
struct ClSharedTables { const cl_base_graph* g[2] = {nullptr, nullptr}; const clhost::PathMergeTable* x[2] = {nullptr, nullptr}; };
extern thread_local ClSharedTables cl_tls_tables;   // cl_align_api.cpp
inline const clhost::PathMergeTable* cl_shared_table(const cl_base_graph* g) {
    for (int i = 0; i < 2; ++i) if (g && cl_tls_tables.g[i] == g) return cl_tls_tables.x[i];
    return nullptr;
}

// the two PathMerge tables of a merge, built on a thread of their own (cl_align_api.cpp)
struct ClPathMergeTables {
    clhost::PathMergeTable *x1, *x2;
    bool ok1 = false, ok2 = false;
    std::thread builder;
    ClPathMergeTables();
    ~ClPathMergeTables();
    ClPathMergeTables(const ClPathMergeTables&) = delete;
    void start(const cl_base_graph* g1, const cl_base_graph* g2, bool chain_merge = false);
    void wait();
};
int cl_core_align_prepared(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* matches, const cl_core_align_params* ap,
                           cl_core_align_result* out, ClPathMergeTables* ready /* may be null */);

// host-side parallel loop over [0, n): f(begin, end) on up to 32 threads (the reference is single-threaded; the host glue around the device
// passes is not part of the compared arithmetic, every iteration writes its own outputs).  The threads come from ONE pool per process
// (cl_api.cpp), made on first use: creating and destroying threads per loop costs an mmap / munmap of every stack, and those take the
// process's address-space lock in write mode — with four MSA workers doing it dozens of times per merge, every page fault of every
// other worker waited (a leaf merge took 2.5 s next to three others, 1.1 s alone).
void cl_pool_run(unsigned n_tasks, const std::function<void(unsigned)>& task);   // task(0 .. n_tasks-1), task 0 on the caller; returns when all are done
unsigned cl_pool_width();                                                       // threads a loop may use (CL_HOST_THREADS, default min(cores, 32))
template <class F>
inline void cl_parallel_for(uint64_t n, F f, uint64_t grain = 32768) {
    uint64_t nt = std::min<uint64_t>(cl_pool_width(), (n + grain - 1) / grain);
    if (nt <= 1) { f((uint64_t)0, n); return; }
    const uint64_t chunk = (n + nt - 1) / nt;
    cl_pool_run((unsigned)nt, [&](unsigned t) { f(std::min(n, (uint64_t)t * chunk), std::min(n, ((uint64_t)t + 1) * chunk)); });
}

#endif
