// cl_peer_api.cpp — one merge over several GPUs: the contexts of a MERGE GROUP (one context per process and device) all run the same merge
// on the same two graphs, and share the one part of it that is both heavy and free of order: the far pass of the affine chaining DP
// (chain_far.hip).  Every member walks, seals and runs the near pass itself — the walk is the serial spine and cannot be cut — but bounds only
// the queries of its own chain combinations (combination c belongs to member c mod n) and stores what it finds straight into the other members'
// INBOXES (device memory of theirs, mapped here through hipIpc*: peer stores over xGMI between devices, plain stores on one device).  The
// results are order-free maxima, so every member ends with the DP values, the chain and the fused graph of the single-context run.
//
// Ordering needs no host and no collective: after its far launch of macro-block k a member writes  epoch << 20 | k + 1  into its arrival word
// of slot k mod kPeerRing in every other inbox (hipStreamWriteValue32, stream order = after the kernel and its system-scope fence), and every
// member's serial stream waits for the arrival words of the others (hipStreamWaitValue32, >=) before it folds the slot into its running maxima
// (far_merge_kernel) and walks block k.  The words only ever grow (the epoch counts shared DPs and is handed in by the caller, the same on
// every member), so nothing is ever reset and a member may run ahead of one that has not entered the DP yet.  Measured on one MI355X, two
// processes: 14 µs per dependent round trip through such words, a 4-MB kernel included (scripts/dev/ipc_probe.cpp).
//
// Replaces nothing in the reference (it is single-threaded); the seam is still Core::align (core.hpp:181-252) through cl_merge.
#include <unistd.h>

#include <chrono>
#include <cstring>

#include "chain_device.h"
#include "cl_internal.hpp"

hipError_t cl_peer_store_word(uint32_t* where, uint32_t value, hipStream_t stream);   // chain_far.hip
hipError_t cl_peer_steal(unsigned long long* word, uint32_t job, uint32_t* out, hipStream_t stream);

namespace {
struct HandleBody {                 // what travels inside cl_peer_handle
    hipIpcMemHandle_t ipc;          // 64 bytes
    uint64_t pid;                   // owner: a member of the same process uses the pointer as it is
    uint64_t ptr;
    int32_t device;
    uint32_t magic;
};
static_assert(sizeof(HandleBody) <= sizeof(cl_peer_handle), "handle");
constexpr uint32_t kMagic = 0x434C5045u;
constexpr size_t kInboxInts = (size_t)kPeerRing * kPeerSlotInts;
constexpr size_t kFlagWords = kPeerFlagWords, kTestWords = kPeerTestWords, kTestInts = kPeerTestInts, kTailWords = kPeerTailWords;   // (chain_device.h)
static_assert(((kInboxInts + kPeerStealAt) * sizeof(uint32_t)) % 8 == 0, "the steal counter is a 64-bit word");
}

extern "C" {

int cl_context_peer_export(cl_context* ctx, cl_peer_handle* out) {
    if (!ctx || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto& P = ctx->peers;
    if (!P.inbox) {
        void* p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, kInboxInts * sizeof(int) + kTailWords * sizeof(uint32_t)));
        P.inbox = static_cast<int*>(p);
        P.flags = reinterpret_cast<uint32_t*>(P.inbox + kInboxInts);
        HIP_TRY(ctx, hipMemset(P.flags, 0, kTailWords * sizeof(uint32_t)));   // arrival words start below every epoch, the steal counter below every job
        HIP_TRY(ctx, hipDeviceSynchronize());
    }
    HandleBody h{};
    HIP_TRY(ctx, hipIpcGetMemHandle(&h.ipc, P.inbox));
    h.pid = (uint64_t)getpid();
    h.ptr = (uint64_t)(uintptr_t)P.inbox;
    h.device = ctx->device;
    h.magic = kMagic;
    memset(out, 0, sizeof(*out));
    memcpy(out, &h, sizeof(h));
    return CL_OK;
}

int cl_context_peer_group(cl_context* ctx, uint32_t n_members, uint32_t my_index, const cl_peer_handle* members, uint32_t epoch_base) {
    if (!ctx) return CL_ERR_INVALID_ARGUMENT;
    auto& P = ctx->peers;
    if (n_members <= 1) { P.n = 0; P.me = 0; return CL_OK; }
    if (!members || my_index >= n_members || n_members > kPeerMaxMembers || epoch_base >= (1u << 12)) { cl_set_error(ctx, "cl_context_peer_group: bad group"); return CL_ERR_INVALID_ARGUMENT; }
    if (!P.inbox) { cl_set_error(ctx, "cl_context_peer_group: cl_context_peer_export has not been called on this context"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (uint32_t m = 0; m < n_members; ++m) {
        P.peer_inbox[m] = nullptr; P.peer_flags[m] = nullptr;
        if (m == my_index) continue;
        HandleBody h;
        memcpy(&h, &members[m], sizeof(h));
        if (h.magic != kMagic) { cl_set_error(ctx, "cl_context_peer_group: member %u is not a handle of cl_context_peer_export", m); return CL_ERR_INVALID_ARGUMENT; }
        void* base = nullptr;
        if (h.pid == (uint64_t)getpid()) base = (void*)(uintptr_t)h.ptr;   // a context of this process: its pointer is ours too
        else {
            const std::string key(reinterpret_cast<const char*>(&h.ipc), sizeof(h.ipc));
            for (auto& o : P.opened) if (o.first == key) base = o.second;
            if (!base) {
                HIP_TRY(ctx, hipIpcOpenMemHandle(&base, h.ipc, hipIpcMemLazyEnablePeerAccess));
                P.opened.emplace_back(key, base);
            }
        }
        P.peer_inbox[m] = static_cast<int*>(base);
        P.peer_flags[m] = reinterpret_cast<uint32_t*>(static_cast<int*>(base) + kInboxInts);
    }
    // the arrival words of an inbox are never reset: a base below an epoch this context has already used would find its words at or above
    // epoch << 20 | k + 1 waiting, every hipStreamWaitValue32(>=) would pass at once and the previous run's slots would be folded in
    if (epoch_base < P.epoch_mark) {
        cl_set_error(ctx, "cl_context_peer_group: epoch base %u lies below an epoch this context has already used (%u): epochs must grow (cl_context_peer_stats reports the mark)", epoch_base, P.epoch_mark);
        return CL_ERR_INVALID_ARGUMENT;
    }
    P.n = n_members;
    P.me = my_index;
    P.epoch = epoch_base;
    P.last_shared_epoch = 0;   // (a new group: its members' "done" words refer to DPs shared from now on)
    return CL_OK;
}

// Every member of the current group calls this at the same time with the same token (> any earlier one): a kernel of this context stores the
// token into a word of every other member's memory, the stream then raises this member's test word there, waits for the others' test words
// here and reads back what their kernels stored.  CL_OK when everything arrived within timeout_ms — peer stores, stream memory operations and
// their ordering work between these devices; CL_ERR_HIP otherwise (the context's stream may then be stuck behind a wait: destroy the context).
int cl_context_peer_selftest(cl_context* ctx, uint32_t token, uint32_t timeout_ms) {
    if (!ctx) return CL_ERR_INVALID_ARGUMENT;
    auto& P = ctx->peers;
    if (P.n <= 1) return CL_OK;
    if (token <= P.test_mark) { cl_set_error(ctx, "cl_context_peer_selftest: token %u is not above the last one used (%u): the test would pass on stale words", token, P.test_mark); return CL_ERR_INVALID_ARGUMENT; }
    P.test_mark = token;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    for (uint32_t m = 0; m < P.n; ++m) {
        if (m == P.me) continue;
        uint32_t* their_ints = P.peer_flags[m] + kFlagWords + kTestWords;
        HIP_TRY(ctx, cl_peer_store_word(their_ints + 8 * P.me, token, s));                         // a kernel's store into the other device
        HIP_TRY(ctx, hipStreamWriteValue32(s, P.peer_flags[m] + kFlagWords + P.me, token, 0));   // then the arrival word
    }
    for (uint32_t m = 0; m < P.n; ++m)
        if (m != P.me) HIP_TRY(ctx, hipStreamWaitValue32(s, P.flags + kFlagWords + m, token, hipStreamWaitValueGte, 0xFFFFFFFFu));
    uint32_t* host = static_cast<uint32_t*>(cl_pinned(ctx, kTestInts * sizeof(uint32_t)));
    if (!host) { cl_set_error(ctx, "cl_context_peer_selftest: no page-locked memory"); return CL_ERR_OUT_OF_MEMORY; }
    HIP_TRY(ctx, hipMemcpyAsync(host, P.flags + kFlagWords + kTestWords, kTestInts * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { cl_set_error(ctx, "cl_context_peer_selftest: %s", hipGetErrorString(q)); return CL_ERR_HIP; }
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) {
            cl_set_error(ctx, "cl_context_peer_selftest: the other members' arrival words did not come within %u ms", timeout_ms);
            return CL_ERR_HIP;
        }
        usleep(200);
    }
    for (uint32_t m = 0; m < P.n; ++m)
        if (m != P.me && host[8 * m] != token) { cl_set_error(ctx, "cl_context_peer_selftest: member %u's store did not arrive (%u instead of %u)", m, host[8 * m], token); return CL_ERR_HIP; }
    return CL_OK;
}

// Work stealing between the members of the current group (north_star: the stitch subproblems of a merge shard across the GPUs "for work-stealing only").
// Every member holds the same list of chunks (the subproblems in LPT order, cut at a cell count: centrolign_amd/dist.py) and pulls chunk numbers from ONE
// counter until they run out — the counter is a 64-bit word in member 0's exported memory, job << 32 | chunks handed out, advanced by a one-thread kernel with
// system-scope atomics (peer atomics over xGMI between devices; plain L2 atomics between processes on one device).  `job` must be the same on every member
// and larger than any job the group has used (the first member to arrive takes the word over: nothing is reset, nobody waits for anybody).
// *chunk_out = the chunk this call won: the caller stops when it is >= its number of chunks.  A group of one (or no group) counts locally.
int cl_context_peer_steal(cl_context* ctx, uint32_t job, uint32_t* chunk_out) {
    if (!ctx || !chunk_out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    auto& P = ctx->peers;
    if (P.n <= 1) {   // nobody to share with
        if (job != P.steal_job) { P.steal_job = job; P.steal_next = 0; }
        *chunk_out = P.steal_next++;
        return CL_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t* tail = (P.me == 0 ? P.flags : P.peer_flags[0]) + kPeerStealAt;
    uint32_t* answer = P.flags + kPeerStealAt + 2;   // in this member's own memory
    HIP_TRY(ctx, cl_peer_steal(reinterpret_cast<unsigned long long*>(tail), job, answer, ctx->stream));
    uint32_t got = 0;
    HIP_TRY(ctx, cl_copy_sync(ctx, &got, answer, sizeof(got), hipMemcpyDeviceToHost));
    if (got == 0xFFFFFFFFu) { cl_set_error(ctx, "cl_context_peer_steal: job %u is older than the job the group's counter holds: job numbers must grow", job); return CL_ERR_INVALID_ARGUMENT; }
    *chunk_out = got;
    ++P.steals;
    return CL_OK;
}

int cl_context_peer_stats(const cl_context* ctx, cl_peer_stats* out) {
    if (!ctx || !out) return CL_ERR_INVALID_ARGUMENT;
    out->shared_dps = ctx->peers.shared_dps;
    out->shared_far_launches = ctx->peers.shared_far_launches;
    out->merged_blocks = ctx->peers.merged_blocks;
    out->epoch_mark = ctx->peers.epoch_mark;
    out->selftest_mark = ctx->peers.test_mark;
    out->steals = ctx->peers.steals;
    return CL_OK;
}

int cl_context_memory(cl_context* ctx, cl_memory_stats* out, int reset_peak) {
    if (!ctx || !out) return CL_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    {
        std::lock_guard<std::mutex> lock(ctx->pool_mutex);
        out->live_bytes = ctx->dev_live_bytes;
        out->peak_bytes = ctx->dev_peak_bytes;
        out->cached_bytes = ctx->pool_free_bytes;
        if (reset_peak) ctx->dev_peak_bytes = ctx->dev_live_bytes;
    }
    {
        std::lock_guard<std::mutex> lock(ctx->pinned_mutex);
        out->pinned_host_bytes = ctx->pinned_bytes;
    }
    size_t fr = 0, tot = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); cl_set_error(ctx, "hipMemGetInfo failed"); return CL_ERR_HIP; }
    out->device_free_bytes = fr;
    out->device_total_bytes = tot;
    return CL_OK;
}

}  // extern "C"

void cl_peers_release(cl_context* ctx) {
    auto& P = ctx->peers;
    for (auto& o : P.opened) (void)hipIpcCloseMemHandle(o.second);
    P.opened.clear();
    if (P.inbox) (void)hipFree(P.inbox);
    P.inbox = nullptr; P.flags = nullptr; P.n = 0;
}
