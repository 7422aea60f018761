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

#include <cstring>

#include "chain_device.h"
#include "cl_internal.hpp"

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
constexpr size_t kFlagWords = (size_t)kPeerMaxMembers * kPeerRing;
}

extern "C" {

int cl_context_peer_export(cl_context* ctx, cl_peer_handle* out) {
    if (!ctx || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto& P = ctx->peers;
    if (!P.inbox) {
        void* p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, kInboxInts * sizeof(int) + kFlagWords * sizeof(uint32_t)));
        P.inbox = static_cast<int*>(p);
        P.flags = reinterpret_cast<uint32_t*>(P.inbox + kInboxInts);
        HIP_TRY(ctx, hipMemset(P.flags, 0, kFlagWords * sizeof(uint32_t)));   // arrival words start below every epoch
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
    P.n = n_members;
    P.me = my_index;
    P.epoch = epoch_base;
    return CL_OK;
}

int cl_context_peer_stats(const cl_context* ctx, cl_peer_stats* out) {
    if (!ctx || !out) return CL_ERR_INVALID_ARGUMENT;
    out->shared_dps = ctx->peers.shared_dps;
    out->shared_far_launches = ctx->peers.shared_far_launches;
    out->merged_blocks = ctx->peers.merged_blocks;
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
