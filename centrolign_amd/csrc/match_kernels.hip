// match_kernels.hip — suffix array + LCP array of the joined path text, the index PathESA builds with SA-IS and Kasai's
// algorithm (include/centrolign/path_esa.hpp:101-123,174-200).  Both of those are sequential pointer chases; here:
//
//   * suffix array by PREFIX DOUBLING: suffixes are ranked by their first 8 characters (one 64-bit key per position,
//     one radix sort), then by (rank[i], rank[i + h]) for h = 8, 16, 32, ... (one key-building kernel, one radix sort
//     (cl_radix.h: stable LSD passes over just the bytes a rank pair needs, one flag + scan + scatter to re-rank) until every rank is unique.  The
//     text ends in a unique smallest character (path_esa.hpp:113-117), so ranks past the end never decide a comparison.
//   * the rank array of EVERY round is kept (4 bytes x text length x ~log2(longest repeat) — nothing next to 288 GB), which
//     turns the LCP of two suffixes into a descent over the rounds: equal round-k ranks mean the next 8 * 2^k characters
//     agree.  One thread per suffix-array position, no sequential dependence (Kasai's loop carries its match length
//     from one text position to the next).
//
// Everything is HBM-streaming integer work: per round ~ (8 + 4) B x n of key/rank traffic on top of the radix sort's passes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "cl_internal.hpp"
#include "cl_radix.h"
#include "match_device.h"

namespace {

constexpr uint32_t kBlock = 256;

__global__ void first_keys_kernel(const uint8_t* __restrict__ text, uint32_t n, uint64_t* __restrict__ key, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint64_t k = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) k = (k << 8) | (i + j < n ? (uint64_t)text[i + j] : 0ull);
    key[i] = k;
    idx[i] = i;
}

// key of round h: (rank[i], rank[i + h]) packed into 2 * bits bits; ranks are 1-based, 0 = past the end
__global__ void pair_keys_kernel(const uint32_t* __restrict__ rank, uint32_t n, uint32_t h, uint32_t bits, uint64_t* __restrict__ key,
                                 uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t lo = (uint64_t)i + h < n ? rank[i + h] : 0u;
    key[i] = ((uint64_t)rank[i] << bits) | lo;
    idx[i] = i;
}

__global__ void head_flags_kernel(const uint64_t* __restrict__ sorted_key, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    flag[j] = (j == 0 || sorted_key[j] != sorted_key[j - 1]) ? 1u : 0u;
}

__global__ void scatter_rank_kernel(const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ dense, uint32_t n,
                                    uint32_t* __restrict__ rank) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    rank[sorted_idx[j]] = dense[j];
}

// lcp[p] = longest common prefix of the suffixes at sa[p-1] and sa[p] (lcp[0] = 0, path_esa.hpp:181-183); isa[sa[p]] = p
__global__ void lcp_kernel(const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ sa, const uint32_t* const* __restrict__ level,
                           int n_levels, uint32_t* __restrict__ lcp, uint32_t* __restrict__ isa) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t i = sa[p];
    isa[i] = p;
    if (p == 0) { lcp[0] = 0; return; }
    const uint32_t j = sa[p - 1];
    uint32_t len = 0;
    for (int k = n_levels - 1; k >= 0; --k) {
        const uint32_t a = i + len, b = j + len;
        if (a < n && b < n && level[k][a] == level[k][b]) len += 8u << k;
    }
    // the two suffixes differ within the next 8 characters (their round-0 ranks differ)
    while (i + len < n && j + len < n && text[i + len] == text[j + len]) ++len;
    lcp[p] = len;
}

uint32_t bits_for(uint32_t v) {   // bits needed to hold values 0 .. v
    uint32_t b = 1;
    while (b < 32 && (v >> b) != 0) ++b;
    return b;
}

}  // namespace

int cl_match_suffix_array(cl_context* ctx, const uint8_t* h_text, uint32_t n, uint32_t* h_sa, uint32_t* h_lcp, uint32_t* h_isa, ClSuffixStats* st) {
    if (st) *st = ClSuffixStats{};
    if (n == 0) return CL_OK;
    hipStream_t s = ctx->stream;
    DevBuf<uint8_t> text;
    DevBuf<uint64_t> key_in, key_out;
    DevBuf<uint32_t> idx_in, idx_out, flag, dense, lcp, isa;
    DevBuf<char> temp;
    std::vector<DevBuf<uint32_t>> level;
    DevBuf<const uint32_t*> level_ptr;
    struct Release {   // DevBuf has no destructor (the other translation units free explicitly)
        std::vector<DevBuf<uint32_t>>& lv;
        DevBuf<uint8_t>& a; DevBuf<uint64_t>& b; DevBuf<uint64_t>& c; DevBuf<uint32_t>& d; DevBuf<uint32_t>& e; DevBuf<uint32_t>& f;
        DevBuf<uint32_t>& g; DevBuf<uint32_t>& h; DevBuf<uint32_t>& i; DevBuf<char>& j; DevBuf<const uint32_t*>& k;
        ~Release() { for (auto& x : lv) x.release(); a.release(); b.release(); c.release(); d.release(); e.release(); f.release();
                     g.release(); h.release(); i.release(); j.release(); k.release(); }
    } release{level, text, key_in, key_out, idx_in, idx_out, flag, dense, lcp, isa, temp, level_ptr};

    int rc;
    const bool timing = getenv("CL_CHAIN_TIMING") != nullptr;
    auto t_host = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[cl_match_suffix_array] %-20s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count());
        t_host = std::chrono::steady_clock::now();
    };
    if ((rc = text.alloc(ctx, n)) || (rc = key_in.alloc(ctx, n)) || (rc = key_out.alloc(ctx, n)) || (rc = idx_in.alloc(ctx, n)) ||
        (rc = idx_out.alloc(ctx, n)) || (rc = flag.alloc(ctx, n)) || (rc = dense.alloc(ctx, n)) || (rc = lcp.alloc(ctx, n)) || (rc = isa.alloc(ctx, n)))
        return rc;
    lap("allocations");
    HIP_TRY(ctx, hipMemcpyAsync(text.p, h_text, n, hipMemcpyHostToDevice, s));
    const size_t sort_bytes = clradix::sort_temp_bytes<uint64_t>(n), scan_bytes = clradix::scan_temp_bytes(n);
    if ((rc = temp.alloc(ctx, sort_bytes > scan_bytes ? sort_bytes : scan_bytes))) return rc;
    size_t temp_bytes = temp.n;

    hipEvent_t e0, e1, e2;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventCreate(&e2));
    struct Events { hipEvent_t a, b, c; ~Events() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } events{e0, e1, e2};
    HIP_TRY(ctx, hipEventRecord(e0, s));

    const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
    const uint32_t bits = bits_for(n);
    uint32_t rounds = 0;
    for (uint32_t h = 0;; h = h ? h * 2 : 8) {
        if (h == 0) {
            hipLaunchKernelGGL(first_keys_kernel, grid, block, 0, s, text.p, n, key_in.p, idx_in.p);
            HIP_TRY(ctx, clradix::sort_pairs<uint64_t>(temp.p, temp_bytes, key_in.p, key_out.p, idx_in.p, idx_out.p, (size_t)n, 0u, 64u, s));
        } else {
            hipLaunchKernelGGL(pair_keys_kernel, grid, block, 0, s, level.back().p, n, h, bits, key_in.p, idx_in.p);
            HIP_TRY(ctx, clradix::sort_pairs<uint64_t>(temp.p, temp_bytes, key_in.p, key_out.p, idx_in.p, idx_out.p, (size_t)n, 0u, 2 * bits, s));
        }
        hipLaunchKernelGGL(head_flags_kernel, grid, block, 0, s, key_out.p, n, flag.p);
        HIP_TRY(ctx, clradix::inclusive_sum(temp.p, temp_bytes, flag.p, dense.p, (size_t)n, s));
        level.emplace_back();
        if ((rc = level.back().alloc(ctx, n))) return rc;
        hipLaunchKernelGGL(scatter_rank_kernel, grid, block, 0, s, idx_out.p, dense.p, n, level.back().p);
        HIP_TRY(ctx, hipGetLastError());
        uint32_t distinct = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&distinct, dense.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        ++rounds;
        if (distinct == n) break;
        if (h >= n) { cl_set_error(ctx, "suffix ranks did not separate: the text must end in a unique smallest character"); return CL_ERR_INVALID_ARGUMENT; }
    }
    HIP_TRY(ctx, hipEventRecord(e1, s));
    lap("doubling rounds");
    // idx_out now holds the suffix array (every key distinct)
    std::vector<const uint32_t*> ptrs;
    for (auto& l : level) ptrs.push_back(l.p);
    if ((rc = level_ptr.upload(ctx, ptrs))) return rc;
    hipLaunchKernelGGL(lcp_kernel, grid, block, 0, s, text.p, n, idx_out.p, level_ptr.p, (int)level.size(), lcp.p, isa.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(e2, s));
    if (timing) { HIP_TRY(ctx, hipStreamSynchronize(s)); lap("lcp"); }
    HIP_TRY(ctx, hipMemcpyAsync(h_sa, idx_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(h_lcp, lcp.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(h_isa, isa.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    lap("download");
    if (st) {
        st->rounds = rounds;
        (void)hipEventElapsedTime(&st->sort_ms, e0, e1);
        (void)hipEventElapsedTime(&st->lcp_ms, e1, e2);
    }
    return CL_OK;
}
