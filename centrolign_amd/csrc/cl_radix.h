// cl_radix.h — a stable LSD radix sort of (key, 32-bit value) pairs and an inclusive prefix sum, written for gfx950's 64-wide waves (round 6: in place of
// the device-wide library primitives of rounds 1-5; round-5 verdict item 6).  What they serve: the prefix-doubling rounds of the suffix array (match_kernels.hip: keys are
// (rank, rank) pairs of bounded width — only the bytes a round's keys occupy are sorted), the traceback's value index (chain_sort.hip) and the static orders of the
// branch-and-bound far pass (chain_far.hip).  Reference for what the sorts feed: include/centrolign/path_esa.hpp:174-205 (suffix array + LCP), anchorer.hpp:2196-2237.
//
// One pass per 8-bit digit, three launches per pass:
//   radix_hist_kernel     tile of 2 048 keys per workgroup -> table[digit][tile] (LDS atomics, one row-major store per digit)
//   radix_rowscan_kernel  one workgroup per digit: exclusive scan of its row of the table in place, the row's total on the side
//   radix_scatter_kernel  the same tiles again: a key's destination = (keys of smaller digits) + (same digit in earlier tiles) + (same digit earlier in this tile).
//                         The last term keeps the pass STABLE: a wave takes 512 consecutive keys in eight rounds of 64; inside a round the lanes that hold the same
//                         digit find one another with eight ballots (one per digit bit) and rank themselves by the population count of the peers below them; the
//                         count of earlier rounds sits in a per-wave LDS counter row (LDS operations of one wave complete in order: every peer reads the counter,
//                         then the highest peer writes it back), and the waves' rows are scanned once at the end.  No atomics on the ranking path, no sort inside LDS.
// Passes ping-pong between the caller's output arrays and a scratch pair inside `temp`, arranged so that the last pass lands in the output; the input is only read.
// HBM traffic per pass: keys read twice, values once, both written once — (2 K + 4) + (K + 4) bytes per pair; the launches are what a sort of a few million pairs
// costs (3 x ~6 us per pass), exactly as for the library sorts they replace.
#ifndef CL_RADIX_H
#define CL_RADIX_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clradix {
namespace {   // (a header of kernels included by several translation units: internal linkage)

constexpr uint32_t kThreads = 256, kItems = 8, kTile = kThreads * kItems, kWaves = kThreads / 64;

template <class K>
__global__ void __launch_bounds__(kThreads) radix_hist_kernel(const K* __restrict__ keys, uint32_t n, unsigned shift, unsigned mask, uint32_t* __restrict__ table, uint32_t n_tiles) {
    __shared__ uint32_t hist[256];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t base = tile * kTile;
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) {
        const uint32_t i = base + r * kThreads + tid;
        if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    table[(size_t)tid * n_tiles + tile] = hist[tid];
}

// row d of the table: exclusive scan over the tiles, in place; totals[d] = the row's sum
__global__ void __launch_bounds__(kThreads) radix_rowscan_kernel(uint32_t* __restrict__ table, uint32_t n_tiles, uint32_t* __restrict__ totals) {
    __shared__ uint32_t part[kThreads];
    const uint32_t tid = threadIdx.x;
    uint32_t* const row = table + (size_t)blockIdx.x * n_tiles;
    const uint32_t per = (n_tiles + kThreads - 1) / kThreads, lo = tid * per, hi = min(n_tiles, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += row[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < kThreads; step <<= 1) {   // inclusive Hillis-Steele over the 256 partial sums
        const uint32_t v = tid >= step ? part[tid - step] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = row[i]; row[i] = run; run += v; }
    if (tid == kThreads - 1) totals[blockIdx.x] = part[tid];
}

template <class K>
__global__ void __launch_bounds__(kThreads) radix_scatter_kernel(const K* __restrict__ keys_in, K* __restrict__ keys_out, const uint32_t* __restrict__ vals_in,
                                                                 uint32_t* __restrict__ vals_out, uint32_t n, unsigned shift, unsigned mask,
                                                                 const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals, uint32_t n_tiles) {
    __shared__ uint32_t cnt[kWaves][256];
    __shared__ uint32_t base[256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, tile = blockIdx.x;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) cnt[w][tid] = 0;
    // digit tid's first destination for this tile: every key of a smaller digit, then the same digit in earlier tiles
    base[tid] = totals[tid];
    __syncthreads();
    for (uint32_t step = 1; step < 256; step <<= 1) {
        const uint32_t v = tid >= step ? base[tid - step] : 0u;
        __syncthreads();
        base[tid] += v;
        __syncthreads();
    }
    const uint32_t mine = base[tid] - totals[tid] + table[(size_t)tid * n_tiles + tile];
    __syncthreads();
    base[tid] = mine;
    __syncthreads();
    K key[kItems];
    uint32_t rank[kItems];
    const uint32_t first = tile * kTile + wave * (kTile / kWaves);
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) {
        const uint32_t i = first + r * 64u + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : (K)0;
        const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (uint32_t bit = 0; bit < 8; ++bit) {
            const bool set = (d >> bit) & 1u;
            const unsigned long long m = __ballot(set);
            peers &= set ? m : ~m;
        }
        uint32_t old = 0;
        if (valid) old = cnt[wave][d];
        rank[r] = old + (uint32_t)__popcll(peers & below);
        if (valid && (peers >> lane) <= 1ull) cnt[wave][d] = old + (uint32_t)__popcll(peers);   // the highest peer writes the round's count back
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // counters of the waves -> exclusive offsets per digit
        uint32_t run = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) { const uint32_t v = cnt[w][tid]; cnt[w][tid] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) {
        const uint32_t i = first + r * 64u + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
            const uint32_t pos = base[d] + cnt[wave][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = vals_in[i];
        }
    }
}

inline uint32_t n_tiles_of(size_t n) { return (uint32_t)((n + kTile - 1) / kTile); }

// scratch: the alternate (key, value) pair of the ping-pong, the digit table [256][tiles], the row totals [256]
template <class K>
inline size_t sort_temp_bytes(size_t n) {
    const size_t al = 256;
    auto up = [&](size_t b) { return (b + al - 1) / al * al; };
    return up(n * sizeof(K)) + up(n * sizeof(uint32_t)) + up((size_t)256 * n_tiles_of(n) * sizeof(uint32_t)) + up(256 * sizeof(uint32_t)) + al;
}

// keys_out / vals_out = the pairs of keys_in / vals_in in ascending order of key bits [begin_bit, end_bit), equal keys in input order.  n < 2^32.
template <class K>
inline hipError_t sort_pairs(void* temp, size_t temp_bytes, const K* keys_in, K* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n, unsigned begin_bit,
                             unsigned end_bit, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 32) || end_bit <= begin_bit || end_bit > sizeof(K) * 8) return hipErrorInvalidValue;
    if (temp_bytes < sort_temp_bytes<K>(n)) return hipErrorInvalidValue;
    const size_t al = 256;
    auto up = [&](size_t b) { return (b + al - 1) / al * al; };
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(temp) + al - 1) / al * al);
    K* key_alt = reinterpret_cast<K*>(p); p += up(n * sizeof(K));
    uint32_t* val_alt = reinterpret_cast<uint32_t*>(p); p += up(n * sizeof(uint32_t));
    const uint32_t tiles = n_tiles_of(n);
    uint32_t* table = reinterpret_cast<uint32_t*>(p); p += up((size_t)256 * tiles * sizeof(uint32_t));
    uint32_t* totals = reinterpret_cast<uint32_t*>(p);
    const unsigned passes = (end_bit - begin_bit + 7) / 8;
    const K* src_k = keys_in;
    const uint32_t* src_v = vals_in;
    for (unsigned pass = 0; pass < passes; ++pass) {
        const unsigned shift = begin_bit + 8 * pass, width = end_bit - shift < 8 ? end_bit - shift : 8, mask = (1u << width) - 1u;
        const bool to_out = (passes - 1 - pass) % 2 == 0;   // the last pass lands in the caller's output
        K* dst_k = to_out ? keys_out : key_alt;
        uint32_t* dst_v = to_out ? vals_out : val_alt;
        hipLaunchKernelGGL((radix_hist_kernel<K>), dim3(tiles), dim3(kThreads), 0, stream, src_k, (uint32_t)n, shift, mask, table, tiles);
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(kThreads), 0, stream, table, tiles, totals);
        hipLaunchKernelGGL((radix_scatter_kernel<K>), dim3(tiles), dim3(kThreads), 0, stream, src_k, dst_k, src_v, dst_v, (uint32_t)n, shift, mask, table, totals, tiles);
        src_k = dst_k;
        src_v = dst_v;
    }
    return hipGetLastError();
}

// ---- inclusive prefix sum of uint32 (the dense ranks of the suffix rounds: a scan over head flags) ------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) scan_tiles_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ tile_sum) {
    __shared__ uint32_t part[kThreads];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * kTile + tid * kItems;
    uint32_t v[kItems], sum = 0;
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) { v[r] = base + r < n ? in[base + r] : 0u; sum += v[r]; v[r] = sum; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < kThreads; step <<= 1) {
        const uint32_t x = tid >= step ? part[tid - step] : 0u;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    const uint32_t before = part[tid] - sum;
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) if (base + r < n) out[base + r] = v[r] + before;
    if (tid == kThreads - 1) tile_sum[blockIdx.x] = part[tid];
}
// one workgroup: tile_sum[] -> exclusive prefix, in place
__global__ void __launch_bounds__(1024) scan_sums_kernel(uint32_t* __restrict__ tile_sum, uint32_t n_tiles) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, per = (n_tiles + 1023u) / 1024u, lo = tid * per, hi = min(n_tiles, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += tile_sum[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < 1024; step <<= 1) {
        const uint32_t x = tid >= step ? part[tid - step] : 0u;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = tile_sum[i]; tile_sum[i] = run; run += v; }
}
__global__ void __launch_bounds__(kThreads) scan_add_kernel(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ tile_before) {
    const uint32_t add = tile_before[blockIdx.x], base = blockIdx.x * kTile + threadIdx.x;
    if (!add) return;
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) { const uint32_t i = base + r * kThreads; if (i < n) out[i] += add; }
}

inline size_t scan_temp_bytes(size_t n) { return (size_t)n_tiles_of(n) * sizeof(uint32_t) + 512; }

inline hipError_t inclusive_sum(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 32) || temp_bytes < scan_temp_bytes(n)) return hipErrorInvalidValue;
    uint32_t* sums = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(temp) + 255) / 256 * 256);
    const uint32_t tiles = n_tiles_of(n);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(kThreads), 0, stream, in, out, (uint32_t)n, sums);
    if (tiles > 1) {
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, tiles);
        hipLaunchKernelGGL(scan_add_kernel, dim3(tiles), dim3(kThreads), 0, stream, out, (uint32_t)n, sums);
    }
    return hipGetLastError();
}

}  // namespace
}  // namespace clradix

#endif
