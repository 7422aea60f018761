// chain_kernels.hip — the Anchorer's sparse affine chaining DP (include/centrolign/anchorer.hpp:1812-2471) as a
// blocked all-pairs max on the GPU.
//
// The reference walks graph 1 in topological order and answers, per match pair m', 7 range-max queries per
// (chain1, chain2) combination over search trees keyed by diagonal shift and graph-2 offset
// (anchorer.hpp:2352-2416).  The VALUE such a query returns is a plain maximum over the predecessors m that
//   - lie on the chain pair (p1, p2),
//   - were inserted before the query:      index_on(e1(m), p1) <= predecessor_index(b1(m'), p1)
//   - end before m' starts in graph 2:     index_on(e2(m), p2) <  predecessor_index(b2(m'), p2) + 1
//   - and have shift == / < / > the query shift (gap-free tree / odd trees / even trees),
// of the value stored for m (dp(m), or dp(m) +- scale*extend_k*shift rounded to float, :2318-2342).  A maximum does
// not care in which order it is taken, so the GPU evaluates it by brute force: the match pairs are sorted by the
// topological position of their first graph-1 node and cut into blocks of kChainBlock; for block k
//   chain_inter_kernel : every earlier (final) match pair against every pair of the block, fully parallel,
//                        partial maxima merged with integer atomicMax on an order-preserving float encoding;
//   chain_intra_kernel : one workgroup walks the block in order; pair j is finalised (its dp value = the
//                        reference's update_dp maximum, same float/double arithmetic) and broadcast through LDS to
//                        the pairs after it.
// O(M^2/2) pair evaluations instead of O(M log^2 M) tree steps, but all of them independent: 922 k pairs
// (the 2 x 1 Mbp config) are 4.3e11 evaluations.  Which predecessor the reference's trees would report among
// EQUAL maxima is decided afterwards, only for the pairs on the optimal chain (chain_candidates_kernel + host).
//
// Compiled with -ffp-contract=off: the candidate values must round exactly like the reference's scalar code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chain_device.h"

namespace {

__device__ __forceinline__ int enc(float f) {  // order-preserving float -> int
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

// accumulate one predecessor record into the 7 running maxima of a query
__device__ __forceinline__ void accumulate(int (&acc)[7], uint32_t qt, uint32_t qoff, int32_t q, uint32_t ins_t, uint32_t off,
                                           int32_t sigma, const float (&v)[7]) {
    if (ins_t <= qt && off < qoff) {
        if (sigma == q) acc[0] = max(acc[0], enc(v[0]));
        else if (sigma < q) {  // odd trees: d1 > d2, value = dp + scale*extend_k*shift (anchorer.hpp:2328-2331, 2394-2403)
            acc[2] = max(acc[2], enc(v[2])); acc[4] = max(acc[4], enc(v[4])); acc[6] = max(acc[6], enc(v[6]));
        } else {               // even trees (:2332-2335, 2404-2412)
            acc[1] = max(acc[1], enc(v[1])); acc[3] = max(acc[3], enc(v[3])); acc[5] = max(acc[5], enc(v[5]));
        }
    }
}

// the reference's candidate values for one (chain1, chain2) combination and the running dp maximum
// (anchorer.hpp:2379-2412; update_dp keeps the first strictly greater value, match_bank.hpp:177)
__device__ __forceinline__ float apply_candidates(float best, const int (&acc)[7], float w, int32_t q, const ClChainParams& P) {
    const int none = enc(CL_CHAIN_NEG);
    if (acc[0] != none) best = fmaxf(best, dec(acc[0]) + w);
#pragma unroll
    for (int pw = 0; pw < 6; ++pw) {
        if (acc[1 + pw] == none) continue;
        const float stored = dec(acc[1 + pw]);
        const double go = P.gap_open[pw / 2], ge = P.gap_extend[pw / 2];
        const double pen = (pw % 2 == 1) ? P.scale * (go + ge * (double)q) : P.scale * (go - ge * (double)q);
        const float cand = (float)((double)(stored + w) - pen);
        best = fmaxf(best, cand);
    }
    return best;
}

}  // namespace

// acc index layout: [0] gap-free, [1 + pw] tree pw (pw even: shift > query, pw odd: shift < query)
__global__ void __launch_bounds__(256) chain_inter_kernel(ClChainDevice D, uint32_t block_first, uint32_t block_count) {
    const ClChainCombo cb = D.combos[blockIdx.z];
    const uint32_t n_prefix = cb.prefix[block_first / kChainBlock];
    const uint32_t tile0 = blockIdx.x * kChainTile;
    if (tile0 >= n_prefix) return;
    const uint32_t tile_end = min(tile0 + kChainTile, n_prefix);
    const uint32_t mi = blockIdx.y * 256 + threadIdx.x;
    const uint32_t s = block_first + mi;
    const bool active = mi < block_count;
    uint32_t qt = 0xFFFFFFFFu, qoff = 0;
    int32_t q = 0;
    if (active) { qt = cb.qt[s]; qoff = cb.qoff[s]; q = cb.q[s]; }
    const bool live = active && qt != 0xFFFFFFFFu;
    int acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = enc(CL_CHAIN_NEG);

    __shared__ uint32_t s_ins[256], s_off[256];
    __shared__ int32_t s_sig[256];
    __shared__ float s_val[7][256];
    for (uint32_t base = tile0; base < tile_end; base += 256) {
        const uint32_t r = base + threadIdx.x;
        __syncthreads();
        if (r < tile_end) {
            s_ins[threadIdx.x] = cb.ins_t[r];
            s_off[threadIdx.x] = cb.off[r];
            s_sig[threadIdx.x] = cb.sigma[r];
#pragma unroll
            for (int i = 0; i < 7; ++i) s_val[i][threadIdx.x] = cb.val[(size_t)i * cb.n_recs + r];
        }
        __syncthreads();
        const uint32_t cnt = min(256u, tile_end - base);
        if (live) {
            for (uint32_t j = 0; j < cnt; ++j) {
                float v[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) v[i] = s_val[i][j];
                accumulate(acc, qt, qoff, q, s_ins[j], s_off[j], s_sig[j], v);
            }
        }
    }
    if (live) {
        int* dst = cb.acc + (size_t)s * 7;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (acc[i] != enc(CL_CHAIN_NEG)) atomicMax(dst + i, acc[i]);
    }
}

// one workgroup, kChainBlock threads: thread i owns sorted match pair block_first + i
__global__ void __launch_bounds__(kChainBlock) chain_intra_kernel(ClChainDevice D, uint32_t block_first, uint32_t block_count) {
    const uint32_t i = threadIdx.x;
    const uint32_t s = block_first + i;
    const bool active = i < block_count;
    const ClChainCombo* combos = D.combos;
    // the first combination is kept in registers (the only one in a pairwise problem)
    const ClChainCombo c0 = combos[0];
    uint32_t qt0 = 0xFFFFFFFFu, qoff0 = 0;
    int32_t q0 = 0;
    int acc0[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc0[k] = enc(CL_CHAIN_NEG);
    if (active) {
        qt0 = c0.qt[s]; qoff0 = c0.qoff[s]; q0 = c0.q[s];
#pragma unroll
        for (int k = 0; k < 7; ++k) acc0[k] = c0.acc[(size_t)s * 7 + k];
    }
    __shared__ uint32_t s_n[2];
    __shared__ uint32_t s_combo[2][kChainMaxRecs], s_ins[2][kChainMaxRecs], s_off[2][kChainMaxRecs];
    __shared__ int32_t s_sig[2][kChainMaxRecs];
    __shared__ float s_val[2][kChainMaxRecs][7];

    for (uint32_t j = 0; j < block_count; ++j) {
        const uint32_t slot = j & 1u;
        if (i == j) {
            // finalise pair j: dp = max(own weight, every candidate) — anchorer.hpp:2041, 2379-2412
            const float w = D.weight[s];
            float best = w;
            best = apply_candidates(best, acc0, w, q0, D.params);
            for (uint32_t c = 1; c < D.n_combos; ++c) {
                const ClChainCombo cc = combos[c];
                if (cc.qt[s] == 0xFFFFFFFFu) continue;
                int a[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) a[k] = cc.acc[(size_t)s * 7 + k];
                best = apply_candidates(best, a, w, cc.q[s], D.params);
            }
            D.dp[s] = best;
            // publish its records: stored values of every tree it sits in (anchorer.hpp:2318-2342)
            const uint32_t r0 = D.rec_off[s], r1 = D.rec_off[s + 1];
            s_n[slot] = r1 - r0;
            for (uint32_t r = r0; r < r1; ++r) {
                const uint32_t c = D.rec_combo[r], pos = D.rec_pos[r];
                const ClChainCombo cc = combos[c];
                const int32_t sg = cc.sigma[pos];
                float v[7];
                v[0] = best;
#pragma unroll
                for (int pw = 0; pw < 6; ++pw) {
                    const double t = D.params.scale * D.params.gap_extend[pw / 2] * (double)sg;
                    v[1 + pw] = (pw % 2 == 1) ? (float)((double)best + t) : (float)((double)best - t);
                }
#pragma unroll
                for (int k = 0; k < 7; ++k) cc.val[(size_t)k * cc.n_recs + pos] = v[k];
                const uint32_t l = r - r0;
                if (l < kChainMaxRecs) {
                    s_combo[slot][l] = c; s_ins[slot][l] = cc.ins_t[pos]; s_off[slot][l] = cc.off[pos]; s_sig[slot][l] = sg;
#pragma unroll
                    for (int k = 0; k < 7; ++k) s_val[slot][l][k] = v[k];
                }
            }
        }
        __syncthreads();
        if (active && i > j) {
            const uint32_t n = s_n[slot];
            for (uint32_t l = 0; l < n; ++l) {
                uint32_t c, ins, off;
                int32_t sg;
                float v[7];
                if (l < kChainMaxRecs) {
                    c = s_combo[slot][l]; ins = s_ins[slot][l]; off = s_off[slot][l]; sg = s_sig[slot][l];
#pragma unroll
                    for (int k = 0; k < 7; ++k) v[k] = s_val[slot][l][k];
                } else {  // more records than the LDS slot holds: read them back from HBM (visible after the barrier)
                    const uint32_t r = D.rec_off[block_first + j] + l;
                    c = D.rec_combo[r];
                    const uint32_t pos = D.rec_pos[r];
                    const ClChainCombo cc = combos[c];
                    ins = cc.ins_t[pos]; off = cc.off[pos]; sg = cc.sigma[pos];
#pragma unroll
                    for (int k = 0; k < 7; ++k) v[k] = cc.val[(size_t)k * cc.n_recs + pos];
                }
                if (c == 0) {
                    if (qt0 != 0xFFFFFFFFu) accumulate(acc0, qt0, qoff0, q0, ins, off, sg, v);
                } else {
                    const ClChainCombo cc = combos[c];
                    const uint32_t qt = cc.qt[s];
                    if (qt == 0xFFFFFFFFu) continue;
                    int a[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) a[k] = cc.acc[(size_t)s * 7 + k];
                    accumulate(a, qt, cc.qoff[s], cc.q[s], ins, off, sg, v);
#pragma unroll
                    for (int k = 0; k < 7; ++k) cc.acc[(size_t)s * 7 + k] = a[k];
                }
            }
        }
    }
    // keep the final maxima: the traceback needs the value every query returned
    if (active) {
#pragma unroll
        for (int k = 0; k < 7; ++k) c0.acc[(size_t)s * 7 + k] = acc0[k];
    }
}

// For the pairs on the optimal chain: list the predecessors that attain the maximum of one query
// (combination, kind); the host then picks the one the reference's tree traversal would report.
__global__ void __launch_bounds__(256) chain_candidates_kernel(ClChainDevice D, const ClChainQuery* __restrict__ queries,
                                                               uint32_t n_queries, uint32_t* __restrict__ cand_count,
                                                               uint32_t* __restrict__ cand_list) {
    const uint32_t qi = blockIdx.y;
    if (qi >= n_queries) return;
    const ClChainQuery Q = queries[qi];
    const ClChainCombo cb = D.combos[Q.combo];
    const uint32_t qt = cb.qt[Q.s], qoff = cb.qoff[Q.s];
    const int32_t q = cb.q[Q.s];
    const int target = cb.acc[(size_t)Q.s * 7 + Q.kind];
    const float* val = cb.val + (size_t)Q.kind * cb.n_recs;
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < cb.n_recs; r += gridDim.x * 256) {
        if (cb.rec_s[r] >= Q.s) break;  // records are sorted by pair: later pairs cannot precede
        const int32_t sg = cb.sigma[r];
        const bool kind_ok = Q.kind == 0 ? sg == q : ((Q.kind - 1) % 2 == 1 ? sg < q : sg > q);
        if (kind_ok && cb.ins_t[r] <= qt && cb.off[r] < qoff && enc(val[r]) == target) {
            const uint32_t k = atomicAdd(cand_count + qi, 1u);
            if (k < kChainMaxCand) cand_list[(size_t)qi * kChainMaxCand + k] = r;
        }
    }
}

hipError_t cl_chain_launch_inter(const ClChainDevice& D, uint32_t block_first, uint32_t block_count, uint32_t max_prefix,
                                 hipStream_t stream) {
    if (max_prefix == 0) return hipSuccess;
    dim3 grid((max_prefix + kChainTile - 1) / kChainTile, (block_count + 255) / 256, D.n_combos);
    hipLaunchKernelGGL(chain_inter_kernel, grid, dim3(256), 0, stream, D, block_first, block_count);
    return hipGetLastError();
}

hipError_t cl_chain_launch_intra(const ClChainDevice& D, uint32_t block_first, uint32_t block_count, hipStream_t stream) {
    hipLaunchKernelGGL(chain_intra_kernel, dim3(1), dim3(kChainBlock), 0, stream, D, block_first, block_count);
    return hipGetLastError();
}

hipError_t cl_chain_launch_candidates(const ClChainDevice& D, const ClChainQuery* queries, uint32_t n_queries,
                                      uint32_t* cand_count, uint32_t* cand_list, hipStream_t stream) {
    if (!n_queries) return hipSuccess;
    hipLaunchKernelGGL(chain_candidates_kernel, dim3(64, n_queries), dim3(256), 0, stream, D, queries, n_queries, cand_count, cand_list);
    return hipGetLastError();
}
