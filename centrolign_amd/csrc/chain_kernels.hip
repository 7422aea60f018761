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
// EQUAL maxima is decided afterwards on the host, only for the pairs on the optimal chain (cl_chain_api.cpp).
//
// Compiled with -ffp-contract=off: the candidate values must round exactly like the reference's scalar code.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include "device_once.h"
#include "chain_device.h"

namespace {

__device__ __forceinline__ int enc(float f) {  // order-preserving float -> int; -0.0f and +0.0f map to the same key (the
    int b = __float_as_int(f);                  // reference's search-tree maxima and traceback '==' treat them as equal)
    if (b == (int)0x80000000) b = 0;
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

// accumulate one predecessor record into the 7 running maxima of a query.  v[] holds the record's stored values
// already in the order-preserving integer encoding; everything is branch-free selects + integer max.
__device__ __forceinline__ void accumulate(int (&acc)[7], uint32_t qt, uint32_t qoff, int32_t q, uint32_t ins_t, uint32_t off,
                                           int32_t sigma, const int (&v)[7]) {
    const int none = INT32_MIN;
    const bool ok = ins_t <= qt && off < qoff;
    const bool eq = ok && sigma == q, lt = ok && sigma < q, gt = ok && sigma > q;
    acc[0] = max(acc[0], eq ? v[0] : none);
    // odd trees: shift < query (anchorer.hpp:2328-2331, 2394-2403); even trees: shift > query (:2332-2335, 2404-2412)
    acc[2] = max(acc[2], lt ? v[2] : none); acc[4] = max(acc[4], lt ? v[4] : none); acc[6] = max(acc[6], lt ? v[6] : none);
    acc[1] = max(acc[1], gt ? v[1] : none); acc[3] = max(acc[3], gt ? v[3] : none); acc[5] = max(acc[5], gt ? v[5] : none);
}

// the six gap penalties of a query, scale*(open_k +- extend_k*query) (anchorer.hpp:2400, 2409), evaluated in double
__device__ __forceinline__ void query_penalties(double (&pen)[6], int32_t q, const ClChainParams& P) {
#pragma unroll
    for (int pw = 0; pw < 6; ++pw) {
        const double go = P.gap_open[pw / 2], ge = P.gap_extend[pw / 2];
        pen[pw] = (pw % 2 == 1) ? P.scale * (go + ge * (double)q) : P.scale * (go - ge * (double)q);
    }
}

// the reference's candidate values for one (chain1, chain2) combination and the running dp maximum
// (anchorer.hpp:2379-2412; update_dp keeps the first strictly greater value, match_bank.hpp:177)
__device__ __forceinline__ float apply_candidates(float best, const int (&acc)[7], float w, const double (&pen)[6]) {
    const int none = enc(CL_CHAIN_NEG);
    if (acc[0] != none) best = fmaxf(best, dec(acc[0]) + w);
#pragma unroll
    for (int pw = 0; pw < 6; ++pw) {
        if (acc[1 + pw] == none) continue;
        const float cand = (float)((double)(dec(acc[1 + pw]) + w) - pen[pw]);
        best = fmaxf(best, cand);
    }
    return best;
}

// workgroup barrier that orders LDS traffic only: the walk below communicates through LDS, its global stores (dp, stored
// values) are consumed by later kernels, so there is no need to drain them (__syncthreads waits for vmcnt) at every group
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace

// acc index layout: [0] gap-free, [1 + pw] tree pw (pw even: shift > query, pw odd: shift < query)
// predecessors = the records of blocks [src_block_lo, src_block_hi)
__global__ void __launch_bounds__(256) chain_inter_kernel(ClChainDevice D, uint32_t block_first, uint32_t block_count,
                                                          uint32_t src_block_lo, uint32_t src_block_hi, uint32_t tile_recs) {
    const ClChainCombo cb = D.combos[blockIdx.z];
    const uint32_t rec_lo = cb.prefix[src_block_lo] & ~D.lo_mask, rec_hi = cb.prefix[src_block_hi];
    const uint32_t tile0 = rec_lo + blockIdx.x * tile_recs;
    if (tile0 >= rec_hi) return;
    const uint32_t tile_end = min(tile0 + tile_recs, rec_hi);
    const uint32_t mi = blockIdx.y * 256 + threadIdx.x;
    const uint32_t s = block_first + mi;
    const bool active = mi < block_count;
    uint32_t qt = 0, qoff = 0;   // qoff == 0 can never be exceeded: an inactive lane accumulates nothing
    int32_t q = 0;
    if (active && cb.qt[s] != 0xFFFFFFFFu) { qt = cb.qt[s]; qoff = cb.qoff[s]; q = cb.q[s]; }
    int acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = INT32_MIN;

    // one record = 10 dwords: ins_t, off, sigma, 7 encoded values; padded to 12 so that it is read as three b128
    __shared__ __attribute__((aligned(16))) int s_rec[256][12];
    for (uint32_t base = tile0; base < tile_end; base += 256) {
        const uint32_t r = base + threadIdx.x;
        __syncthreads();
        if (r < tile_end) {
            s_rec[threadIdx.x][0] = (int)cb.ins_t[r];
            s_rec[threadIdx.x][1] = (int)cb.off[r];
            s_rec[threadIdx.x][2] = cb.sigma[r];
#pragma unroll
            for (int i = 0; i < 7; ++i) s_rec[threadIdx.x][3 + i] = enc(cb.val[(size_t)i * cb.n_recs + r]);
        }
        __syncthreads();
        const uint32_t cnt = min(256u, tile_end - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            const int4 a = *reinterpret_cast<const int4*>(&s_rec[j][0]);
            const int4 b = *reinterpret_cast<const int4*>(&s_rec[j][4]);
            const int4 c = *reinterpret_cast<const int4*>(&s_rec[j][8]);
            const int v[7] = {a.w, b.x, b.y, b.z, b.w, c.x, c.y};
            accumulate(acc, qt, qoff, q, (uint32_t)a.x, (uint32_t)a.y, a.z, v);
        }
    }
    if (qoff != 0) {
        int* dst = cb.acc + (size_t)s * 7;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (acc[i] > enc(CL_CHAIN_NEG)) atomicMax(dst + i, acc[i]);
    }
}

// sparse_chain_dp (anchorer.hpp:1511-1750) has one gap-free tree per chain pair and no shift condition: a record is
// (insertion index, offset, dp) and a query keeps ONE maximum.  Same tiling as chain_inter_kernel, a quarter of the work.
__global__ void __launch_bounds__(256) chain_inter_sparse_kernel(ClChainDevice D, uint32_t block_first, uint32_t block_count,
                                                                 uint32_t src_block_lo, uint32_t src_block_hi, uint32_t tile_recs) {
    const ClChainCombo cb = D.combos[blockIdx.z];
    const uint32_t rec_lo = cb.prefix[src_block_lo] & ~D.lo_mask, rec_hi = cb.prefix[src_block_hi];
    const uint32_t tile0 = rec_lo + blockIdx.x * tile_recs;
    if (tile0 >= rec_hi) return;
    const uint32_t tile_end = min(tile0 + tile_recs, rec_hi);
    const uint32_t mi = blockIdx.y * 256 + threadIdx.x;
    const uint32_t s = block_first + mi;
    const bool active = mi < block_count;
    uint32_t qt = 0, qoff = 0;   // qoff == 0 can never be exceeded: an inactive lane accumulates nothing
    if (active && cb.qt[s] != 0xFFFFFFFFu) { qt = cb.qt[s]; qoff = cb.qoff[s]; }
    int acc = INT32_MIN;
    __shared__ __attribute__((aligned(16))) int s_rec[256][4];
    for (uint32_t base = tile0; base < tile_end; base += 256) {
        const uint32_t r = base + threadIdx.x;
        __syncthreads();
        int4 mine = make_int4(-1, -1, INT32_MIN, 0);   // ins_t = 0xFFFFFFFF: never a predecessor
        if (r < tile_end) mine = make_int4((int)cb.ins_t[r], (int)cb.off[r], enc(cb.val[r]), 0);
        *reinterpret_cast<int4*>(&s_rec[threadIdx.x][0]) = mine;
        __syncthreads();
#pragma unroll 4
        for (uint32_t j = 0; j < 256; j += 2) {
            const int4 a = *reinterpret_cast<const int4*>(&s_rec[j][0]);
            const int4 b = *reinterpret_cast<const int4*>(&s_rec[j + 1][0]);
            const int va = ((uint32_t)a.x <= qt && (uint32_t)a.y < qoff) ? a.z : INT32_MIN;
            const int vb = ((uint32_t)b.x <= qt && (uint32_t)b.y < qoff) ? b.z : INT32_MIN;
            acc = max(acc, max(va, vb));
        }
    }
    if (qoff != 0 && acc > enc(CL_CHAIN_NEG)) atomicMax(cb.acc + (size_t)s * 7, acc);
}

// one workgroup, kChainBlock threads: thread i owns sorted match pair block_first + i.  Pairs that start on the same
// graph-1 node cannot precede one another (a predecessor must END before the start), so the block is walked group by
// group: every pair of a group is finalised at once and the group's records are broadcast through LDS.
// near_blocks > 0 (single-combination problems): the kernel first evaluates the records of the near_blocks blocks before this
// one against its queries itself — that is the "near" launch folded into the walk, one launch per block less on the
// sequential stream.
__global__ void __launch_bounds__(kChainBlock) chain_intra_kernel(ClChainDevice D, uint32_t block_first, uint32_t block_count,
                                                                  uint32_t near_blocks) {
    const uint32_t i = threadIdx.x;
    const uint32_t s = block_first + i;
    const bool active = i < block_count;
    const ClChainCombo* combos = D.combos;
    const ClChainCombo c0 = combos[0];  // the first combination lives in registers (the only one in a pairwise problem)
    uint32_t qt0 = 0, qoff0 = 0;
    int32_t q0 = 0;
    int acc0[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc0[k] = enc(CL_CHAIN_NEG);
    float w = 0.f, w_init = 0.f;
    uint32_t r0 = 0, r1 = 0, my_combo = 0, my_pos = 0, my_ins = 0, my_off = 0, my_group = 0xFFFFFFFFu;
    int32_t my_sig = 0;
    bool has_q0 = false;
    if (active) {
        has_q0 = c0.qt[s] != 0xFFFFFFFFu;
        if (has_q0) { qt0 = c0.qt[s]; qoff0 = c0.qoff[s]; q0 = c0.q[s]; }
#pragma unroll
        for (int k = 0; k < 7; ++k) acc0[k] = c0.acc[(size_t)s * 7 + k];
        w = D.weight[s];
        w_init = D.init[s];
        my_group = D.group[s];
        r0 = D.rec_off[s]; r1 = D.rec_off[s + 1];
        if (r1 > r0) {
            my_combo = D.rec_combo[r0]; my_pos = D.rec_pos[r0];
            const ClChainCombo cc = combos[my_combo];
            my_ins = cc.ins_t[my_pos]; my_off = cc.off[my_pos]; my_sig = cc.sigma[my_pos];
        }
    }
    double pen0[6];
    query_penalties(pen0, q0, D.params);
    double my_t[3];  // scale * extend_k * shift of the pair's first record (anchorer.hpp:2330, 2334)
#pragma unroll
    for (int k = 0; k < 3; ++k) my_t[k] = D.params.scale * D.params.gap_extend[k] * (double)my_sig;
    const bool simple = D.n_combos == 1;
    // LDS: the records published by the current group (first kChainLdsRecs of them; the rest are re-read from HBM)
    __shared__ uint32_t s_count;
    __shared__ __attribute__((aligned(16))) int s_rec[kChainLdsRecs][12];  // ins_t, off, sigma, 7 encoded values, combo, -

    if (near_blocks) {   // simple == true
        const uint32_t bcur = block_first / kChainBlock;
        const uint32_t rec_lo = c0.prefix[bcur - near_blocks], rec_hi = c0.prefix[bcur];
        for (uint32_t base = rec_lo; base < rec_hi; base += kChainBlock) {
            const uint32_t r = base + i;
            __syncthreads();
            if (r < rec_hi) {
                s_rec[i][0] = (int)c0.ins_t[r];
                s_rec[i][1] = (int)c0.off[r];
                s_rec[i][2] = c0.sigma[r];
#pragma unroll
                for (int k = 0; k < 7; ++k) s_rec[i][3 + k] = enc(c0.val[(size_t)k * c0.n_recs + r]);
            } else {
                s_rec[i][0] = -1;   // insertion index 0xFFFFFFFF: never a predecessor
                s_rec[i][1] = -1;
            }
            __syncthreads();
            if (has_q0)
                for (uint32_t l = 0; l < kChainBlock; l += 4) {
                    int4 ra[4], rb[4], rc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ra[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][0]);
                        rb[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][4]);
                        rc[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][8]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int v[7] = {ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w, rc[u].x, rc[u].y};
                        accumulate(acc0, qt0, qoff0, q0, (uint32_t)ra[u].x, (uint32_t)ra[u].y, ra[u].z, v);
                    }
                }
        }
        __syncthreads();
    }

    // group ids are consecutive along the sorted pairs, so the block holds every id from its first pair's to its last pair's;
    // the LDS slot of a pair's records and the record count of its (block, group) come precomputed from the host: no LDS
    // atomics in the walk (256 threads hammering one LDS word per group were most of this kernel's time)
    const uint32_t last_group = D.group[block_first + block_count - 1];
    const uint32_t my_base = active ? D.grp_base[s] : 0u, my_total = active ? D.grp_total[s] : 0u;
    uint32_t group = D.group[block_first];
    while (true) {
        if (active && my_group == group) {
            // finalise: dp = max(chain starting here, every candidate) — anchorer.hpp:2026-2041, 2379-2412
            float best = w_init;
            if (has_q0) best = apply_candidates(best, acc0, w, pen0);
            if (!simple) {
                for (uint32_t c = 1; c < D.n_combos; ++c) {
                    const ClChainCombo cc = combos[c];
                    if (cc.qt[s] == 0xFFFFFFFFu) continue;
                    int a[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) a[k] = cc.acc[(size_t)s * 7 + k];
                    double pen[6];
                    query_penalties(pen, cc.q[s], D.params);
                    best = apply_candidates(best, a, w, pen);
                }
            }
            D.dp[s] = best;
            // publish the pair's records: the values stored in every tree it sits in (anchorer.hpp:2318-2342)
            const uint32_t base = my_base;
            if (my_base == 0) s_count = my_total;   // the group's first pair(s) in this block (all write the same value)
            for (uint32_t r = r0; r < r1; ++r) {
                uint32_t c = my_combo, pos = my_pos, ins = my_ins, off = my_off;
                int32_t sg = my_sig;
                if (r != r0) {
                    c = D.rec_combo[r]; pos = D.rec_pos[r];
                    const ClChainCombo cc = combos[c];
                    ins = cc.ins_t[pos]; off = cc.off[pos]; sg = cc.sigma[pos];
                }
                float v[7];
                v[0] = best;
#pragma unroll
                for (int pw = 0; pw < 6; ++pw) {
                    const double t = r == r0 ? my_t[pw / 2] : D.params.scale * D.params.gap_extend[pw / 2] * (double)sg;
                    v[1 + pw] = (pw % 2 == 1) ? (float)((double)best + t) : (float)((double)best - t);
                }
                float* vout = (c == 0 ? c0.val : combos[c].val);
                const uint32_t nrec = (c == 0 ? c0.n_recs : combos[c].n_recs);
#pragma unroll
                for (int k = 0; k < 7; ++k) vout[(size_t)k * nrec + pos] = v[k];
                const uint32_t l = base + (r - r0);
                if (l < kChainLdsRecs) {
                    s_rec[l][0] = (int)ins; s_rec[l][1] = (int)off; s_rec[l][2] = sg;
#pragma unroll
                    for (int k = 0; k < 7; ++k) s_rec[l][3 + k] = enc(v[k]);
                    s_rec[l][10] = (int)c;
                }
            }
        }
        lds_barrier();
        const uint32_t n = s_count;
        if (n > kChainLdsRecs) __syncthreads();   // the overflow path below reads the group's records back from HBM
        if (active && my_group > group) {
            if (simple && n <= kChainLdsRecs) {
                // one combination (a pairwise problem): batches of four records, their LDS reads issued together — the loop
                // is bound by LDS latency, not by the arithmetic
                if (has_q0) {
                    uint32_t l = 0;
                    for (; l + 4 <= n; l += 4) {
                        int4 ra[4], rb[4], rc[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ra[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][0]);
                            rb[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][4]);
                            rc[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][8]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int v[7] = {ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w, rc[u].x, rc[u].y};
                            accumulate(acc0, qt0, qoff0, q0, (uint32_t)ra[u].x, (uint32_t)ra[u].y, ra[u].z, v);
                        }
                    }
                    for (; l < n; ++l) {
                        const int4 a = *reinterpret_cast<const int4*>(&s_rec[l][0]);
                        const int4 b = *reinterpret_cast<const int4*>(&s_rec[l][4]);
                        const int4 cc4 = *reinterpret_cast<const int4*>(&s_rec[l][8]);
                        const int v[7] = {a.w, b.x, b.y, b.z, b.w, cc4.x, cc4.y};
                        accumulate(acc0, qt0, qoff0, q0, (uint32_t)a.x, (uint32_t)a.y, a.z, v);
                    }
                }
            } else if (n <= kChainLdsRecs) {
                for (uint32_t l = 0; l < n; ++l) {
                    const int4 a = *reinterpret_cast<const int4*>(&s_rec[l][0]);
                    const int4 b = *reinterpret_cast<const int4*>(&s_rec[l][4]);
                    const int4 cc4 = *reinterpret_cast<const int4*>(&s_rec[l][8]);
                    const int v[7] = {a.w, b.x, b.y, b.z, b.w, cc4.x, cc4.y};
                    const uint32_t c = (uint32_t)cc4.z;
                    if (c == 0) {
                        if (has_q0) accumulate(acc0, qt0, qoff0, q0, (uint32_t)a.x, (uint32_t)a.y, a.z, v);
                    } else {
                        const ClChainCombo cc = combos[c];
                        const uint32_t qt = cc.qt[s];
                        if (qt == 0xFFFFFFFFu) continue;
                        int ac[7];
#pragma unroll
                        for (int k = 0; k < 7; ++k) ac[k] = cc.acc[(size_t)s * 7 + k];
                        accumulate(ac, qt, cc.qoff[s], cc.q[s], (uint32_t)a.x, (uint32_t)a.y, a.z, v);
#pragma unroll
                        for (int k = 0; k < 7; ++k) cc.acc[(size_t)s * 7 + k] = ac[k];
                    }
                }
            } else {
                // a very large group: walk its pairs' records in HBM (written above, visible after the barrier)
                for (uint32_t sj = block_first; sj < block_first + block_count; ++sj) {
                    if (D.group[sj] != group) continue;
                    for (uint32_t r = D.rec_off[sj]; r < D.rec_off[sj + 1]; ++r) {
                        const uint32_t c = D.rec_combo[r], pos = D.rec_pos[r];
                        const ClChainCombo cc = combos[c];
                        const uint32_t qt = cc.qt[s];
                        if (qt == 0xFFFFFFFFu) continue;
                        int v[7];
#pragma unroll
                        for (int k = 0; k < 7; ++k) v[k] = enc(cc.val[(size_t)k * cc.n_recs + pos]);
                        if (c == 0) {
                            accumulate(acc0, qt0, qoff0, q0, cc.ins_t[pos], cc.off[pos], cc.sigma[pos], v);
                        } else {
                            int ac[7];
#pragma unroll
                            for (int k = 0; k < 7; ++k) ac[k] = cc.acc[(size_t)s * 7 + k];
                            accumulate(ac, qt, cc.qoff[s], cc.q[s], cc.ins_t[pos], cc.off[pos], cc.sigma[pos], v);
#pragma unroll
                            for (int k = 0; k < 7; ++k) cc.acc[(size_t)s * 7 + k] = ac[k];
                        }
                    }
                }
            }
        }
        if (group == last_group) break;
        ++group;
        lds_barrier();
    }
    // keep the final maxima: the traceback needs the value every query returned
    if (active) {
#pragma unroll
        for (int k = 0; k < 7; ++k) c0.acc[(size_t)s * 7 + k] = acc0[k];
    }
}

// the same walk for sparse_chain_dp on a single chain pair (every pairwise merge): a pair has at most one record, a query one
// running maximum, a record is 4 dwords
__global__ void __launch_bounds__(kChainBlock) chain_intra_sparse_kernel(ClChainDevice D, uint32_t block_first, uint32_t block_count,
                                                                         uint32_t near_blocks) {
    const uint32_t i = threadIdx.x;
    const uint32_t s = block_first + i;
    const bool active = i < block_count;
    const ClChainCombo c0 = D.combos[0];
    uint32_t qt0 = 0, qoff0 = 0, my_group = 0xFFFFFFFFu, my_base = 0, my_total = 0, my_pos = 0, my_ins = 0, my_off = 0;
    int acc = enc(CL_CHAIN_NEG);
    float w = 0.f, w_init = 0.f;
    bool has_q0 = false, has_rec = false;
    if (active) {
        has_q0 = c0.qt[s] != 0xFFFFFFFFu;
        if (has_q0) { qt0 = c0.qt[s]; qoff0 = c0.qoff[s]; }
        acc = c0.acc[(size_t)s * 7];
        w = D.weight[s];
        w_init = D.init[s];
        my_group = D.group[s];
        my_base = D.grp_base[s];
        my_total = D.grp_total[s];
        const uint32_t r0 = D.rec_off[s];
        has_rec = D.rec_off[s + 1] > r0;
        if (has_rec) { my_pos = D.rec_pos[r0]; my_ins = c0.ins_t[my_pos]; my_off = c0.off[my_pos]; }
    }
    __shared__ uint32_t s_count;
    __shared__ __attribute__((aligned(16))) int s_rec[kChainBlock][4];
    if (near_blocks) {   // the "near" launch folded in: the records of the blocks just before this one
        const uint32_t bcur = block_first / kChainBlock;
        const uint32_t rec_lo = c0.prefix[bcur - near_blocks], rec_hi = c0.prefix[bcur];
        for (uint32_t base = rec_lo; base < rec_hi; base += kChainBlock) {
            const uint32_t r = base + i;
            __syncthreads();
            int4 mine = make_int4(-1, -1, INT32_MIN, 0);
            if (r < rec_hi) mine = make_int4((int)c0.ins_t[r], (int)c0.off[r], enc(c0.val[r]), 0);
            *reinterpret_cast<int4*>(&s_rec[i][0]) = mine;
            __syncthreads();
            if (has_q0)
                for (uint32_t l = 0; l < kChainBlock; l += 4) {
                    int4 r4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) r4[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][0]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = max(acc, ((uint32_t)r4[u].x <= qt0 && (uint32_t)r4[u].y < qoff0) ? r4[u].z : INT32_MIN);
                }
        }
        __syncthreads();
    }
    const uint32_t last_group = D.group[block_first + block_count - 1];
    uint32_t group = D.group[block_first];
    while (true) {
        if (active && my_group == group) {
            // dp = max(chain starting here, best predecessor + weight) — anchorer.hpp:1640-1700
            float best = w_init;
            if (has_q0 && acc != enc(CL_CHAIN_NEG)) best = fmaxf(best, dec(acc) + w);
            D.dp[s] = best;
            if (has_rec) {
                c0.val[my_pos] = best;
                *reinterpret_cast<int4*>(&s_rec[my_base][0]) = make_int4((int)my_ins, (int)my_off, enc(best), 0);
            }
            if (my_base == 0) s_count = my_total;
        }
        lds_barrier();
        const uint32_t n = s_count;
        if (active && has_q0 && my_group > group) {
            uint32_t l = 0;
            for (; l + 4 <= n; l += 4) {
                int4 r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][0]);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = max(acc, ((uint32_t)r[u].x <= qt0 && (uint32_t)r[u].y < qoff0) ? r[u].z : INT32_MIN);
            }
            for (; l < n; ++l) {
                const int4 r = *reinterpret_cast<const int4*>(&s_rec[l][0]);
                acc = max(acc, ((uint32_t)r.x <= qt0 && (uint32_t)r.y < qoff0) ? r.z : INT32_MIN);
            }
        }
        if (group == last_group) break;
        ++group;
        lds_barrier();
    }
    if (active) c0.acc[(size_t)s * 7] = acc;   // the traceback needs the value the query returned
}


// ---------------------------------------------------------------------------------------------------------------------
// chain_walk_kernel — the sequential part of the DP for kChainMacro consecutive match pairs, ONE WORKGROUP PER CHAIN
// COMBINATION.  Thread t of workgroup c owns the query of sorted pair first + t in combination c: its 7 running maxima live in
// registers for the whole walk, and only the records of combination c pass through the workgroup's LDS.  The pairs are walked
// group by group (pairs of a group cannot precede one another).  Finalising a pair needs the best candidate of EVERY
// combination (anchorer.hpp:2352-2416 loops over forward edges x chains of graph 2): each workgroup stores its combination's
// candidate as an 8-byte {tag, value} granule (write-through, agent scope) and the group's threads of every workgroup sweep the
// n_combos granules of their pair until all carry the pair's tag — the data is the flag, no fence and no counter
// (cdna_hip_programming.md §6 Guideline 16, recipe R2).  Every workgroup derives the same dp value from the same granules, so
// the stored values it publishes for its own combination are the reference's (anchorer.hpp:2318-2342).
// The workgroups wait for one another, so all n_combos of them must be resident: the host uses this kernel only up to
// kChainWalkMaxCombos combinations; every spin is bounded and reports through D.status.
template <bool SPARSE>
__global__ void __launch_bounds__(kChainMacro) chain_walk_kernel(ClChainDevice D, uint32_t first, uint32_t count) {
    constexpr int NK = SPARSE ? 1 : 7;
    constexpr int RW = SPARSE ? 4 : 12;
    const uint32_t c = blockIdx.x;
    const ClChainCombo cb = D.combos[c];
    const uint32_t t = threadIdx.x, s = first + t;
    const bool active = t < count;
    const int none = enc(CL_CHAIN_NEG);
    bool has_q = false, has_rec = false;
    uint32_t qt = 0, qoff = 0, pos = 0, ins = 0, off = 0;
    int32_t q = 0, sig = 0;
    int acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = none;
    float w = 0.f, w_init = CL_CHAIN_NEG;
    if (active) {
        qt = cb.qt[s];
        has_q = qt != 0xFFFFFFFFu;
        if (has_q) { qoff = cb.qoff[s]; q = cb.q[s]; }
#pragma unroll
        for (int k = 0; k < NK; ++k) acc[k] = cb.acc[(size_t)s * 7 + k];
        w = D.weight[s];
        w_init = D.init[s];
        pos = cb.own_rec[s];
        has_rec = pos != 0xFFFFFFFFu;
        if (has_rec) { ins = cb.ins_t[pos]; off = cb.off[pos]; sig = cb.sigma[pos]; }
    }
    if (!has_q) qoff = 0;   // a record's offset is never below 0: such a lane accumulates nothing
    __shared__ __attribute__((aligned(16))) int s_rec[kChainMacro][RW];   // slot t = the record of pair first + t in this combination
    __shared__ int s_abort;
    if (t == 0) s_abort = 0;
    __syncthreads();
    const uint32_t end = first + count;
    const uint32_t n_combos = D.n_combos;
    uint32_t cur = first;
    while (cur < end) {
        const uint32_t gend = min(D.group_end[cur], end);
        if (active && s >= cur && s < gend) {
            // this combination's best candidate for the pair (anchorer.hpp:2379-2412)
            float cand = CL_CHAIN_NEG;
            if (has_q) {
                if (SPARSE) {
                    if (acc[0] != none) cand = dec(acc[0]) + w;
                } else {
                    double pen[6];
                    query_penalties(pen, q, D.params);
                    cand = apply_candidates(CL_CHAIN_NEG, acc, w, pen);
                }
            }
            float best = fmaxf(w_init, cand);
            if (n_combos > 1 && D.xred) {
                // many combinations: instead of every workgroup reading every other's granule (n_combos^2 loads per pair), one atomic maximum
                // and one arrival count per pair; the maximum of the same set of floats, whoever comes first
                uint32_t* red = D.xred + 2 * (size_t)s;
                if (cand != CL_CHAIN_NEG) __hip_atomic_fetch_max(red, (uint32_t)enc(cand) ^ 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(red + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(red + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n_combos) {
                    if (++spins > (1u << 20) || ((spins & 1023u) == 0 && __hip_atomic_load(D.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicExch(D.status, 1u);
                        s_abort = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t m = __hip_atomic_load(red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (m != 0) best = fmaxf(best, dec((int)(m ^ 0x80000000u)));
            } else if (n_combos > 1) {
                const unsigned long long tag = (unsigned long long)(s + 1u) << 32;
                __hip_atomic_store(&D.xch[(size_t)c * kChainMacro + t], tag | (unsigned)__float_as_int(cand), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (true) {
                    bool ok = true;
                    float m = w_init;
                    for (uint32_t cc = 0; cc < n_combos; cc += 4) {
                        unsigned long long x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            x[u] = cc + u < n_combos ? __hip_atomic_load(&D.xch[(size_t)(cc + u) * kChainMacro + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag | (unsigned)__float_as_int(CL_CHAIN_NEG);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ok = ok && (x[u] >> 32) == (tag >> 32);
                            m = fmaxf(m, __int_as_float((int)(unsigned)x[u]));
                        }
                    }
                    if (ok) { best = m; break; }
                    // a sibling workgroup never arrived (not resident), or another one has already given up
                    if (++spins > (1u << 20) || ((spins & 1023u) == 0 && __hip_atomic_load(D.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicExch(D.status, 1u);
                        s_abort = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (c == 0) D.dp[s] = best;
            // publish the pair's record of this combination: the values stored in its trees (anchorer.hpp:2318-2342)
            if (has_rec) {
                float v[7];
                v[0] = best;
                if (!SPARSE) {
#pragma unroll
                    for (int pw = 0; pw < 6; ++pw) {
                        const double tt = D.params.scale * D.params.gap_extend[pw / 2] * (double)sig;
                        v[1 + pw] = (pw % 2 == 1) ? (float)((double)best + tt) : (float)((double)best - tt);
                    }
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) cb.val[(size_t)k * cb.n_recs + pos] = v[k];
                if (D.far_rec) {   // the image the branch-and-bound far pass reads (chain_far.hip)
                    int4* fr = reinterpret_cast<int4*>(D.far_rec + (size_t)(D.far_base[c] + pos) * 12);
                    fr[0] = make_int4((int)ins, (int)off, sig, enc(v[0]));
                    if (!SPARSE) {
                        fr[1] = make_int4(enc(v[1]), enc(v[2]), enc(v[3]), enc(v[4]));
                        fr[2] = make_int4(enc(v[5]), enc(v[6]), 0, 0);
                    }
                }
                if (SPARSE) {
                    *reinterpret_cast<int4*>(&s_rec[t][0]) = make_int4((int)ins, (int)off, enc(best), 0);
                } else {
                    *reinterpret_cast<int4*>(&s_rec[t][0]) = make_int4((int)ins, (int)off, sig, enc(v[0]));
                    *reinterpret_cast<int4*>(&s_rec[t][4]) = make_int4(enc(v[1]), enc(v[2]), enc(v[3]), enc(v[4]));
                    *reinterpret_cast<int4*>(&s_rec[t][8]) = make_int4(enc(v[5]), enc(v[6]), 0, 0);
                }
            } else {
                *reinterpret_cast<int4*>(&s_rec[t][0]) = make_int4(-1, -1, 0, INT32_MIN);   // insertion index 0xFFFFFFFF: never a predecessor
            }
        }
        lds_barrier();
        if (s_abort) break;
        if (active && s >= gend && has_q) {
            const uint32_t l0 = cur - first, l1 = gend - first;
            if (SPARSE) {
                for (uint32_t l = l0; l < l1; ++l) {
                    const int4 r = *reinterpret_cast<const int4*>(&s_rec[l][0]);
                    acc[0] = max(acc[0], ((uint32_t)r.x <= qt && (uint32_t)r.y < qoff) ? r.z : INT32_MIN);
                }
            } else {
                uint32_t l = l0;
                for (; l + 2 <= l1; l += 2) {
                    int4 ra[2], rb[2], rc[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        ra[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][0]);
                        rb[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][4]);
                        rc[u] = *reinterpret_cast<const int4*>(&s_rec[l + u][8]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int v[7] = {ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w, rc[u].x, rc[u].y};
                        accumulate(acc, qt, qoff, q, (uint32_t)ra[u].x, (uint32_t)ra[u].y, ra[u].z, v);
                    }
                }
                for (; l < l1; ++l) {
                    const int4 a = *reinterpret_cast<const int4*>(&s_rec[l][0]);
                    const int4 b = *reinterpret_cast<const int4*>(&s_rec[l][4]);
                    const int4 cc4 = *reinterpret_cast<const int4*>(&s_rec[l][8]);
                    const int v[7] = {a.w, b.x, b.y, b.z, b.w, cc4.x, cc4.y};
                    accumulate(acc, qt, qoff, q, (uint32_t)a.x, (uint32_t)a.y, a.z, v);
                }
            }
        }
        cur = gend;
    }
    // keep the final maxima: the traceback needs the value every query returned
    if (active) {
#pragma unroll
        for (int k = 0; k < NK; ++k) cb.acc[(size_t)s * 7 + k] = acc[k];
    }
}

// chain_walk_fold_kernel — chain_walk_kernel for MORE combinations than workgroups can be resident (kChainWalkMaxCombos): workgroup c
// walks the F combinations c, c + G, c + 2 G (G workgroups in all), thread t holding the query of pair first + t in each of them (F x 7
// running maxima in registers) and F record images in LDS (F x 48 KB in the affine DP: F <= 3, a root of 27 + 27 paths).  A workgroup's
// candidate for a pair is the maximum over its F combinations; the workgroups exchange through the atomic maximum + arrival count per
// pair (D.xred), counted in workgroups.  Same arithmetic and same stored values as chain_walk_kernel, combination by combination.
template <bool SPARSE, int F>
__global__ void __launch_bounds__(kChainMacro) chain_walk_fold_kernel(ClChainDevice D, uint32_t first, uint32_t count) {
    constexpr int NK = SPARSE ? 1 : 7;
    constexpr int RW = SPARSE ? 4 : 12;
    const uint32_t c0 = blockIdx.x, G = gridDim.x;
    const uint32_t t = threadIdx.x, s = first + t;
    const bool active = t < count;
    const int none = enc(CL_CHAIN_NEG);
    const uint32_t n_combos = D.n_combos;
    bool has_q[F], has_rec[F];
    uint32_t qt[F], qoff[F], pos[F];
    int32_t q[F];
    int acc[F][7];
    float w = 0.f, w_init = CL_CHAIN_NEG;
    if (active) { w = D.weight[s]; w_init = D.init[s]; }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        has_q[f] = has_rec[f] = false;
        qt[f] = 0; qoff[f] = 0; pos[f] = 0xFFFFFFFFu; q[f] = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[f][k] = none;
        const uint32_t c = c0 + f * G;
        if (active && c < n_combos) {
            const ClChainCombo& cb = D.combos[c];
            qt[f] = cb.qt[s];
            has_q[f] = qt[f] != 0xFFFFFFFFu;
            if (has_q[f]) { qoff[f] = cb.qoff[s]; q[f] = cb.q[s]; }
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[f][k] = cb.acc[(size_t)s * 7 + k];
            pos[f] = cb.own_rec[s];
            has_rec[f] = pos[f] != 0xFFFFFFFFu;
        }
        if (!has_q[f]) qoff[f] = 0;   // a record's offset is never below 0: such a lane accumulates nothing
    }
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [F][kChainMacro][RW], then the abort flag
    int (*s_rec)[kChainMacro][RW] = reinterpret_cast<int (*)[kChainMacro][RW]>(s_dyn);
    int* s_abort = s_dyn + (size_t)F * kChainMacro * RW;
    if (t == 0) *s_abort = 0;
    __syncthreads();
    const uint32_t end = first + count;
    uint32_t cur = first;
    while (cur < end) {
        const uint32_t gend = min(D.group_end[cur], end);
        if (active && s >= cur && s < gend) {
            float cand = CL_CHAIN_NEG;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (!has_q[f]) continue;
                if (SPARSE) {
                    if (acc[f][0] != none) cand = fmaxf(cand, dec(acc[f][0]) + w);
                } else {
                    double pen[6];
                    query_penalties(pen, q[f], D.params);
                    cand = apply_candidates(cand, acc[f], w, pen);
                }
            }
            float best = fmaxf(w_init, cand);
            {
                uint32_t* red = D.xred + 2 * (size_t)s;
                if (cand != CL_CHAIN_NEG) __hip_atomic_fetch_max(red, (uint32_t)enc(cand) ^ 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(red + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(red + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < G) {
                    if (++spins > (1u << 20) || ((spins & 1023u) == 0 && __hip_atomic_load(D.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicExch(D.status, 1u);
                        *s_abort = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t m = __hip_atomic_load(red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (m != 0) best = fmaxf(best, dec((int)(m ^ 0x80000000u)));
            }
            if (c0 == 0) D.dp[s] = best;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (has_rec[f]) {
                    const ClChainCombo& cb = D.combos[c0 + f * G];
                    const uint32_t ins = cb.ins_t[pos[f]], off = cb.off[pos[f]];
                    const int32_t sig = cb.sigma[pos[f]];
                    float v[7];
                    v[0] = best;
                    if (!SPARSE) {
#pragma unroll
                        for (int pw = 0; pw < 6; ++pw) {
                            const double tt = D.params.scale * D.params.gap_extend[pw / 2] * (double)sig;
                            v[1 + pw] = (pw % 2 == 1) ? (float)((double)best + tt) : (float)((double)best - tt);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NK; ++k) cb.val[(size_t)k * cb.n_recs + pos[f]] = v[k];
                    if (D.far_rec) {
                        int4* fr = reinterpret_cast<int4*>(D.far_rec + (size_t)(D.far_base[c0 + f * G] + pos[f]) * 12);
                        fr[0] = make_int4((int)ins, (int)off, sig, enc(v[0]));
                        if (!SPARSE) {
                            fr[1] = make_int4(enc(v[1]), enc(v[2]), enc(v[3]), enc(v[4]));
                            fr[2] = make_int4(enc(v[5]), enc(v[6]), 0, 0);
                        }
                    }
                    if (SPARSE) {
                        *reinterpret_cast<int4*>(&s_rec[f][t][0]) = make_int4((int)ins, (int)off, enc(best), 0);
                    } else {
                        *reinterpret_cast<int4*>(&s_rec[f][t][0]) = make_int4((int)ins, (int)off, sig, enc(v[0]));
                        *reinterpret_cast<int4*>(&s_rec[f][t][4]) = make_int4(enc(v[1]), enc(v[2]), enc(v[3]), enc(v[4]));
                        *reinterpret_cast<int4*>(&s_rec[f][t][8]) = make_int4(enc(v[5]), enc(v[6]), 0, 0);
                    }
                } else {
                    *reinterpret_cast<int4*>(&s_rec[f][t][0]) = make_int4(-1, -1, 0, INT32_MIN);   // insertion index 0xFFFFFFFF: never a predecessor
                }
            }
        }
        lds_barrier();
        if (*s_abort) break;
        if (active && s >= gend) {
            const uint32_t l0 = cur - first, l1 = gend - first;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (!has_q[f]) continue;
                if (SPARSE) {
                    for (uint32_t l = l0; l < l1; ++l) {
                        const int4 r = *reinterpret_cast<const int4*>(&s_rec[f][l][0]);
                        acc[f][0] = max(acc[f][0], ((uint32_t)r.x <= qt[f] && (uint32_t)r.y < qoff[f]) ? r.z : INT32_MIN);
                    }
                } else {
                    for (uint32_t l = l0; l < l1; ++l) {
                        const int4 a = *reinterpret_cast<const int4*>(&s_rec[f][l][0]);
                        if (a.x == -1) continue;   // (most pairs have no record in a given combination when there are hundreds of them)
                        const int4 b = *reinterpret_cast<const int4*>(&s_rec[f][l][4]);
                        const int4 cc4 = *reinterpret_cast<const int4*>(&s_rec[f][l][8]);
                        const int v[7] = {a.w, b.x, b.y, b.z, b.w, cc4.x, cc4.y};
                        accumulate(acc[f], qt[f], qoff[f], q[f], (uint32_t)a.x, (uint32_t)a.y, a.z, v);
                    }
                }
            }
        }
        cur = gend;
    }
    // keep the final maxima: the traceback needs the value every query returned
    if (active) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const uint32_t c = c0 + f * G;
            if (c >= n_combos) continue;
#pragma unroll
            for (int k = 0; k < NK; ++k) D.combos[c].acc[(size_t)s * 7 + k] = acc[f][k];
        }
    }
}

// own_rec[c][s] = position of pair s's record in combination c (a pair has at most one record per combination)
__global__ void __launch_bounds__(256) chain_own_rec_kernel(const ClChainCombo* combos) {
    const ClChainCombo cb = combos[blockIdx.y];
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < cb.n_recs) cb.own_rec[cb.rec_s[r]] = r;
}

hipError_t cl_chain_launch_own_rec(const ClChainDevice& D, uint32_t max_recs, hipStream_t stream) {
    if (max_recs == 0) return hipSuccess;
    hipLaunchKernelGGL(chain_own_rec_kernel, dim3((max_recs + 255) / 256, D.n_combos), dim3(256), 0, stream, D.combos);
    return hipGetLastError();
}

// the query results of ONE pair in every combination (7 per combination), for a traceback that fetches rows as it goes instead of downloading
// 28 bytes x pairs x combinations (cl_chain_api.cpp)
__global__ void __launch_bounds__(256) chain_acc_row_kernel(const ClChainCombo* combos, uint32_t n_combos, uint32_t s, int* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_combos * 7) out[i] = combos[i / 7].acc[(size_t)s * 7 + i % 7];
}

// the dense query tables out of their factors (cl_chain_api.cpp: factored queries): combination c = (tag1, tag2); fa_d == nullptr: the gap-free DP (no shifts)
__global__ void __launch_bounds__(256) chain_expand_queries_kernel(const uint32_t* fa_qt, const uint32_t* fa_d, const uint32_t* fb_qoff, const uint32_t* fb_d,
                                                                   const uint32_t* combo_tags, uint32_t n_pairs, uint32_t* qt, uint32_t* qoff, int32_t* q) {
    const uint32_t c = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_pairs) return;
    const size_t a = (size_t)combo_tags[2 * c] * n_pairs + s, b = (size_t)combo_tags[2 * c + 1] * n_pairs + s, at = (size_t)c * n_pairs + s;
    const uint32_t t = fa_qt[a];
    const bool ok = t != 0xFFFFFFFFu;
    qt[at] = t;
    qoff[at] = ok ? fb_qoff[b] : 0u;
    q[at] = ok && fa_d ? (int32_t)(fa_d[a] - fb_d[b]) : 0;
}

hipError_t cl_chain_expand_queries(const uint32_t* fa_qt, const uint32_t* fa_d, const uint32_t* fb_qoff, const uint32_t* fb_d, const uint32_t* combo_tags, uint32_t n_combos,
                                   uint32_t n_pairs, uint32_t* qt, uint32_t* qoff, int32_t* q, hipStream_t stream) {
    if (n_combos == 0 || n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(chain_expand_queries_kernel, dim3((n_pairs + 255) / 256, n_combos), dim3(256), 0, stream, fa_qt, fa_d, fb_qoff, fb_d, combo_tags, n_pairs, qt, qoff, q);
    return hipGetLastError();
}

hipError_t cl_chain_acc_row(const ClChainDevice& D, uint32_t s, int* out, hipStream_t stream) {
    hipLaunchKernelGGL(chain_acc_row_kernel, dim3((D.n_combos * 7 + 255) / 256), dim3(256), 0, stream, D.combos, D.n_combos, s, out);
    return hipGetLastError();
}

// fold = combinations per workgroup (2 or 3; needs D.xred)
template <bool SPARSE, int F>
static hipError_t launch_walk_fold(const ClChainDevice& D, uint32_t first, uint32_t count, hipStream_t stream, hipEvent_t done) {
    const size_t lds = (size_t)F * kChainMacro * (SPARSE ? 4 : 12) * sizeof(int) + 16;
    static ClDeviceOnce raised;   // (beyond 64 KB of dynamic LDS the kernel has to be told once per device)
    hipError_t attr_rc = hipSuccess;
    raised([&] { attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_walk_fold_kernel<SPARSE, F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    if (attr_rc != hipSuccess) return attr_rc;
    const uint32_t G = (D.n_combos + F - 1) / F;
    hipExtLaunchKernelGGL((chain_walk_fold_kernel<SPARSE, F>), dim3(G), dim3(kChainMacro), lds, stream, nullptr, done, 0, D, first, count);
    return hipGetLastError();
}

hipError_t cl_chain_launch_walk_fold(const ClChainDevice& D, uint32_t fold, uint32_t first, uint32_t count, hipStream_t stream, hipEvent_t done) {
    if (!D.xred || fold < 2 || fold > 3) return hipErrorInvalidValue;
    if (D.sparse) return fold == 2 ? launch_walk_fold<true, 2>(D, first, count, stream, done) : launch_walk_fold<true, 3>(D, first, count, stream, done);
    return fold == 2 ? launch_walk_fold<false, 2>(D, first, count, stream, done) : launch_walk_fold<false, 3>(D, first, count, stream, done);
}

hipError_t cl_chain_launch_walk(const ClChainDevice& D, uint32_t first, uint32_t count, hipStream_t stream, hipEvent_t done) {   // done: see cl_chain_launch_walk2
    if (D.sparse) hipExtLaunchKernelGGL(chain_walk_kernel<true>, dim3(D.n_combos), dim3(kChainMacro), 0, stream, nullptr, done, 0, D, first, count);
    else hipExtLaunchKernelGGL(chain_walk_kernel<false>, dim3(D.n_combos), dim3(kChainMacro), 0, stream, nullptr, done, 0, D, first, count);
    return hipGetLastError();
}

// tile_recs (a multiple of 256): predecessor records per workgroup.  A launch lasts at least one tile, and every workgroup
// ends with one atomic merge per query and kind: large tiles for the bulk ("far") launches, small ones for the short
// "near" launches that sit on the sequential stream.
hipError_t cl_chain_launch_inter(const ClChainDevice& D, uint32_t block_first, uint32_t block_count, uint32_t src_block_lo,
                                 uint32_t src_block_hi, uint32_t max_recs, uint32_t tile_recs, hipStream_t stream) {
    if (max_recs == 0) return hipSuccess;
    const uint32_t tiles = (max_recs + tile_recs - 1) / tile_recs;
    if (D.sparse) hipLaunchKernelGGL(chain_inter_sparse_kernel, dim3(tiles, (block_count + 255) / 256, D.n_combos), dim3(256), 0, stream, D, block_first, block_count, src_block_lo, src_block_hi, tile_recs);
    else hipLaunchKernelGGL(chain_inter_kernel, dim3(tiles, (block_count + 255) / 256, D.n_combos), dim3(256), 0, stream, D, block_first, block_count, src_block_lo, src_block_hi, tile_recs);
    return hipGetLastError();
}

// ---- chaining DPs over MORE combinations than the walk kernels take (round 6) ---------------------------------------------------------------------------------
// Beyond 768 chain combinations (27 + 27 paths) a DP used to fall to chain_intra_kernel: ONE workgroup that, per group of pairs, loops over every combination per pair and
// re-reads a group's records from HBM once they exceed its LDS — 584 ms for a realignment of 316 match pairs over 2 601 combinations (51 + 51 paths: fourteen such DPs were
// 8.2 of the 12.7 s a polishing step of the 50 x 8 kbp golden took).  The groups stay the unit of order (pairs of a group cannot precede one another); inside a group
// everything is parallel over the combinations, three small launches per group:
//   chain_group_reduce_kernel  dp of the group's pairs: the maximum over the combinations of the candidates their running maxima give (integer atomic maximum on the
//                              order-preserving encoding, started at the value of a chain that begins at the pair);
//   chain_group_store_kernel   the values the group's records keep in their trees (anchorer.hpp:2318-2342), one thread per record, and dp itself;
//   chain_group_push_kernel    the group's records against the queries of every LATER pair (chain_inter_kernel's tile loop; the group's records of a combination are found
//                              by two binary searches in its sorted record list), merged into the running maxima.
// Same arithmetic as chain_intra_kernel (apply_candidates, the stored values in double), maxima are order-free: bit-identical dp.
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(256) chain_group_reduce_kernel(ClChainDevice D, uint32_t s0, uint32_t s1, int* __restrict__ dp_enc) {
    const uint32_t s = s0 + blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (s >= s1 || c >= D.n_combos) return;
    const ClChainCombo cc = D.combos[c];
    if (cc.qt[s] == 0xFFFFFFFFu) return;
    int a[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) a[k] = cc.acc[(size_t)s * 7 + k];
    double pen[6];
    query_penalties(pen, cc.q[s], D.params);
    const float best = apply_candidates(CL_CHAIN_NEG, a, D.weight[s], pen);
    if (best != CL_CHAIN_NEG) atomicMax(dp_enc + s, enc(best));
}

__global__ void __launch_bounds__(256) chain_group_store_kernel(ClChainDevice D, uint32_t s0, uint32_t s1, const int* __restrict__ dp_enc) {
    const uint32_t r_lo = D.rec_off[s0], r_hi = D.rec_off[s1];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < s1 - s0) D.dp[s0 + i] = dec(dp_enc[s0 + i]);
    const uint32_t r = r_lo + i;
    if (r >= r_hi) return;
    // the pair of record r: the last s in [s0, s1) with rec_off[s] <= r
    uint32_t lo = s0, hi = s1;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (D.rec_off[mid] <= r) lo = mid; else hi = mid; }
    const float best = dec(dp_enc[lo]);
    const uint32_t c = D.rec_combo[r], pos = D.rec_pos[r];
    const ClChainCombo cc = D.combos[c];
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
}

// the records of the pairs [s0, s1) of combination blockIdx.z against the queries of the pairs [s1 + 256 blockIdx.y, ...): tile blockIdx.x of 256 records
__global__ void __launch_bounds__(256) chain_group_push_kernel(ClChainDevice D, uint32_t s0, uint32_t s1, uint32_t n_pairs) {
    const ClChainCombo cb = D.combos[blockIdx.z];
    const uint32_t rec_lo = lower_bound_u32(cb.rec_s, cb.n_recs, s0), rec_hi = lower_bound_u32(cb.rec_s, cb.n_recs, s1);
    const uint32_t tile0 = rec_lo + blockIdx.x * 256;
    if (tile0 >= rec_hi) return;
    const uint32_t tile_end = min(tile0 + 256u, rec_hi);
    const uint32_t s = s1 + blockIdx.y * 256 + threadIdx.x;
    const bool active = s < n_pairs;
    uint32_t qt = 0, qoff = 0;
    int32_t q = 0;
    if (active && cb.qt[s] != 0xFFFFFFFFu) { qt = cb.qt[s]; qoff = cb.qoff[s]; q = cb.q[s]; }
    int acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = INT32_MIN;
    __shared__ __attribute__((aligned(16))) int s_rec[256][12];
    {
        const uint32_t r = tile0 + threadIdx.x;
        if (r < tile_end) {
            s_rec[threadIdx.x][0] = (int)cb.ins_t[r];
            s_rec[threadIdx.x][1] = (int)cb.off[r];
            s_rec[threadIdx.x][2] = D.sparse ? 0 : cb.sigma[r];
#pragma unroll
            for (int i = 0; i < 7; ++i) s_rec[threadIdx.x][3 + i] = (D.sparse && i) ? INT32_MIN : enc(cb.val[(size_t)i * cb.n_recs + r]);
        }
    }
    __syncthreads();
    const uint32_t cnt = tile_end - tile0;
    for (uint32_t j = 0; j < cnt; ++j) {
        const int4 a = *reinterpret_cast<const int4*>(&s_rec[j][0]);
        const int4 b = *reinterpret_cast<const int4*>(&s_rec[j][4]);
        const int4 c = *reinterpret_cast<const int4*>(&s_rec[j][8]);
        const int v[7] = {a.w, b.x, b.y, b.z, b.w, c.x, c.y};
        accumulate(acc, qt, qoff, q, (uint32_t)a.x, (uint32_t)a.y, a.z, v);
    }
    if (qoff != 0) {
        int* dst = cb.acc + (size_t)s * 7;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (acc[i] > enc(CL_CHAIN_NEG)) atomicMax(dst + i, acc[i]);
    }
}

// one group of pairs [s0, s1): reduce -> store -> push.  n_group_recs: records of the group's pairs over all combinations (the store grid); a combination holds at most one
// record per pair, so s1 - s0 bounds what it holds of the group (the push grid)
hipError_t cl_chain_launch_group(const ClChainDevice& D, uint32_t s0, uint32_t s1, uint32_t n_group_recs, int* dp_enc, hipStream_t stream) {
    hipLaunchKernelGGL(chain_group_reduce_kernel, dim3(s1 - s0, (D.n_combos + 255) / 256), dim3(256), 0, stream, D, s0, s1, dp_enc);
    const uint32_t n_store = n_group_recs > s1 - s0 ? n_group_recs : s1 - s0;
    hipLaunchKernelGGL(chain_group_store_kernel, dim3((n_store + 255) / 256), dim3(256), 0, stream, D, s0, s1, dp_enc);
    if (s1 < D.n_pairs && n_group_recs)
        hipLaunchKernelGGL(chain_group_push_kernel, dim3((s1 - s0 + 255) / 256, (D.n_pairs - s1 + 255) / 256, D.n_combos), dim3(256), 0, stream, D, s0, s1, D.n_pairs);
    return hipGetLastError();
}

hipError_t cl_chain_launch_intra(const ClChainDevice& D, uint32_t block_first, uint32_t block_count, uint32_t near_blocks, hipStream_t stream) {
    static const bool no_sparse_intra = getenv("CL_CHAIN_NO_SPARSE_INTRA") != nullptr;   // A/B switch for measurements
    if (D.sparse && D.n_combos == 1 && !no_sparse_intra) hipLaunchKernelGGL(chain_intra_sparse_kernel, dim3(1), dim3(kChainBlock), 0, stream, D, block_first, block_count, near_blocks);
    else hipLaunchKernelGGL(chain_intra_kernel, dim3(1), dim3(kChainBlock), 0, stream, D, block_first, block_count, near_blocks);
    return hipGetLastError();
}
