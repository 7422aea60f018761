// chain_far.hip — the "far" pass of the chaining DP (chain_kernels.hip) as a branch-and-bound instead of an all-pairs sweep.
//
// What a query needs from the records that were final long before it is ONE number per tree kind: the maximum stored value
// over the records in range (include/centrolign/anchorer.hpp:2379-2412).  The all-pairs sweep evaluates every such record; almost
// all of them lose by a wide margin, because DP values grow along the graphs and a gap costs at least its opening penalty.
// This pass proves that for whole blocks of records at once and looks only at the blocks it cannot rule out.
//
// Records of a combination lie in sorted-pair order and are cut into aligned nodes of 64 * 8^l records.  When the walk has
// finalised every record of a node, far_seal_kernel writes, in two static orders of the node's records — by offset, and by
// (shift bucket of 65536, offset) — the running maximum of the DP value.  For a query (offset bound qoff, shift q, weight w)
// and a node, with  A = max dp over the node's records with offset < qoff  and  B = the same over the three shift buckets
// around q, every candidate the node can give is, in exact arithmetic, at most
//        max( B , A - P ) + w          P = least gap cost of a shift difference >= one bucket width,
// because a candidate is dp - cost(shift difference) + w with cost >= 0, and >= P outside the three buckets (the cost of
// anchorer.hpp:1906-1918 grows with the difference).  The kernels round (float stores of double sums, :2318-2342, 2394-2412);
// three roundings of magnitudes below |dp| + scale * extend_0 * (|shift| + |q|) + scale * open_max + |w| are covered by the
// allowance  2^-21 x that magnitude  added to the bound.  A node whose bound is STRICTLY below the query's best candidate so
// far cannot hold a record that attains the query's final DP value in any tree kind, so skipping it changes neither the DP
// value nor any running maximum the traceback will read for an attaining kind (cl_chain_api.cpp); nodes that survive are
// opened level by level, and surviving leaves are evaluated record by record with the exact arithmetic of the sweep.
//
// Eight lanes work on one query: a round tests eight sibling nodes (one per lane), surviving leaves are scanned by the eight
// lanes TOGETHER (eight records each, coalesced, nearest leaf first, re-tested against the best candidate the leaves before it
// gave), surviving inner nodes go on a small per-query stack in LDS (nearest on top); the eight lanes share their best
// candidate after every leaf and every round.
//
// A node's two static orders are kept as 8-ary search trees (chain_device.h): the bound of a node costs level + 2 dependent
// loads (one 32-byte index block per tree level, then one 64-byte block with the keys AND the running maxima) — the binary
// searches of round 2 cost 8 .. 17, and a far launch lasts as long as the dependent loads of its slowest query.
//
// Compiled with -ffp-contract=off like chain_kernels.hip: the leaf evaluation must round like the reference's scalar code.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "cl_radix.h"
#include <stdint.h>

#include "chain_device.h"

namespace {

__device__ __forceinline__ int enc(float f) {
    int b = __float_as_int(f);
    if (b == (int)0x80000000) b = 0;
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__device__ __forceinline__ void accumulate(int (&acc)[7], uint32_t qt, uint32_t qoff, int32_t q, uint32_t ins_t, uint32_t off,
                                           int32_t sigma, const int (&v)[7]) {
    const int none = INT32_MIN;
    const bool ok = ins_t <= qt && off < qoff;
    const bool eq = ok && sigma == q, lt = ok && sigma < q, gt = ok && sigma > q;
    acc[0] = max(acc[0], eq ? v[0] : none);
    acc[2] = max(acc[2], lt ? v[2] : none); acc[4] = max(acc[4], lt ? v[4] : none); acc[6] = max(acc[6], lt ? v[6] : none);
    acc[1] = max(acc[1], gt ? v[1] : none); acc[3] = max(acc[3], gt ? v[3] : none); acc[5] = max(acc[5], gt ? v[5] : none);
}

__device__ __forceinline__ float best_candidate(float best, const int (&acc)[7], float w, const double (&pen)[6]) {
    const int none = enc(CL_CHAIN_NEG);
    if (acc[0] > none) best = fmaxf(best, dec(acc[0]) + w);
#pragma unroll
    for (int pw = 0; pw < 6; ++pw) {
        if (acc[1 + pw] <= none) continue;
        const float cand = (float)((double)(dec(acc[1 + pw]) + w) - pen[pw]);
        best = fmaxf(best, cand);
    }
    return best;
}

// ---- setup: the record image and the sort keys -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) far_init_kernel(const ClChainCombo* combos, const uint32_t* base, uint32_t n_combos, uint32_t r_pad,
                                                       int32_t sig_bias, uint32_t band_shift, uint32_t off_bits, int* rec, uint32_t* key_off, uint32_t* key_band, uint32_t* idx) {
    const uint32_t c = blockIdx.y;
    const ClChainCombo cb = combos[c];
    const uint32_t b0 = base[c], b1 = c + 1 < n_combos ? base[c + 1] : r_pad;
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    if (b0 + pos >= b1) return;
    const uint32_t g = b0 + pos;
    const int none = enc(CL_CHAIN_NEG);
    uint32_t ins = 0xFFFFFFFFu, off = 0xFFFFFFFFu, kb = 0xFFFFFFFFu;   // padding: beyond every real key
    int32_t sg = 0;
    if (pos < cb.n_recs) {
        ins = cb.ins_t[pos]; off = cb.off[pos]; sg = cb.sigma[pos];
        kb = (((uint32_t)(sg + sig_bias) >> band_shift) << off_bits) | off;
    }
    int4* r = reinterpret_cast<int4*>(rec + (size_t)g * 12);
    r[0] = make_int4((int)ins, (int)off, sg, none);
    r[1] = make_int4(none, none, none, none);
    r[2] = make_int4(none, none, 0, 0);
    key_off[g] = off;
    key_band[g] = kb;
    idx[g] = g;
}

__global__ void __launch_bounds__(256) far_node_key_kernel(const uint32_t* order, uint32_t n, uint32_t shift, uint32_t* node_key) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) node_key[i] = order[i] >> shift;
}

// a table entry (ClFarDevice::tab) is a position in the arena in BLOCKS OF 8 WORDS: every structure starts on a 32-byte block, and 32 bits of blocks reach 2^35 words
// (the arena of a 50-sequence root — 625 combinations, 690 M records — is 1.2 x 10^10 words; with word offsets the far pass was off there: round 5)
template <class T> __device__ __forceinline__ T* far_at(T* arena, uint32_t blocks) { return arena + ((size_t)blocks << 3); }

// the keys of one order of one level into their search tree: position i of the node-wise ascending order holds record perm[i]; its key
// goes to the blocked array (8 keys | 8 running maxima per block) and, when i is a multiple of 8^j, to index array j
__global__ void __launch_bounds__(256) far_layout_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ key, uint32_t n, uint32_t* arena, uint32_t ord_off,
                                                         uint32_t ix0, uint32_t ix1, uint32_t ix2, uint32_t ix3, uint32_t n_ix) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = key[perm[i]];
    // (ord_off, ix*: table entries, in blocks of 8 words — far_at)
    far_at(arena, ord_off)[((i >> 3) << 4) + (i & 7u)] = k;
    const uint32_t ix[4] = {ix0, ix1, ix2, ix3};
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (j < n_ix && (i & ((8u << (3 * j)) - 1u)) == 0) far_at(arena, ix[j])[i >> (3 * (j + 1))] = k;
}

// ---- sealing: running maxima of the DP value in the node's two orders ----------------------------------------------------
// one wave per node
__global__ void __launch_bounds__(64) far_seal_kernel(ClFarDevice F, const int* __restrict__ rec, const uint32_t* __restrict__ items, uint32_t item0) {
    const uint32_t it = items[item0 + blockIdx.x];
    const uint32_t l = it >> 28, node = it & 0x0FFFFFFFu;
    const uint32_t shift = kFarLeafShift + kFarFanShift * l;
    const uint32_t g0 = node << shift, n = 1u << shift;
    const uint32_t lane = threadIdx.x;
    const uint32_t* __restrict__ perm_o = F.perm_o[l];
    const uint32_t* __restrict__ perm_b = F.perm_b[l];
    uint32_t* ord_o = far_at(F.arena, F.tab[l][0]);
    const bool banded = F.tab[l][1] != 0xFFFFFFFFu;   // sparse_chain_dp: one order only
    uint32_t* ord_b = far_at(F.arena, banded ? F.tab[l][1] : 0u);
    int carry_o = INT32_MIN, carry_b = INT32_MIN;
    uint32_t carry_bucket = 0xFFFFFFFEu;
    // the values come through two dependent gathers (order -> record -> dp): U chunks of 64 are fetched together so that a large node pays
    // one such round trip per U chunks, not per chunk (a node of 32 768 records: 512 chunks, 1-2.5 ms before, on the path of the far pass)
    constexpr uint32_t U = 8;
    for (uint32_t at0 = 0; at0 < n; at0 += 64 * U) {
        int xs[U], ys[U];
        uint32_t bks[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t at = at0 + 64 * u;
            if (at < n) {
                const uint32_t i = g0 + at + lane;
                xs[u] = rec[(size_t)perm_o[i] * 12 + 3];
                if (banded) { ys[u] = rec[(size_t)perm_b[i] * 12 + 3]; bks[u] = ord_b[((i >> 3) << 4) + (i & 7u)] >> F.off_bits; }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t at = at0 + 64 * u;
            if (at >= n) break;
            const uint32_t i = g0 + at + lane;
            const uint32_t slot = ((i >> 3) << 4) + 8u + (i & 7u);
            // by offset: plain inclusive maximum
            int x = xs[u];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(x, d);
                if ((int)lane >= d) x = max(x, o);
            }
            x = max(x, carry_o);
            ord_o[slot] = (uint32_t)x;
            carry_o = __shfl(x, 63);
            if (!banded) continue;
            // by (bucket, offset): maximum within the bucket
            int y = ys[u];
            const uint32_t bk = bks[u];
            const uint32_t prev = __shfl_up(bk, 1);
            int head = (lane == 0 ? bk != carry_bucket : bk != prev) ? 1 : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int oy = __shfl_up(y, d);
                const int oh = __shfl_up(head, d);
                if ((int)lane >= d) {
                    if (!head) y = max(y, oy);
                    head |= oh;
                }
            }
            if (!head) y = max(y, carry_b);   // the bucket began in an earlier chunk
            ord_b[slot] = (uint32_t)y;
            carry_b = __shfl(y, 63);
            carry_bucket = __shfl(bk, 63);
        }
    }
}

// The same for a LARGE node (4 096 or 32 768 records), one workgroup of 1 024 threads per node (round 4; one wave took up to 0.9 ms for a
// 32 768-record node, and every far launch behind it waited).  The node is swept in tiles of 8 192 records: a thread takes eight consecutive
// positions of both orders — their record numbers and bucket keys are contiguous, the DP values come through sixteen gathers that are all
// in flight together — scans them, the thread totals are scanned across the wave by shuffles and across the sixteen waves through LDS, and
// two uniform carries go from tile to tile.  The bucket order is a SEGMENTED maximum; a segment starts where the bucket changes, which a
// position reads off the static keys (its predecessor's key is in memory whatever thread owns it), so the scan operator is
// (f1, v1) + (f2, v2) = (f1 | f2, f2 ? v2 : max(v1, v2)) on (a segment starts inside, maximum of the trailing segment).
__global__ void __launch_bounds__(1024) far_seal_big_kernel(ClFarDevice F, const int* __restrict__ rec, const uint32_t* __restrict__ items, uint32_t item0) {
    const uint32_t it = items[item0 + blockIdx.x];
    const uint32_t l = it >> 28, node = it & 0x0FFFFFFFu;
    const uint32_t shift = kFarLeafShift + kFarFanShift * l;
    const uint32_t g0 = node << shift, n = 1u << shift;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t* __restrict__ perm_o = F.perm_o[l];
    const uint32_t* __restrict__ perm_b = F.perm_b[l];
    uint32_t* ord_o = far_at(F.arena, F.tab[l][0]);
    const bool banded = F.tab[l][1] != 0xFFFFFFFFu;   // sparse_chain_dp: one order only
    uint32_t* ord_b = far_at(F.arena, banded ? F.tab[l][1] : 0u);
    __shared__ int s_wx[16], s_wy[16], s_wf[16], s_carry[2];
    int carry_o = INT32_MIN, carry_b = INT32_MIN;     // by offset: maximum so far; by bucket: maximum of the segment that is open at the tile's start
    for (uint32_t tile = 0; tile < n; tile += 8192) {
        const uint32_t i0 = g0 + tile + t * 8;        // this thread's eight positions (n is a multiple of 8 192 or equals 4 096: t * 8 < n decides)
        const bool in = tile + t * 8 < n;
        int x[8], y[8];
        uint32_t bk[8], bprev = 0xFFFFFFFEu;
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = INT32_MIN; y[e] = INT32_MIN; bk[e] = 0xFFFFFFFFu; }
        if (in) {
            const uint4 pa = *reinterpret_cast<const uint4*>(perm_o + i0), pb = *reinterpret_cast<const uint4*>(perm_o + i0 + 4);
            const uint32_t po[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = rec[(size_t)po[e] * 12 + 3];
            if (banded) {
                const uint4 qa = *reinterpret_cast<const uint4*>(perm_b + i0), qb = *reinterpret_cast<const uint4*>(perm_b + i0 + 4);
                const uint32_t pq[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = rec[(size_t)pq[e] * 12 + 3];
                const uint32_t* kb = ord_b + ((i0 >> 3) << 4);
                const uint4 ka = *reinterpret_cast<const uint4*>(kb), kc = *reinterpret_cast<const uint4*>(kb + 4);
                const uint32_t kk[8] = {ka.x, ka.y, ka.z, ka.w, kc.x, kc.y, kc.z, kc.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) bk[e] = kk[e] >> F.off_bits;
                // the position in front of this thread's first one (the node's first position starts a segment whatever lies in front of the node)
                if (i0 != g0) bprev = ord_b[(((i0 - 1) >> 3) << 4) + ((i0 - 1) & 7u)] >> F.off_bits;
            }
        }
        // thread-local inclusive scans; the bucket order's: fy = a segment starts at or after the thread's first position, vy = maximum of the trailing segment
#pragma unroll
        for (int e = 1; e < 8; ++e) x[e] = max(x[e], x[e - 1]);
        int fy = 0;
        if (banded) {
            fy = bk[0] != bprev ? 1 : 0;
#pragma unroll
            for (int e = 1; e < 8; ++e) {
                const bool head = bk[e] != bk[e - 1];
                if (!head) y[e] = max(y[e], y[e - 1]);
                fy |= head ? 1 : 0;
            }
        }
        // exclusive scan of the thread totals over the workgroup: in the wave by shuffles ...
        int tx = x[7], ty = y[7], tf = fy;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ox = __shfl_up(tx, d), oy = __shfl_up(ty, d), of = __shfl_up(tf, d);
            if ((int)lane >= d) {
                tx = max(tx, ox);
                if (!tf) ty = max(ty, oy);
                tf |= of;
            }
        }
        if (lane == 63) { s_wx[wave] = tx; s_wy[wave] = ty; s_wf[wave] = tf; }
        __syncthreads();
        // ... the waves before this one (and the tile's carries) ...
        int px = carry_o, py = carry_b, pf = 0;
        for (uint32_t w = 0; w < wave; ++w) {
            px = max(px, s_wx[w]);
            if (s_wf[w]) { py = s_wy[w]; pf = 1; } else py = max(py, s_wy[w]);
        }
        // ... then the lanes before this one in its wave
        const int lx = __shfl_up(tx, 1), ly = __shfl_up(ty, 1), lf = __shfl_up(tf, 1);
        if (lane > 0) {
            px = max(px, lx);
            if (lf) { py = ly; pf = 1; } else py = max(py, ly);
        }
        (void)pf;
        if (in) {
            uint32_t ox[8], oy[8];
            bool open = true;                          // still inside the segment that was open in front of this thread
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ox[e] = (uint32_t)max(x[e], px);
                if (banded) {
                    if (bk[e] != (e == 0 ? bprev : bk[e - 1])) open = false;
                    oy[e] = (uint32_t)(open ? max(y[e], py) : y[e]);
                }
            }
            uint32_t* wo = ord_o + ((i0 >> 3) << 4) + 8;
            *reinterpret_cast<uint4*>(wo) = make_uint4(ox[0], ox[1], ox[2], ox[3]);
            *reinterpret_cast<uint4*>(wo + 4) = make_uint4(ox[4], ox[5], ox[6], ox[7]);
            if (banded) {
                uint32_t* wb = ord_b + ((i0 >> 3) << 4) + 8;
                *reinterpret_cast<uint4*>(wb) = make_uint4(oy[0], oy[1], oy[2], oy[3]);
                *reinterpret_cast<uint4*>(wb + 4) = make_uint4(oy[4], oy[5], oy[6], oy[7]);
            }
            if (t == 1023 || tile + (t + 1) * 8 >= n) {   // the tile's last thread hands its inclusive totals on
                s_carry[0] = (int)ox[7];
                s_carry[1] = banded ? (int)oy[7] : INT32_MIN;
            }
        }
        __syncthreads();
        carry_o = s_carry[0];
        carry_b = s_carry[1];
        __syncthreads();
    }
}

// ---- the pass -------------------------------------------------------------------------------------------------------------
// stack entries per query: one round of cover nodes (<= W) + per level 7 more for each of the G entries opened per round (W + 21 G: 29 / 58 / 116)
template <int G> struct FarStack { static constexpr uint32_t n = G == 1 ? 80u : G == 2 ? 96u : 160u; };
constexpr uint32_t kFarCoverCache = 12;   // rounds of eight cover nodes whose bounds the probe leaves in LDS for the pass proper (96 nodes: 2.7 M records)

struct FarQuery {
    uint32_t qt, qoff, bq;
    int32_t q;
    float w;
    double slack_q;
};

struct Blk8 { uint4 a, b; };
__device__ __forceinline__ Blk8 load8(const uint32_t* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    return Blk8{q[0], q[1]};
}
__device__ __forceinline__ uint32_t count_lt(const Blk8& k, uint32_t x) {
    return (k.a.x < x ? 1u : 0u) + (k.a.y < x ? 1u : 0u) + (k.a.z < x ? 1u : 0u) + (k.a.w < x ? 1u : 0u) +
           (k.b.x < x ? 1u : 0u) + (k.b.y < x ? 1u : 0u) + (k.b.z < x ? 1u : 0u) + (k.b.w < x ? 1u : 0u);
}
// the largest running maximum among the block's positions whose key k satisfies lo <= k < hi (as k - lo < hi - lo, unsigned); the running
// maxima do not decrease along such a run, so this is the running maximum at the LAST such position; INT32_MIN when there is none
__device__ __forceinline__ int max_in(const Blk8& k, const Blk8& m, uint32_t lo, uint32_t span) {
    int r = INT32_MIN;
    r = max(r, k.a.x - lo < span ? (int)m.a.x : INT32_MIN); r = max(r, k.a.y - lo < span ? (int)m.a.y : INT32_MIN);
    r = max(r, k.a.z - lo < span ? (int)m.a.z : INT32_MIN); r = max(r, k.a.w - lo < span ? (int)m.a.w : INT32_MIN);
    r = max(r, k.b.x - lo < span ? (int)m.b.x : INT32_MIN); r = max(r, k.b.y - lo < span ? (int)m.b.y : INT32_MIN);
    r = max(r, k.b.z - lo < span ? (int)m.b.z : INT32_MIN); r = max(r, k.b.w - lo < span ? (int)m.b.w : INT32_MIN);
    return r;
}

// upper bound of every candidate the node (level lvl, first record g0 of the global image) can give the query; -inf if no
// record of the node lies below the query's offset.  `tab` = ClFarDevice::tab in LDS.
template <bool SPARSE>
__device__ __forceinline__ double node_bound(const ClFarDevice& F, const uint32_t* tab, uint32_t lvl, uint32_t g0, const FarQuery& Q) {
    const uint32_t* __restrict__ A = F.arena;
    const uint32_t* t = tab + lvl * kFarTabWidth;
    // the searches run in lockstep (their loads overlap): offset < qoff in the offset order; key < (b << off_bits | qoff) for the three
    // buckets b around the query's in the bucket order.  A bucket no record can have searches for key 0.
    uint32_t xb[3], xlo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t bk = Q.bq + (uint32_t)(j - 1);
        const bool real = !SPARSE && bk < (1u << (32 - F.off_bits)) - 1u && bk != 0;
        xlo[j] = real ? bk << F.off_bits : 0u;
        xb[j] = real ? xlo[j] + Q.qoff : 0u;   // qoff <= largest offset + 1 < 2^off_bits
    }
    uint32_t po = g0 >> (3 * (lvl + 1)), p0 = po, p1 = po, p2 = po;   // entry numbers in the current index array: the node's top block
    bool none = false;
    for (uint32_t j = lvl + 1; j >= 1; --j) {
        const uint32_t* io = far_at(A, t[2 + (j - 1)]);
        const Blk8 eo = load8(io + po);
        uint32_t co, c0 = 1, c1 = 1, c2 = 1;
        if (!SPARSE) {
            const uint32_t* ib = far_at(A, t[2 + kFarMaxLevels + (j - 1)]);
            const Blk8 e0 = load8(ib + p0), e1 = load8(ib + p1), e2 = load8(ib + p2);
            c0 = count_lt(e0, xb[0]); c1 = count_lt(e1, xb[1]); c2 = count_lt(e2, xb[2]);
        }
        co = count_lt(eo, Q.qoff);
        if (j == lvl + 1) none = co == 0;      // below the top block a followed entry is itself below the bound
        po = (po + (co ? co - 1 : 0u)) << 3;
        p0 = (p0 + (c0 ? c0 - 1 : 0u)) << 3; p1 = (p1 + (c1 ? c1 - 1 : 0u)) << 3; p2 = (p2 + (c2 ? c2 - 1 : 0u)) << 3;
    }
    if (none) return -HUGE_VAL;
    // po .. p2 are positions (multiples of 8) in the ascending orders: blocks of 16 words
    const uint32_t* oo = far_at(A, t[0]) + 2 * (size_t)po;
    const Blk8 ko = load8(oo), mo = load8(oo + 8);
    int dband = INT32_MIN;
    if (!SPARSE) {
        const uint32_t* ob = far_at(A, t[1]);
        const Blk8 k0 = load8(ob + 2 * (size_t)p0), m0 = load8(ob + 2 * (size_t)p0 + 8);
        const Blk8 k1 = load8(ob + 2 * (size_t)p1), m1 = load8(ob + 2 * (size_t)p1 + 8);
        const Blk8 k2 = load8(ob + 2 * (size_t)p2), m2 = load8(ob + 2 * (size_t)p2 + 8);
        dband = max(max_in(k0, m0, xlo[0], xb[0] - xlo[0]), max(max_in(k1, m1, xlo[1], xb[1] - xlo[1]), max_in(k2, m2, xlo[2], xb[2] - xlo[2])));
    }
    const int pa = max_in(ko, mo, 0u, Q.qoff);
    if (pa == INT32_MIN) return -HUGE_VAL;
    const double da = (double)dec(pa);
    if (SPARSE) return da + (double)Q.w + 0x1p-21 * (fabs(da) + fabs((double)Q.w));
    double m = da - F.band_pen;
    if (dband != INT32_MIN) m = fmax(m, (double)dec(dband));
    return m + (double)Q.w + 0x1p-21 * (fabs(da) + Q.slack_q);
}

// the exact evaluation of the sweep over the 64 records of a leaf, by the eight lanes of the query's group: lane `sub` takes records sub, sub + 8, ...
template <bool SPARSE>
__device__ __forceinline__ void scan_leaf(const int* __restrict__ rec, uint32_t g0, uint32_t sub, const FarQuery& Q, int (&acc)[7]) {
    const int4* r4 = reinterpret_cast<const int4*>(rec + (size_t)g0 * 12);
    if (SPARSE) {
        int4 ra[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[u] = r4[(sub + 8 * u) * 3];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[0] = max(acc[0], ((uint32_t)ra[u].x <= Q.qt && (uint32_t)ra[u].y < Q.qoff) ? ra[u].w : INT32_MIN);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int4 ra[4], rb[4];
            int2 rc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = sub + 8 * (4 * h + u);
                ra[u] = r4[i * 3]; rb[u] = r4[i * 3 + 1]; rc[u] = reinterpret_cast<const int2*>(r4 + i * 3 + 2)[0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v[7] = {ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w, rc[u].x, rc[u].y};
                accumulate(acc, Q.qt, Q.qoff, Q.q, (uint32_t)ra[u].x, (uint32_t)ra[u].y, ra[u].z, v);
            }
        }
    }
}

template <bool SPARSE>
__device__ __forceinline__ float best_of(float best, const int (&acc)[7], const FarQuery& Q, const double (&pen)[6]) {
    if (SPARSE) {
        if (acc[0] > enc(CL_CHAIN_NEG)) best = fmaxf(best, dec(acc[0]) + Q.w);
    } else {
        best = best_candidate(best, acc, Q.w, pen);
    }
    best = fmaxf(best, __shfl_xor(best, 1));
    best = fmaxf(best, __shfl_xor(best, 2));
    best = fmaxf(best, __shfl_xor(best, 4));
    return best;
}

// this lane's share of the next W cover nodes of [0, p) (lane li of the query's W): aligned nodes, the nearest (smallest) first
template <int W>
__device__ __forceinline__ void next_cover(uint32_t& p, uint32_t li, uint32_t top, uint32_t& lvl, uint32_t& a) {
    uint32_t pp = p;
    lvl = 0xFFu;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)W; ++j) {
        if (pp == 0) break;
        uint32_t l = (uint32_t)(__builtin_ctz(pp >> kFarLeafShift)) / kFarFanShift;
        if (l > top) l = top;
        const uint32_t nn = 1u << (kFarLeafShift + kFarFanShift * l);
        if (j == li) { lvl = l; a = pp - nn; }
        pp -= nn;
    }
    p = pp;
}

// the node with the largest bound among the W lanes of the query (ties: the smaller code), the same answer in every lane
template <int W>
__device__ __forceinline__ void group_argmax(double& b, uint32_t& code) {
#pragma unroll
    for (int m = 1; m < W; m <<= 1) {
        const double ob = __shfl_xor(b, m);
        const uint32_t oc = __shfl_xor(code, m);
        if (ob > b || (ob == b && oc < code)) { b = ob; code = oc; }
    }
}

template <int W>
__device__ __forceinline__ float max_over_query(float x) {
#pragma unroll
    for (int m = 8; m < W; m <<= 1) x = fmaxf(x, __shfl_xor(x, m));
    return x;
}

// G groups of eight lanes work on one query (W = 8 G lanes): the cover nodes are bounded W at a time, G stack entries are opened per round (eight
// children each) and G surviving leaves are scanned side by side, each by one group of eight.  A query's search is a chain of dependent
// loads — ≈ 140 of them with eight lanes, whatever the launch holds — so launches with few chain combinations (few queries) take more lanes
// per query: the chain gets shorter and the device has the room.
template <bool SPARSE, int G>
__global__ void __launch_bounds__(256) far_prune_kernel(ClChainDevice D, ClFarDevice F, uint32_t first, uint32_t count, uint32_t end_block) {
    constexpr int W = 8 * G;
    constexpr uint32_t QPW = 256 / W;                            // queries per workgroup
    __shared__ uint32_t s_tab[kFarMaxLevels * kFarTabWidth];
    if (threadIdx.x < kFarMaxLevels * kFarTabWidth) s_tab[threadIdx.x] = F.tab_dev[threadIdx.x];
    __syncthreads();
    uint32_t by = blockIdx.y, bx = blockIdx.x;
    if (F.xcd_x) {   // (see ClFarDevice: combination by XCD for the full rounds of eight combinations, the rest dealt as before)
        const uint32_t full = (F.xcd_n & ~7u) * F.xcd_x;
        if (blockIdx.x < full) {
            const uint32_t j = blockIdx.x >> 3;
            by = (blockIdx.x & 7u) + 8u * (j / F.xcd_x);
            bx = j % F.xcd_x;
        } else {
            const uint32_t r = blockIdx.x - full;
            by = (F.xcd_n & ~7u) + r / F.xcd_x;
            bx = r % F.xcd_x;
        }
    }
    const uint32_t c = F.share_n > 1 ? F.share_i + by * F.share_n : by;   // (a shared far pass: this member's combinations)
    const ClChainCombo cb = D.combos[c];
    const uint32_t sub = threadIdx.x & 7u;                       // lane within its group of eight
    const uint32_t li = threadIdx.x & (W - 1u);                  // lane within the query's W
    const uint32_t sg = li >> 3;                                 // group of eight within the query
    const uint32_t grp = threadIdx.x / W;                        // query within the workgroup
    const uint32_t qi = bx * QPW + grp;
    const uint32_t s = first + qi;
    const uint32_t E = cb.prefix[end_block] & ~63u;              // records [0, E) are final, every node inside is sealed
    const int none = enc(CL_CHAIN_NEG);
    bool live = qi < count && E != 0;
    FarQuery Q{};
    if (live) {
        Q.qt = cb.qt[s];
        live = Q.qt != 0xFFFFFFFFu;
        if (live) { Q.qoff = cb.qoff[s]; Q.q = cb.q[s]; live = Q.qoff != 0; }
    }
    if (!live) {                                                 // uniform over the query's lanes
        if (F.share_n > 1 && li < 7 && qi < count)                // the other members read every query of this combination: "nothing found"
            for (uint32_t p = 0; p + 1 < F.share_n; ++p)
                if (F.peer_out[p]) F.peer_out[p][((size_t)c * kChainMacro + qi) * 7 + li] = enc(CL_CHAIN_NEG);
        return;
    }
    Q.w = D.weight[s];
    float best = D.init[s];
    int acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = INT32_MIN;
    double pen[6];
#pragma unroll
    for (int pw = 0; pw < 6; ++pw) {
        const double go = D.params.gap_open[pw / 2], ge = D.params.gap_extend[pw / 2];
        pen[pw] = (pw % 2 == 1) ? D.params.scale * (go + ge * (double)Q.q) : D.params.scale * (go - ge * (double)Q.q);
    }
    const uint32_t base = D.far_base[c];
    const int* __restrict__ rec = D.far_rec;
    Q.slack_q = F.slack_t0 + F.slack_e0 * fabs((double)Q.q) + fabs((double)Q.w);
    // shift bucket of the query; the host picks sig_bias so that every record's bucket is >= 1: a query whose biased shift is negative is
    // more than a bucket width away from every record
    const long long qb = (long long)Q.q + (long long)F.sig_bias;
    Q.bq = qb < 0 ? 0xFFFF0000u : (uint32_t)(qb >> F.band_shift);
    const uint32_t top = F.n_levels - 1;
    const uint32_t gsh = (threadIdx.x & 63u) & ~(uint32_t)(W - 1);   // where the query's W lanes sit in a ballot
    const unsigned long long wmask = W == 64 ? ~0ull : ((1ull << W) - 1ull);

    // bounds of the cover nodes, computed once by the probe: they depend on the query alone, only what they are compared with changes
    __shared__ double s_cover[QPW][kFarCoverCache][W];
    uint32_t n_scanned = 0, probe_scans = 0;
    // ---- probe: follow the largest bound down to one leaf.  A query whose best predecessor lies far back (a pair on a distant
    //      diagonal chains from wherever the main chain passed its graph-2 position) would otherwise open every node between
    //      itself and that place before it knows what it is looking for.
    {
        double pb = -HUGE_VAL;
        uint32_t pcode = 0xFFFFFFFFu;
        uint32_t p = E;
        for (uint32_t round = 0; p > 0; ++round) {
            uint32_t lvl, a = 0;
            next_cover<W>(p, li, top, lvl, a);
            if (lvl != 0xFFu) {
                const double b = node_bound<SPARSE>(F, s_tab, lvl, base + a, Q);
                if (round < kFarCoverCache) s_cover[grp][round][li] = b;
                const uint32_t code = (lvl << 28) | (a >> kFarLeafShift);
                if (b > pb || (b == pb && code < pcode)) { pb = b; pcode = code; }
            }
        }
        group_argmax<W>(pb, pcode);
        // the descent: the eight children of the best node, by the first group of eight (the others have nothing to add: one node, eight children)
        while (pcode != 0xFFFFFFFFu && (pcode >> 28) != 0 && pb > -HUGE_VAL) {
            const uint32_t lvl = (pcode >> 28) - 1;
            const uint32_t a = ((pcode & 0x0FFFFFFFu) << kFarLeafShift) + (sub << (kFarLeafShift + kFarFanShift * lvl));
            pb = sg == 0 ? node_bound<SPARSE>(F, s_tab, lvl, base + a, Q) : -HUGE_VAL;
            pcode = sg == 0 ? (lvl << 28) | (a >> kFarLeafShift) : 0xFFFFFFFFu;
            group_argmax<W>(pb, pcode);
        }
        if (pcode != 0xFFFFFFFFu && (pcode >> 28) == 0 && pb > -HUGE_VAL) {   // (uniform over the query's lanes)
            if (sg == 0) scan_leaf<SPARSE>(rec, base + ((pcode & 0x0FFFFFFFu) << kFarLeafShift), sub, Q, acc);
            best = max_over_query<W>(best_of<SPARSE>(best, acc, Q, pen));
            ++probe_scans;
        }
    }

    // ---- the branch-and-bound proper, nearest nodes first
    __shared__ uint32_t s_stack[QPW][FarStack<G>::n];
    volatile uint32_t* st = s_stack[grp];
    uint32_t sp = 0;
    uint32_t p = E;                                              // cover nodes still to hand out lie in [0, p)
    uint32_t cover_round = 0;
    while (true) {
        // this round's node for this lane: (level, first record), level 0xFF = none
        uint32_t lvl = 0xFFu, a = 0;
        bool cached = false;
        if (sp > 0) {
            // the top G entries, one per group of eight (the topmost — nearest — to the first group); a group without an entry sits the round out
            const uint32_t take = sp < (uint32_t)G ? sp : (uint32_t)G;
            if (sg < take) {
                const uint32_t e = st[sp - 1 - sg];
                lvl = (e >> 28) - 1;
                a = ((e & 0x0FFFFFFFu) << kFarLeafShift) + ((7u - sub) << (kFarLeafShift + kFarFanShift * lvl));   // lane 0 takes the nearest child
            }
            sp -= take;
        } else if (p > 0) {
            next_cover<W>(p, li, top, lvl, a);
            cached = cover_round < kFarCoverCache;
            ++cover_round;
        } else {
            break;
        }
        double bnd = -HUGE_VAL;
        if (lvl != 0xFFu) bnd = cached ? s_cover[grp][cover_round - 1][li] : node_bound<SPARSE>(F, s_tab, lvl, base + a, Q);
        bool hit = lvl != 0xFFu && bnd >= (double)best;
        // surviving leaves, nearest (lane 0) first, G at a time — one per group of eight; each is tested again against what the rounds before gave
        unsigned long long leaves = (__ballot(hit && lvl == 0) >> gsh) & wmask;
        while (leaves) {
            uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!leaves) break;
                const uint32_t j = (uint32_t)__builtin_ctzll(leaves);
                leaves &= leaves - 1;
                if ((uint32_t)g == sg) mine = j;
            }
            const double bj = __shfl(bnd, (int)(mine & (W - 1)), W);
            const uint32_t aj = __shfl(a, (int)(mine & (W - 1)), W);
            const bool go = mine != 0xFFFFFFFFu && bj >= (double)best;
            if (go) scan_leaf<SPARSE>(rec, base + aj, sub, Q, acc);
            // (a group that scanned nothing brings its unchanged maxima: best_of is idempotent)
            best = max_over_query<W>(best_of<SPARSE>(best, acc, Q, pen));
            n_scanned += go && sub == 0 ? 1u : 0u;
        }
        // surviving inner nodes go on the stack, nearest (lane 0) on top
        const bool push = hit && lvl != 0 && bnd >= (double)best;
        const unsigned long long mask = (__ballot(push) >> gsh) & wmask;
        if (push) {
            const uint32_t above = __popcll(mask >> (li + 1));   // surviving lanes farther than this one go below it
            st[sp + above] = (lvl << 28) | (a >> kFarLeafShift);
        }
        sp += __popcll(mask);
    }
    // bookkeeping for the host's choice between this pass and the all-pairs sweep (cl_chain_api.cpp): leaves this query scanned
    // against the leaves it had in range.  A scanned leaf costs about what sixteen leaves cost the sweep (divergent loads
    // instead of LDS broadcasts, plus the tests that led to it).
    {
        // total over the query: the probe's scan (the same in every lane) + what lane 0 of every group of eight counted in the pass proper
        uint32_t tot = sub == 0 ? n_scanned : 0u;
#pragma unroll
        for (int m = 8; m < W; m <<= 1) tot += __shfl_xor(tot, m);
        n_scanned = tot + probe_scans;
    }
    if (li == 0) {
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(D.status + 2);
        // only queries with a long history say anything about the trend (early ones open their few leaves whatever happens)
        const unsigned long long add_sc = E >= (1u << 16) ? n_scanned : 0u, add_in = E >= (1u << 16) ? (E >> kFarLeafShift) : 0u;
        if (add_in) { atomicAdd(cnt, add_sc); atomicAdd(cnt + 1, add_in); }
    }
    // merge the lanes' maxima and hand them to the walk
    constexpr int NK = SPARSE ? 1 : 7;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        int v = acc[k];
#pragma unroll
        for (int m = 1; m < W; m <<= 1) v = max(v, __shfl_xor(v, m));
        if (li == 0 && v > none) atomicMax(cb.acc + (size_t)s * 7 + k, v);
        if (F.share_n > 1 && li == 0)
            for (uint32_t p = 0; p + 1 < F.share_n; ++p)
                if (F.peer_out[p]) F.peer_out[p][((size_t)c * kChainMacro + qi) * 7 + k] = v;
    }
    if (F.share_n > 1) __threadfence_system();   // the stores to the other devices are out before the stream's flag write that follows the kernel
}

// what the other members of the merge group found for the queries of THEIR combinations of this macro-block (the slot of this context's
// inbox they stored into) joins this context's running maxima, in front of the walk of the macro-block
__global__ void __launch_bounds__(256) far_merge_kernel(ClChainDevice D, const int* slot, uint32_t first, uint32_t count, uint32_t share_n, uint32_t share_i) {
    const uint32_t c = blockIdx.y;
    if (c % share_n == share_i) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t nk = D.sparse ? 1u : 7u;
    if (i >= count * nk) return;
    const uint32_t q = i / nk, k = i % nk;
    const int v = slot[((size_t)c * kChainMacro + q) * 7 + k];
    if (v > enc(CL_CHAIN_NEG)) atomicMax(D.combos[c].acc + (size_t)(first + q) * 7 + k, v);
}

}  // namespace

// ---- host entry points ---------------------------------------------------------------------------------------------------
hipError_t cl_chain_far_init(const ClChainDevice& D, const uint32_t* d_base, uint32_t max_padded, uint32_t r_pad, int32_t sig_bias, uint32_t band_shift,
                             uint32_t off_bits, uint32_t* key_off, uint32_t* key_band, uint32_t* idx, hipStream_t stream) {
    hipLaunchKernelGGL(far_init_kernel, dim3((max_padded + 255) / 256, D.n_combos), dim3(256), 0, stream, D.combos, d_base, D.n_combos, r_pad,
                       sig_bias, band_shift, off_bits, D.far_rec, key_off, key_band, idx);
    return hipGetLastError();
}

size_t cl_chain_far_sort_temp_bytes(uint32_t n) { return clradix::sort_temp_bytes<uint32_t>(n) + 256; }

// order_out = idx sorted by key (32-bit keys, bits [0, end_bit)); stable
hipError_t cl_chain_far_sort32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                               uint32_t n, int end_bit, hipStream_t stream) {
    return clradix::sort_pairs<uint32_t>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)end_bit, stream);
}

hipError_t cl_chain_far_node_keys(const uint32_t* order, uint32_t n, uint32_t shift, uint32_t* node_key, hipStream_t stream) {
    hipLaunchKernelGGL(far_node_key_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, order, n, shift, node_key);
    return hipGetLastError();
}

// the search tree of one order of level `lvl`: arena[ord_off ...] blocked keys, arena[ix[j] ...] every 8^(j+1)-th key (j = 0 .. lvl)
hipError_t cl_chain_far_layout(const uint32_t* perm, const uint32_t* key, uint32_t n, uint32_t* arena, uint32_t ord_off, const uint32_t* ix, uint32_t n_ix,
                               hipStream_t stream) {
    hipLaunchKernelGGL(far_layout_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, perm, key, n, arena, ord_off, ix[0], ix[1], ix[2], ix[3], n_ix);
    return hipGetLastError();
}

// the first n_big items are nodes of 4 096 records or more: a workgroup each; the others a wave each
// done (may be null): recorded behind the last launch, as part of that launch (see cl_chain_launch_walk2)
hipError_t cl_chain_far_seal(const ClChainDevice& D, const ClFarDevice& F, const uint32_t* items, uint32_t item0, uint32_t n_items, uint32_t n_big, hipStream_t stream, hipEvent_t done) {
    const bool small = n_items > n_big;
    if (n_big) hipExtLaunchKernelGGL(far_seal_big_kernel, dim3(n_big), dim3(1024), 0, stream, nullptr, small ? nullptr : done, 0, F, D.far_rec, items, item0);
    if (small) hipExtLaunchKernelGGL(far_seal_kernel, dim3(n_items - n_big), dim3(64), 0, stream, nullptr, done, 0, F, D.far_rec, items, item0 + n_big);
    else if (!n_big && done) return hipEventRecord(done, stream);
    return hipGetLastError();
}

namespace {
__global__ void peer_store_word_kernel(uint32_t* where, uint32_t value) {
    *where = value;
    __threadfence_system();
}
// cl_context_peer_steal: the next chunk of job `job` from the group's counter word — job << 32 | chunks handed out — which lives in member 0's exported
// memory (another device's, over xGMI, for every other member: system-scope atomics).  The first member to arrive for a job finds an older job in the
// word and replaces it by job << 32 | 1 (it has chunk 0): nobody resets anything, job numbers only grow.  *out = chunk, or 0xFFFFFFFF when the word
// already holds a LATER job (the caller's job number is stale)
__global__ void peer_steal_kernel(unsigned long long* word, uint32_t job, uint32_t* out) {
    unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    while (true) {
        const uint32_t have = (uint32_t)(v >> 32);
        if (have > job) { *out = 0xFFFFFFFFu; break; }
        const unsigned long long want = have == job ? v + 1 : ((unsigned long long)job << 32) | 1ull;
        if (__hip_atomic_compare_exchange_strong(word, &v, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
            *out = have == job ? (uint32_t)v : 0u;
            break;
        }
    }
    __threadfence_system();
}
}
hipError_t cl_peer_steal(unsigned long long* word, uint32_t job, uint32_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(peer_steal_kernel, dim3(1), dim3(1), 0, stream, word, job, out);
    return hipGetLastError();
}
// cl_context_peer_selftest: a kernel's store into (possibly another device's) memory
hipError_t cl_peer_store_word(uint32_t* where, uint32_t value, hipStream_t stream) {
    hipLaunchKernelGGL(peer_store_word_kernel, dim3(1), dim3(1), 0, stream, where, value);
    return hipGetLastError();
}

hipError_t cl_chain_far_merge(const ClChainDevice& D, const int* slot, uint32_t first, uint32_t count, uint32_t share_n, uint32_t share_i, hipStream_t stream) {
    hipLaunchKernelGGL(far_merge_kernel, dim3((count * 7 + 255) / 256, D.n_combos), dim3(256), 0, stream, D, slot, first, count, share_n, share_i);
    return hipGetLastError();
}

hipError_t cl_chain_far_launch(const ClChainDevice& D, const ClFarDevice& F, uint32_t first, uint32_t count, uint32_t end_block, hipStream_t stream, hipEvent_t done) {
    const uint32_t mine = F.share_n > 1 ? (D.n_combos > F.share_i ? (D.n_combos - F.share_i + F.share_n - 1) / F.share_n : 0u) : D.n_combos;
    if (mine == 0) return done ? hipEventRecord(done, stream) : hipSuccess;
    // lanes per query (CL_CHAIN_FAR_LANES = 8 / 16 / 32 pins it): the fewer combinations a launch holds, the more lanes a query gets
    static const int pinned = [] { const char* e = getenv("CL_CHAIN_FAR_LANES"); return e ? atoi(e) : 0; }();
    // (10 x 1 Mbp, device time of a merge's two DPs with 8 / 16 / 32 lanes: 1 combination 348 / 303 / 293 ms, 4 combinations 583 / 512 / 476 ms, 25 combinations
    // 1 193 / 1 115 / 1 090 ms)
    const int lanes = (pinned == 8 || pinned == 16 || pinned == 32) ? pinned : (mine <= 32 ? 32 : mine <= 128 ? 16 : 8);   // (a whole wave per query was tried: it does not terminate)
    dim3 grid((count * (uint32_t)lanes + 255) / 256, mine);
    ClFarDevice Fx = F;
    // workgroup w of a 1-D grid goes to XCD w % 8: with eight and more combinations every full round of eight is pinned, combination c to XCD c % 8, so that an
    // XCD's L2 holds the search structures of an eighth of the combinations (10 x 1 Mbp root, 25 combinations: HBM fetches per far launch 1 / 3, device time of the
    // merge's DPs 874 -> 835 ms).  CL_CHAIN_FAR_XCD=0: the 2-D grid of rounds 2-4 (A/B)
    static const bool xcd_env = [] { const char* e = getenv("CL_CHAIN_FAR_XCD"); return !e || e[0] != '0'; }();
    if (xcd_env && mine >= 8) {
        Fx.xcd_x = grid.x; Fx.xcd_n = mine;
        grid = dim3(mine * grid.x, 1);
    } else { Fx.xcd_x = 0; Fx.xcd_n = 0; }
#define CL_FAR_LAUNCH(G) do { \
        if (D.sparse) hipExtLaunchKernelGGL((far_prune_kernel<true, G>), grid, dim3(256), 0, stream, nullptr, done, 0, D, Fx, first, count, end_block); \
        else hipExtLaunchKernelGGL((far_prune_kernel<false, G>), grid, dim3(256), 0, stream, nullptr, done, 0, D, Fx, first, count, end_block); } while (0)
    if (lanes >= 32) CL_FAR_LAUNCH(4);
    else if (lanes >= 16) CL_FAR_LAUNCH(2);
    else CL_FAR_LAUNCH(1);
#undef CL_FAR_LAUNCH
    return hipGetLastError();
}
