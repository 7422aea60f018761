// chain_walk2.hip — the sequential part of the chaining DP (include/centrolign/anchorer.hpp:2290-2416, sparse: :1640-1700) for one macro-block of
// kChainMacro match pairs, spread over SEVERAL compute units per chain combination (round 4; chain_walk_kernel of chain_kernels.hip keeps the
// whole macro-block on one).
//
// What is serial in the walk is short: a pair's DP value needs the records of every pair that can precede it, and pairs are finalised group by
// group (maximal runs of pairs none of which can precede another, ~45 pairs).  What is heavy is not serial: a record must reach EVERY later
// query of the macro-block (1024 x 1024 / 2 record x query evaluations of ~20 operations, all on one compute unit until round 3: 102 us of
// instruction issue per macro-block).  Here a combination's walk is one MAIN workgroup and up to six HELPER workgroups:
//
//   main    walks the groups.  It keeps only a WINDOW of the next 128 x QPT queries in registers — eight lanes per query, each lane taking
//           every eighth record of a group out of LDS, so a group of 45 records costs a lane 6 evaluations per query instead of 45 — and
//           publishes every finalised DP value as one 8-byte {pair tag, value} granule (write-through store, no fence, no flag:
//           cdna_hip_programming.md §6 Guideline 16, recipe R2).  The eight lanes of a query also share the finalisation: after a butterfly
//           maximum over the lanes (three DPP steps per kind, no LDS) lane k evaluates the candidate of tree kind k and stores the value
//           tree kind k keeps for the pair (anchorer.hpp:2318-2342), so the dependent chain of a group is one candidate long, not seven.
//   helper  owns the queries of a few sub-blocks of 32 pairs beyond the window.  It follows the main workgroup's granules, rebuilds the
//           records (a record is a function of the DP value and static fields), and gives its queries every record that was final before
//           they entered the main workgroup's window; the maxima go back as seven granules per query.  These are order-free maxima of real
//           candidates: the main workgroup takes them when they are there, and otherwise — a helper that is late, not resident, or absent —
//           evaluates the same records itself out of LDS, where it keeps every record of the macro-block.  It never waits for a helper, so
//           nothing here needs the workgroups to be resident together (the exchange BETWEEN combinations still does: chain_walk_kernel's rule).
//
// Coverage (checked exhaustively on a model, scripts/dev/walk2_model.py): a query of sub-block m (32 pairs) enters the window when the walk
// reaches sub-block m - BW + 1 (BW = window / 32); the main workgroup gives it every record finalised from that step on, its helper every
// record of the sub-blocks <= m - BW, all of which are final by then.  Records seen twice are harmless (maxima).
//
// Compiled with -ffp-contract=off like chain_kernels.hip: candidates and stored values must round like the reference's scalar code.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include "device_once.h"

#include "chain_device.h"

namespace {

// pointers of the walk's loops are GLOBAL (address space 1), never generic: a pending flat_ store makes the compiler wait for vmcnt(0) at every
// later wait (flat operations may return out of order), i.e. for every store of the step
typedef __attribute__((address_space(1))) int g_i32;
typedef __attribute__((address_space(1))) float g_f32;
typedef __attribute__((address_space(1))) unsigned int g_u32;
typedef __attribute__((address_space(1))) unsigned long long g_u64;
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4i32 g_i32x4;

constexpr uint32_t kSub = 32;                       // pairs per sub-block: the unit the window slides by and helpers own
constexpr uint32_t kSlots = kChainMacro / 8;        // queries a main workgroup holds per QPT (eight lanes each)
constexpr uint32_t kPollBatch = 64;                 // granules a helper's polling wave reads at a time
constexpr uint32_t kHelperSubs = kSlots / kSub;     // sub-blocks a helper owns (128 queries x 8 lanes)

__device__ __forceinline__ int enc(float f) {
    int b = __float_as_int(f);
    if (b == (int)0x80000000) b = 0;
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Values that are loaded once (vector memory) and used inside the walk's loop: the compiler's wait-count bookkeeping loses track of them round the
// loop's back edge and puts an s_waitcnt vmcnt in front of their first use in EVERY iteration — which waits for the iteration's stores.  Taken
// through a scalar register (uniform values) or an empty asm (per-lane values) after the prologue's barrier they are no longer "loaded" values.
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
template <class T> __device__ __forceinline__ T uni_ptr(T p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned long long r = ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
    return (T)r;
}
__device__ __forceinline__ double uni_f64(double x) {
    const unsigned long long v = (unsigned long long)__double_as_longlong(x);
    return __longlong_as_double((long long)(((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v)));
}
__device__ __forceinline__ double settled(double x) { asm volatile("" : "+v"(x)); return x; }

// tag of a helper's granule: the macro-block (never 0, so zeroed memory never matches) and how many of the macro-block's records [0, cov) the value covers
__device__ __forceinline__ uint32_t cover_tag(uint32_t block, uint32_t cov) { return ((block % 0x1FFFFFu) + 1u) << 11 | cov; }

// maximum over the eight lanes of a query (aligned groups of eight lanes), the same in all of them: quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror (lane i <- lane 7 - i of its half row)
__device__ __forceinline__ int max8_i(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ uint32_t min8_u(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ float max8_f(float f) {
    int v = __float_as_int(f);
    f = fmaxf(f, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false)));
    v = __float_as_int(f);
    f = fmaxf(f, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false)));
    v = __float_as_int(f);
    f = fmaxf(f, __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false)));
    return f;
}

__device__ __forceinline__ void accumulate7(int (&acc)[7], uint32_t qt, uint32_t qoff, int32_t q, const int4& a, const int4& b, const int4& c) {
    const int none = INT32_MIN;
    const bool ok = (uint32_t)a.x <= qt && (uint32_t)a.y < qoff;
    const bool eq = ok && a.z == q, lt = ok && a.z < q, gt = ok && a.z > q;
    acc[0] = max(acc[0], eq ? a.w : none);
    // odd trees: shift < query (anchorer.hpp:2328-2331, 2394-2403); even trees: shift > query (:2332-2335, 2404-2412)
    acc[2] = max(acc[2], lt ? b.y : none); acc[4] = max(acc[4], lt ? b.w : none); acc[6] = max(acc[6], lt ? c.y : none);
    acc[1] = max(acc[1], gt ? b.x : none); acc[3] = max(acc[3], gt ? b.z : none); acc[5] = max(acc[5], gt ? c.x : none);
}

// the value tree kind k keeps for a pair with DP value `best` (anchorer.hpp:2318-2342): k = 0 the value itself; k = 1 + pw: best -+ scale * extend * shift
__device__ __forceinline__ float stored_value(float best, uint32_t k, double tt) {
    if (k == 0) return best;
    return ((k - 1u) % 2u == 1u) ? (float)((double)best + tt) : (float)((double)best - tt);
}

template <bool SPARSE> struct W2Lds {
    static constexpr uint32_t RW = SPARSE ? 4 : 12;           // words per record in LDS
    static constexpr uint32_t NK = SPARSE ? 1 : 7;
    static constexpr uint32_t AW = SPARSE ? 1 : 8;            // words per query of the initial maxima (padded)
    static constexpr size_t rec = 0;
    static constexpr size_t q = rec + (size_t)kChainMacro * RW * 4;
    static constexpr size_t st = q + (size_t)kChainMacro * 16;
    static constexpr size_t w = st + (size_t)kChainMacro * 16;
    static constexpr size_t acc = w + (size_t)kChainMacro * 8;
    static constexpr size_t flags = acc + (size_t)kChainMacro * AW * 4;
    static constexpr size_t main_bytes = flags + 16;
    // helper: static record fields of every pair, two record buffers, two counts
    static constexpr size_t h_stat = 0;
    static constexpr size_t h_buf = (size_t)kChainMacro * 16;
    static constexpr size_t h_n = h_buf + 2 * (size_t)kPollBatch * RW * 4;
    static constexpr size_t helper_bytes = h_n + 16;
    static constexpr size_t bytes = main_bytes > helper_bytes ? main_bytes : helper_bytes;
};

// ---- the main workgroup of combination c -----------------------------------------------------------------------------------------------
template <bool SPARSE, int QPT>
__device__ __forceinline__ void walk2_main(const ClChainDevice& D, const uint32_t first, const uint32_t count, const uint32_t c, const bool helped, char* smem) {
    using L = W2Lds<SPARSE>;
    constexpr uint32_t RW = L::RW, NK = L::NK, AW = L::AW;
    constexpr uint32_t WIN = kSlots * QPT, BW = WIN / kSub;
    int* s_rec = reinterpret_cast<int*>(smem + L::rec);
    int4* s_q = reinterpret_cast<int4*>(smem + L::q);        // qt, qoff (0: no query), q, position of the pair's record in this combination (~0: none)
    int4* s_st = reinterpret_cast<int4*>(smem + L::st);      // ins, off, shift of that record, end of the pair's group (index in the macro-block)
    float2* s_w = reinterpret_cast<float2*>(smem + L::w);    // weight, value of the chain that starts at the pair
    int* s_acc = reinterpret_cast<int*>(smem + L::acc);      // what the far and near passes found (the running maxima when the launch starts)
    int* s_flags = reinterpret_cast<int*>(smem + L::flags);  // [0] abort

    const unsigned long long t_entry = D.debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long c_entry = D.debug ? __builtin_amdgcn_s_memtime() : 0ull;
    const ClChainCombo cb = D.combos[c];
    const uint32_t t = threadIdx.x;
    const int none = enc(CL_CHAIN_NEG);
    const uint32_t n_combos = uni(D.n_combos);
    {   // stage the macro-block's queries: one pair per thread, coalesced
        int4 a = make_int4(-1, 0, 0, -1), b = make_int4(-1, -1, 0, (int)count);
        float2 ww = make_float2(0.f, CL_CHAIN_NEG);
        if (t < count) {
            const uint32_t s = first + t;
            const uint32_t qt = cb.qt[s];
            if (qt != 0xFFFFFFFFu) { a.x = (int)qt; a.y = (int)cb.qoff[s]; a.z = cb.q[s]; }
            const uint32_t pos = cb.own_rec[s];
            a.w = (int)pos;
            if (pos != 0xFFFFFFFFu) { b.x = (int)cb.ins_t[pos]; b.y = (int)cb.off[pos]; b.z = cb.sigma[pos]; }
            b.w = (int)(min(D.group_end[s], first + count) - first);
            ww = make_float2(D.weight[s], D.init[s]);
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) s_acc[t * AW + k] = cb.acc[(size_t)s * 7 + k];
            if (!SPARSE) s_acc[t * AW + 7] = none;
        }
        s_q[t] = a; s_st[t] = b; s_w[t] = ww;
        if (t == 0) s_flags[0] = 0;
    }
    __syncthreads();

    const uint32_t j = t & 7u, slot = t >> 3;
    // this lane's tree kind: the penalty and stored-value terms of kind j = 1 + pw (anchorer.hpp:2400, 2409; :2330, 2334)
    const uint32_t pw = j >= 1 && j <= 6 ? j - 1 : 0;
    const double go_j = settled(pw / 2 == 0 ? D.params.gap_open[0] : pw / 2 == 1 ? D.params.gap_open[1] : D.params.gap_open[2]);   // (selects, not a load indexed by the lane)
    const double ge_j = settled(pw / 2 == 0 ? D.params.gap_extend[0] : pw / 2 == 1 ? D.params.gap_extend[1] : D.params.gap_extend[2]);
    const double sc = uni_f64(D.params.scale);
    uint32_t qi[QPT], qt[QPT], qoff[QPT], pos[QPT], ins[QPT], off[QPT];
    int32_t q[QPT], sig[QPT];
    float w[QPT], w_init[QPT];
    int acc[QPT][7], ext[QPT];
    double pen[QPT], tt[QPT];
    bool act[QPT];
    auto load_query = [&](int u, uint32_t nq) {
        qi[u] = nq;
        act[u] = nq < count;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[u][k] = none;
        ext[u] = none; qt[u] = 0; qoff[u] = 0; q[u] = 0; pos[u] = 0xFFFFFFFFu; ins[u] = 0; off[u] = 0; sig[u] = 0; w[u] = 0.f; w_init[u] = CL_CHAIN_NEG;
        pen[u] = 0.0; tt[u] = 0.0;
        if (act[u]) {
            const int4 a = s_q[nq], b = s_st[nq];
            const float2 ww = s_w[nq];
            qt[u] = (uint32_t)a.x; qoff[u] = (uint32_t)a.y; q[u] = a.z; pos[u] = (uint32_t)a.w;
            ins[u] = (uint32_t)b.x; off[u] = (uint32_t)b.y; sig[u] = b.z;
            w[u] = ww.x; w_init[u] = ww.y;
            if (j < NK) ext[u] = s_acc[nq * AW + j];
            if (!SPARSE) {
                pen[u] = (pw % 2 == 1) ? sc * (go_j + ge_j * (double)q[u]) : sc * (go_j - ge_j * (double)q[u]);
                tt[u] = sc * ge_j * (double)sig[u];
            }
        }
    };
#pragma unroll
    for (int u = 0; u < QPT; ++u) load_query(u, slot + kSlots * u);

    g_u64* const hacc = uni_ptr((g_u64*)(D.hacc + (size_t)c * kChainMacro * 8));
    g_u64* const xdp = uni_ptr((g_u64*)(D.xdp + (size_t)c * kChainMacro));
    g_u64* const xch = uni_ptr((g_u64*)D.xch);
    g_u32* const xred = uni_ptr((g_u32*)D.xred);
    g_u32* const status = uni_ptr((g_u32*)D.status);
    g_i32* const acc_out = uni_ptr((g_i32*)cb.acc);
    g_f32* const val_out = uni_ptr((g_f32*)cb.val);
    g_f32* const dp_out = uni_ptr((g_f32*)D.dp);
    g_i32* const far_rec = uni_ptr((g_i32*)D.far_rec);
    const uint32_t n_recs = uni(cb.n_recs);
    const uint32_t far_base_c = uni(D.far_rec ? D.far_base[c] : 0u);
    const bool dbg = D.debug != 0 && c == 0;
    uint32_t n_steps = 0, n_fin = 0, n_polled = 0, n_lds = 0;
    unsigned long long t_pre = 0, t_bar = 0, t_post = 0, t0 = 0, t1 = 0, t2 = 0, t_pre_f = 0, t_post_f = 0, t_post_last = 0;
    uint32_t n_fsteps = 0, n_aqmiss = 0, n_tagmiss = 0;
    unsigned long long px[QPT];   // the helper's granule of this lane's kind for query aq[u], asked for one step before the query is finalised
    uint32_t aq[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) { px[u] = 0; aq[u] = 0xFFFFFFFFu; }
    const uint32_t block = first / kChainMacro;
    const unsigned long long t_loop = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t ci = 0;
    uint32_t ge = count ? min((uint32_t)s_st[0].w, WIN) : 0u;
    while (ci < count) {
        // (a group longer than the window is finalised in pieces: its pairs do not precede one another.)  The end of the NEXT step's group is
        // asked for now and waited for at the bottom of the step
        const uint32_t nwb = ge & ~(kSub - 1u);
        const uint32_t ge_next = ge < count ? (uint32_t)s_st[ge].w : count;
        if (dbg) { t0 = __builtin_amdgcn_s_memrealtime(); ++n_steps; }
        // ---- finalise the pairs [ci, ge): only what the other waves wait for — the record in LDS — happens in front of the barrier
        float best_f[QPT];
        int mine_f[QPT];
        bool fin[QPT];
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            best_f[u] = CL_CHAIN_NEG; mine_f[u] = none;
            fin[u] = act[u] && qi[u] >= ci && qi[u] < ge;   // uniform over the query's eight lanes
            if (!fin[u]) continue;
            const uint32_t s = first + qi[u];
            const uint32_t m = qi[u] / kSub;
            const uint32_t bound = m >= BW ? (m - BW + 1u) * kSub : 0u;   // records [0, bound) were final before the query entered the window
            if (dbg && j == 0) ++n_fin;
            if (bound) {
                // what the query's helper has covered so far: a granule per kind, tagged with the macro-block and the number of records [0, cov) it has
                // seen (helpers publish while they go, so a late helper costs a few evaluations here, not a wait); the rest comes out of LDS, where
                // every record of the macro-block lies, every eighth per lane.  No granule of this macro-block (helper not resident, none launched): all of it
                uint32_t cov = bound;
                if (j < NK) {
                    cov = 0;
                    if (dbg && j == 0 && aq[u] != qi[u]) ++n_aqmiss;
                    if (helped && aq[u] == qi[u]) {
                        const uint32_t tg = (uint32_t)(px[u] >> 32);
                        if (dbg && j == 0 && (tg >> 11) != (cover_tag(block, 0) >> 11)) ++n_tagmiss;
                        if ((tg >> 11) == (cover_tag(block, 0) >> 11)) { cov = min(tg & 0x7FFu, bound); ext[u] = max(ext[u], (int)(uint32_t)px[u]); }
                    }
                }
                cov = min8_u(cov);
                if (dbg && j == 0 && cov < bound) { ++n_polled; if (cov == 0) ++n_lds; }
                if (cov < bound && qoff[u] != 0) {
                    for (uint32_t l = cov + j; l < bound; l += 8) {
                        if (SPARSE) {
                            const int4 r = *reinterpret_cast<const int4*>(&s_rec[l * RW]);
                            acc[u][0] = max(acc[u][0], ((uint32_t)r.x <= qt[u] && (uint32_t)r.y < qoff[u]) ? r.z : INT32_MIN);
                        } else {
                            const int4 ra = *reinterpret_cast<const int4*>(&s_rec[l * RW]), rb = *reinterpret_cast<const int4*>(&s_rec[l * RW + 4]),
                                       rc = *reinterpret_cast<const int4*>(&s_rec[l * RW + 8]);
                            accumulate7(acc[u], qt[u], qoff[u], q[u], ra, rb, rc);
                        }
                    }
                }
            }
            // the query's maxima: butterfly over its eight lanes, then lane k keeps kind k
            int mine = none;
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) {
                const int r = max8_i(acc[u][k]);
                mine = j == k ? r : mine;
            }
            mine = max(mine, ext[u]);
            // this combination's candidates for the pair (anchorer.hpp:2379-2412), one kind per lane
            float cand = CL_CHAIN_NEG;
            if (qoff[u] != 0 && j < NK && mine != none) {
                if (SPARSE || j == 0) cand = dec(mine) + w[u];
                else cand = (float)((double)(dec(mine) + w[u]) - pen[u]);
            }
            cand = max8_f(cand);
            float best = fmaxf(w_init[u], cand);
            if (n_combos > 1 && xred) {
                // many combinations: one atomic maximum and one arrival count per pair (chain_walk_kernel's reduction)
                g_u32* red = xred + 2 * (size_t)s;
                if (j == 0) {
                    if (cand != CL_CHAIN_NEG) __hip_atomic_fetch_max(red, (uint32_t)enc(cand) ^ 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(red + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned spins = 0;
                while (__hip_atomic_load(red + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n_combos) {
                    if (++spins > (1u << 20) || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_exchange(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_flags[0] = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t mx = __hip_atomic_load(red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (mx != 0) best = fmaxf(best, dec((int)(mx ^ 0x80000000u)));
            } else if (n_combos > 1) {
                // few combinations: every workgroup reads every other's {tag, candidate} granule of the pair (chain_walk_kernel's sweep)
                const unsigned long long tag = (unsigned long long)(s + 1u) << 32;
                if (j == 0) __hip_atomic_store(&xch[(size_t)c * kChainMacro + qi[u]], tag | (unsigned)__float_as_int(cand), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (true) {
                    bool ok = true;
                    float mx = w_init[u];
                    for (uint32_t cc = 0; cc < n_combos; cc += 4) {
                        unsigned long long x[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            x[v] = cc + v < n_combos ? __hip_atomic_load(&xch[(size_t)(cc + v) * kChainMacro + qi[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag | (unsigned)__float_as_int(CL_CHAIN_NEG);
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            ok = ok && (x[v] >> 32) == (tag >> 32);
                            mx = fmaxf(mx, __int_as_float((int)(unsigned)x[v]));
                        }
                    }
                    if (ok) { best = mx; break; }
                    if (++spins > (1u << 20) || ((spins & 1023u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        __hip_atomic_exchange(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_flags[0] = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            best_f[u] = best; mine_f[u] = mine;
            // the pair's record of this combination: the values stored in its trees (anchorer.hpp:2318-2342), kind k by lane k
            if (pos[u] != 0xFFFFFFFFu) {
                const int e = enc(stored_value(best, SPARSE ? 0u : j, tt[u]));
                if (SPARSE) {
                    if (j == 0) *reinterpret_cast<int4*>(&s_rec[qi[u] * RW]) = make_int4((int)ins[u], (int)off[u], e, 0);
                } else {
                    if (j == 0) *reinterpret_cast<int4*>(&s_rec[qi[u] * RW]) = make_int4((int)ins[u], (int)off[u], sig[u], e);
                    else if (j < NK) s_rec[qi[u] * RW + 3 + j] = e;
                }
            } else if (j == 0) {
                // insertion index 0xFFFFFFFF: never a predecessor
                *reinterpret_cast<int4*>(&s_rec[qi[u] * RW]) = SPARSE ? make_int4(-1, -1, INT32_MIN, 0) : make_int4(-1, -1, 0, INT32_MIN);
            }
        }
        if (dbg) {
            t1 = __builtin_amdgcn_s_memrealtime();
            bool anyfin = false;
#pragma unroll
            for (int u = 0; u < QPT; ++u) anyfin = anyfin || fin[u];
            if (__ballot(anyfin) != 0ull) { ++n_fsteps; t_pre_f += t1 - t0; t_post_f += t_post_last; }
        }
        lds_barrier();
        if (dbg) t2 = __builtin_amdgcn_s_memrealtime();
        if (n_combos > 1 && s_flags[0]) break;
        // ---- slide the window: the sub-blocks the walk has left make room for the next ones, which see this step's records too
        bool slid[QPT];
#pragma unroll
        for (int u = 0; u < QPT; ++u) slid[u] = qi[u] < nwb;
        {   // the waves that hold the NEXT step's pairs go first: their records-to-queries work and their finalisation are the step's critical chain,
            // everything else (the other waves' share of this step's records) fills the issue slots they leave.  Those pairs also ask for their
            // helpers' granules now: by the time they are finalised, a step of work later, the loads have come back
            const uint32_t ge2 = min(ge_next, nwb + WIN);
            bool nf[QPT], any = false;
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const uint32_t nq = slid[u] ? qi[u] + WIN : qi[u];
                nf[u] = nq < count && nq >= ge && nq < ge2;
                any = any || nf[u];
            }
            if (__ballot(any) != 0ull) {
                __builtin_amdgcn_s_setprio(3);
                if (helped) {
#pragma unroll
                    for (int u = 0; u < QPT; ++u) {
                        const uint32_t nq = slid[u] ? qi[u] + WIN : qi[u];
                        if (nf[u] && nq >= WIN && j < NK) { px[u] = __hip_atomic_load(&hacc[(size_t)nq * 8 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aq[u] = nq; }
                    }
                }
            } else __builtin_amdgcn_s_setprio(0);
        }
        // ---- behind the barrier, off the other waves' path: what goes to memory for the pairs just finalised — the DP value, the value every query
        //      returned (the traceback reads it), the granule the helpers follow, the record's stored values and its image for the far pass
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            if (!fin[u]) continue;
            const uint32_t s = first + qi[u];
            if (j < NK) acc_out[(size_t)s * 7 + j] = mine_f[u];
            if (j == 0) {
                if (c == 0) dp_out[s] = best_f[u];
                if (helped) __hip_atomic_store(&xdp[qi[u]], ((unsigned long long)(s + 1u) << 32) | (unsigned)__float_as_int(best_f[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (pos[u] != 0xFFFFFFFFu && j < NK) {
                const float v = stored_value(best_f[u], SPARSE ? 0u : j, tt[u]);
                val_out[(size_t)j * n_recs + pos[u]] = v;
                if (far_rec) {   // the image the branch-and-bound far pass reads (chain_far.hip); its two pad words stay zero
                    g_i32* fr = far_rec + (size_t)(far_base_c + pos[u]) * 12;
                    if (j == 0) *(g_i32x4*)fr = v4i32{(int)ins[u], (int)off[u], sig[u], enc(v)};
                    else fr[3 + j] = enc(v);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < QPT; ++u)
            if (slid[u]) load_query(u, qi[u] + WIN);
        // ---- this step's records to the window's queries: every eighth record per lane, the two queries of a lane share the reads
        {
            bool go[QPT];
            bool any = false;
#pragma unroll
            for (int u = 0; u < QPT; ++u) { go[u] = act[u] && qi[u] >= ge && qoff[u] != 0; any = any || go[u]; }
            if (any) {
                if (SPARSE) {
                    uint32_t l = ci + j;
                    for (; l + 24 < ge; l += 32) {   // four reads in flight: the loop is bound by LDS latency, not by its five operations per record
                        int4 r[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) r[v] = *reinterpret_cast<const int4*>(&s_rec[(l + 8 * v) * RW]);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int u = 0; u < QPT; ++u)
                                if (go[u]) acc[u][0] = max(acc[u][0], ((uint32_t)r[v].x <= qt[u] && (uint32_t)r[v].y < qoff[u]) ? r[v].z : INT32_MIN);
                    }
                    for (; l < ge; l += 8) {
                        const int4 r = *reinterpret_cast<const int4*>(&s_rec[l * RW]);
#pragma unroll
                        for (int u = 0; u < QPT; ++u)
                            if (go[u]) acc[u][0] = max(acc[u][0], ((uint32_t)r.x <= qt[u] && (uint32_t)r.y < qoff[u]) ? r.z : INT32_MIN);
                    }
                } else {
                    uint32_t l = ci + j;
                    for (; l + 8 < ge; l += 16) {
                        int4 ra[2], rb[2], rc[2];
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            ra[v] = *reinterpret_cast<const int4*>(&s_rec[(l + 8 * v) * RW]);
                            rb[v] = *reinterpret_cast<const int4*>(&s_rec[(l + 8 * v) * RW + 4]);
                            rc[v] = *reinterpret_cast<const int4*>(&s_rec[(l + 8 * v) * RW + 8]);
                        }
#pragma unroll
                        for (int v = 0; v < 2; ++v)
#pragma unroll
                            for (int u = 0; u < QPT; ++u)
                                if (go[u]) accumulate7(acc[u], qt[u], qoff[u], q[u], ra[v], rb[v], rc[v]);
                    }
                    if (l < ge) {
                        const int4 ra = *reinterpret_cast<const int4*>(&s_rec[l * RW]), rb = *reinterpret_cast<const int4*>(&s_rec[l * RW + 4]),
                                   rc = *reinterpret_cast<const int4*>(&s_rec[l * RW + 8]);
#pragma unroll
                        for (int u = 0; u < QPT; ++u)
                            if (go[u]) accumulate7(acc[u], qt[u], qoff[u], q[u], ra, rb, rc);
                    }
                }
            }
        }
        ci = ge;
        ge = min(ge_next, (ci & ~(kSub - 1u)) + WIN);
        if (dbg) { const unsigned long long t3 = __builtin_amdgcn_s_memrealtime(); t_pre += t1 - t0; t_bar += t2 - t1; t_post += t3 - t2; t_post_last = t3 - t2; }
    }
    if (dbg) {
        if (t == 0) { atomicAdd(D.status + 27, (uint32_t)(__builtin_amdgcn_s_memtime() - c_entry)); atomicAdd(D.status + 28, (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_entry));
                      atomicAdd(D.status + 22, (uint32_t)(t_loop - t_entry)); atomicAdd(D.status + 23, (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_loop)); atomicAdd(D.status + 24, 1u);
                      atomicAdd(D.status + 19, n_fsteps); atomicAdd(D.status + 20, (uint32_t)t_pre_f); atomicAdd(D.status + 21, (uint32_t)t_post_f); atomicAdd(D.status + 8, n_steps); atomicAdd(D.status + 12, (uint32_t)t_pre); atomicAdd(D.status + 13, (uint32_t)t_bar); atomicAdd(D.status + 14, (uint32_t)t_post); }
        if (n_fin) atomicAdd(D.status + 9, n_fin);
        if (n_polled) atomicAdd(D.status + 10, n_polled);
        if (n_lds) atomicAdd(D.status + 11, n_lds);
        if (n_aqmiss) atomicAdd(D.status + 25, n_aqmiss);
        if (n_tagmiss) atomicAdd(D.status + 26, n_tagmiss);
    }
}

// ---- helper h (of n_help) of combination c ------------------------------------------------------------------------------------------------
template <bool SPARSE>
__device__ __forceinline__ void walk2_helper(const ClChainDevice& D, const uint32_t first, const uint32_t count, const uint32_t c, const uint32_t h,
                                             const uint32_t n_help, const uint32_t WIN, char* smem) {
    using L = W2Lds<SPARSE>;
    constexpr uint32_t RW = L::RW, NK = L::NK;
    int4* s_stat = reinterpret_cast<int4*>(smem + L::h_stat);   // ins, off, shift, 1 if the pair has a record in this combination
    int* s_buf = reinterpret_cast<int*>(smem + L::h_buf);       // [2][kPollBatch][RW]
    uint32_t* s_n = reinterpret_cast<uint32_t*>(smem + L::h_n);
    const ClChainCombo cb = D.combos[c];
    const uint32_t t = threadIdx.x, j = t & 7u, oq = t >> 3;
    const uint32_t BW = WIN / kSub;
    const int none = enc(CL_CHAIN_NEG);
    // the queries this helper owns: sub-blocks BW + h, BW + h + n_help, ... (interleaved: every helper has one urgent sub-block at a time)
    const uint32_t m = BW + h + (oq / kSub) * n_help;
    const uint32_t qp = m * kSub + (oq % kSub);
    const bool act = qp < count;
    const uint32_t bound = (m - BW + 1u) * kSub;               // its queries need the records [0, bound)
    // the largest bound among this helper's sub-blocks that hold a pair at all (uniform)
    uint32_t pmax = 0;
    for (uint32_t k = 0; k < kHelperSubs; ++k) {
        const uint32_t mk = BW + h + k * n_help;
        if (mk * kSub < count) pmax = (mk - BW + 1u) * kSub;
    }
    if (pmax == 0) return;
    uint32_t qt = 0, qoff = 0;
    int32_t q = 0;
    if (act) {
        const uint32_t s = first + qp;
        qt = cb.qt[s];
        if (qt != 0xFFFFFFFFu) { qoff = cb.qoff[s]; q = cb.q[s]; }
    }
    for (uint32_t i = t; i < pmax; i += kChainMacro) {
        int4 st = make_int4(-1, -1, 0, 0);
        const uint32_t pos = cb.own_rec[first + i];
        if (pos != 0xFFFFFFFFu) st = make_int4((int)cb.ins_t[pos], (int)cb.off[pos], cb.sigma[pos], 1);
        s_stat[i] = st;
    }
    __syncthreads();
    int acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = none;
    bool published = !act;
    const g_u64* xdp = (const g_u64*)(D.xdp + (size_t)c * kChainMacro);
    g_u64* hacc = (g_u64*)(D.hacc + (size_t)c * kChainMacro * 8);
    const g_u32* status = (const g_u32*)D.status;
    const double sc = D.params.scale;
    const bool dbg = D.debug != 0 && c == 0 && h == 0;
    uint32_t n_batches = 0, n_idle = 0;
    const uint32_t block = first / kChainMacro;
    constexpr uint32_t kEager = 128;     // a sub-block's maxima go out after every batch once its queries are this close to having seen everything they need
    uint32_t p = 0, b = 0;
    // the polling wave keeps one load of the next granules in flight while the workgroup evaluates a batch
    unsigned long long x = 0;
    if (t < 64) x = t < pmax ? __hip_atomic_load(&xdp[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    while (p < pmax) {
        if (t < 64) {
            // the longest run of fresh granules from p on, turned into records
            uint32_t n = 0;
            unsigned spins = 0;
            while (true) {
                const bool in = p + t < pmax;
                const bool ok = in && (uint32_t)(x >> 32) == first + p + t + 1u;
                const unsigned long long bad = ~__ballot(ok);
                n = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
                if (n) break;
                if (dbg) ++n_idle;
                // the main workgroup has given up (a sibling combination never arrived), or is not there: nothing depends on this helper
                if (++spins > (1u << 18) || ((spins & 255u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { n = 0xFFFFFFFFu; break; }
                __builtin_amdgcn_s_sleep(1);
                x = in ? __hip_atomic_load(&xdp[p + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
            if (n != 0xFFFFFFFFu && t < n) {
                const int4 st = s_stat[p + t];
                const float best = __int_as_float((int)(uint32_t)x);
                int* r = s_buf + ((size_t)b * kPollBatch + t) * RW;
                if (!st.w) {
                    *reinterpret_cast<int4*>(r) = SPARSE ? make_int4(-1, -1, INT32_MIN, 0) : make_int4(-1, -1, 0, INT32_MIN);
                } else if (SPARSE) {
                    *reinterpret_cast<int4*>(r) = make_int4(st.x, st.y, enc(best), 0);
                } else {
                    int e[7];
#pragma unroll
                    for (uint32_t k = 0; k < 7; ++k) {
                        const uint32_t pw = k ? k - 1 : 0;
                        const double tt = sc * D.params.gap_extend[pw / 2] * (double)st.z;
                        e[k] = enc(stored_value(best, k, tt));
                    }
                    *reinterpret_cast<int4*>(r) = make_int4(st.x, st.y, st.z, e[0]);
                    *reinterpret_cast<int4*>(r + 4) = make_int4(e[1], e[2], e[3], e[4]);
                    *reinterpret_cast<int4*>(r + 8) = make_int4(e[5], e[6], 0, 0);
                }
            }
            if (t == 0) s_n[b] = n;
            // the next poll is on its way while the batch is evaluated
            if (n != 0xFFFFFFFFu) x = p + n + t < pmax ? __hip_atomic_load(&xdp[p + n + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        lds_barrier();
        const uint32_t n = s_n[b];
        if (n == 0xFFFFFFFFu) return;
        if (!published && qoff != 0 && bound > p) {
            const uint32_t lim = min(n, bound - p);
            const int* buf = s_buf + (size_t)b * kPollBatch * RW;
            for (uint32_t l = j; l < lim; l += 8) {
                if (SPARSE) {
                    const int4 r = *reinterpret_cast<const int4*>(&buf[l * RW]);
                    acc[0] = max(acc[0], ((uint32_t)r.x <= qt && (uint32_t)r.y < qoff) ? r.z : INT32_MIN);
                } else {
                    const int4 ra = *reinterpret_cast<const int4*>(&buf[l * RW]), rb = *reinterpret_cast<const int4*>(&buf[l * RW + 4]),
                               rc = *reinterpret_cast<const int4*>(&buf[l * RW + 8]);
                    accumulate7(acc, qt, qoff, q, ra, rb, rc);
                }
            }
        }
        p += n;
        if (dbg) ++n_batches;
        if (!published && bound <= p + kEager) {
            // the maxima so far go to the main workgroup, kind k by lane k, tagged with how much they cover: it takes what is there when it finalises
            // the query and evaluates the rest itself, so a helper that is a batch behind costs it eight evaluations, not a wait
            const uint32_t cov = min(p, bound);
            int mine = none;
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) {
                const int r = max8_i(acc[k]);
                mine = j == k ? r : mine;
            }
            if (j < NK) __hip_atomic_store(&hacc[(size_t)qp * 8 + j], ((unsigned long long)cover_tag(block, cov) << 32) | (uint32_t)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            published = cov == bound;
        }
        b ^= 1u;
    }
    if (dbg && t == 0) { atomicAdd(D.status + 16, n_batches); atomicAdd(D.status + 17, p); atomicAdd(D.status + 18, n_idle); }
}

// blockIdx.x = role * stride + c with stride a multiple of 8: the main workgroup of a combination (role 0) and its helpers (roles 1 ..) land on
// the same XCD (workgroups are dealt round-robin over the eight), so granules travel through one L2; placement is for speed only
template <bool SPARSE, int QPT>
__global__ void __launch_bounds__(kChainMacro) chain_walk2_kernel(ClChainDevice D, uint32_t first, uint32_t count, uint32_t n_help, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t role = blockIdx.x / stride, c = blockIdx.x % stride;
    if (c >= D.n_combos) return;
    if (role == 0) walk2_main<SPARSE, QPT>(D, first, count, c, n_help > 0, smem);
    else walk2_helper<SPARSE>(D, first, count, c, role - 1, n_help, kSlots * QPT, smem);
}

}  // namespace

// helpers a main workgroup of window `128 * qpt` needs for full coverage of a macro-block
uint32_t cl_chain_walk2_helpers(uint32_t qpt) { return (kChainMacro / kSub - kSlots * qpt / kSub + kHelperSubs - 1) / kHelperSubs; }

// `done` (may be null): recorded when the launch has finished — the event record rides on the launch (hipExtLaunchKernel) instead of being a runtime
// call of its own: the chaining DP is as long as the host needs for its calls
hipError_t cl_chain_launch_walk2(const ClChainDevice& D, uint32_t first, uint32_t count, uint32_t qpt, uint32_t n_help, hipStream_t stream, hipEvent_t done) {
    static ClDeviceOnce attr_once;   // more than 64 KB of dynamic LDS needs the opt-in once per function
    attr_once([] {
        const int cap = 160 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_walk2_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_walk2_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_walk2_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_walk2_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    });
    const uint32_t stride = (D.n_combos + 7u) & ~7u;
    const dim3 grid(stride * (1u + n_help));
    if (D.sparse) {
        const size_t lds = W2Lds<true>::bytes;
        if (qpt == 1) hipExtLaunchKernelGGL((chain_walk2_kernel<true, 1>), grid, dim3(kChainMacro), (uint32_t)lds, stream, nullptr, done, 0, D, first, count, n_help, stride);
        else hipExtLaunchKernelGGL((chain_walk2_kernel<true, 2>), grid, dim3(kChainMacro), (uint32_t)lds, stream, nullptr, done, 0, D, first, count, n_help, stride);
    } else {
        const size_t lds = W2Lds<false>::bytes;
        if (qpt == 1) hipExtLaunchKernelGGL((chain_walk2_kernel<false, 1>), grid, dim3(kChainMacro), (uint32_t)lds, stream, nullptr, done, 0, D, first, count, n_help, stride);
        else hipExtLaunchKernelGGL((chain_walk2_kernel<false, 2>), grid, dim3(kChainMacro), (uint32_t)lds, stream, nullptr, done, 0, D, first, count, n_help, stride);
    }
    return hipGetLastError();
}
