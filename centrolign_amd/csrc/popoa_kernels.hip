// popoa_kernels.hip — hand-written gfx950 kernels for centrolign's between-anchor graph x graph DP
// (reference: po_poa_internal<true,NumPW>, include/centrolign/alignment.hpp:753-1151).
//
// popoa_general_kernel: one workgroup per subproblem, any DAG pair.  The (n1+1) x (n2+1) matrix is swept
// by anti-diagonals in topological-rank space; every lane owns one cell of the current anti-diagonal.  The
// 1 + 2*NumPW score planes (M, I_k, D_k; cell_t of alignment.hpp:738-751) live in HBM in ANTI-DIAGONAL-MAJOR
// order, so the lanes of a wave write one contiguous run per plane and, for chain-like graphs, read three
// contiguous runs (coalesced).  Lane 0 then picks the best sink pair and walks the traceback with exactly
// the reference's equality tests and tie-break order (alignment.hpp:979-1138).
//
// popoa_ring_kernel: the same sweep for subproblems whose topology and a ring of recent anti-diagonals fit LDS (below).
#include <hip/hip_runtime.h>
#include <mutex>
#include "device_once.h"
#include <stdint.h>

#include "popoa_device.h"

namespace {

// geometry of the anti-diagonal-major storage of one (n1+1) x (n2+1) matrix
struct DiagGeom {
    uint32_t n1, n2, m, mx, tri;
    __device__ DiagGeom(uint32_t n1_, uint32_t n2_) : n1(n1_), n2(n2_) {
        m = n1 < n2 ? n1 : n2;
        mx = n1 < n2 ? n2 : n1;
        tri = m * (m + 1) / 2;
    }
    __device__ __forceinline__ uint32_t lo(uint32_t d) const { return d > n2 ? d - n2 : 0; }
    __device__ __forceinline__ uint32_t hi(uint32_t d) const { return d < n1 ? d : n1; }
    // number of cells on anti-diagonals 0 .. d-1
    __device__ __forceinline__ uint32_t off(uint32_t d) const {
        if (d <= m) return d * (d + 1) / 2;
        if (d <= mx) return tri + (d - m) * (m + 1);
        uint32_t k = d - mx;
        return tri + (mx - m) * (m + 1) + k * (m + 1) - k * (k - 1) / 2;
    }
    __device__ __forceinline__ uint32_t idx(uint32_t a, uint32_t b) const {
        uint32_t d = a + b;
        return off(d) + a - lo(d);
    }
};

__device__ __forceinline__ int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

// a plane pointer every lane holds the same value of (derived from a workgroup's problem descriptor), moved to scalar registers and kept in the GLOBAL
// address space: stores through it take the "scalar base + 32-bit vector offset" form instead of a 64-bit vector add each (a generic pointer would make
// them flat stores, which wait on two counters)
typedef __attribute__((address_space(1))) int32_t g_i32;
__device__ __forceinline__ void plane_store(g_i32* base, uint32_t byte_off, int32_t v) {   // byte_off < 2^32: a matrix has fewer than 2^30 cells
    *reinterpret_cast<g_i32*>(reinterpret_cast<__attribute__((address_space(1))) char*>(base) + byte_off) = v;
}
__device__ __forceinline__ g_i32* uniform_plane(int32_t* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (g_i32*)(((uint64_t)hi << 32) | lo);
}

template <int NPW>
struct Planes {
    int32_t* base;
    uint32_t cells;
    __device__ __forceinline__ int32_t* M() const { return base; }
    __device__ __forceinline__ int32_t* I(int k) const { return base + (size_t)(1 + k) * cells; }
    __device__ __forceinline__ int32_t* D(int k) const { return base + (size_t)(1 + NPW + k) * cells; }
};

// One DP cell in pull form (equivalent to the reference's push form, alignment.hpp:813-938):
//   boundary column b == 0 : I_k = max(-(o_k+e_k) [a is a source], I_k(p,0) - e_k)        (:832-845)
//   boundary row    a == 0 : D_k symmetric                                               (:864-877)
//   interior               : I_k = max_p max(Mf(p,b) - (o_k+e_k), I_k(p,b) - e_k)         (:907-916, :878-885)
//                            D_k = max_q max(Mf(a,q) - (o_k+e_k), D_k(a,q) - e_k)         (:918-927, :846-853)
//                            M   = s(a,b) + max_{p,q} Mf(p,q), with Mf(corner) == 0       (:929-936, :814-818, :854-861, :886-893)
//   Mf = max(M, I_k, D_k)                                                                 (:837, :869, :903-905)
// p ranges over previous1(a) plus the boundary index when a is a source; q likewise.
template <int NPW>
__device__ __forceinline__ void compute_cell(const ClDeviceBatch& B, const ClProbDesc& pd, const DiagGeom& G,
                                             const Planes<NPW>& pl, const ClScoreParams& P, uint32_t a, uint32_t b,
                                             uint32_t self_idx) {
    const uint8_t* lab1 = B.lab[0] + pd.node_base[0];
    const uint8_t* lab2 = B.lab[1] + pd.node_base[1];
    const uint32_t* poff1 = B.poff[0] + pd.node_base[0];
    const uint32_t* poff2 = B.poff[1] + pd.node_base[1];
    int32_t M = CL_NEG_INF, I[NPW], D[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) { I[k] = CL_NEG_INF; D[k] = CL_NEG_INF; }

    uint32_t e1b = 0, e1e = 0, e2b = 0, e2e = 0;
    uint32_t l1 = 0, l2 = 0;
    if (a) { e1b = poff1[a - 1]; e1e = poff1[a]; l1 = lab1[a - 1]; }
    if (b) { e2b = poff2[b - 1]; e2e = poff2[b]; l2 = lab2[b - 1]; }
    const bool src1 = l1 & 0x80, src2 = l2 & 0x80;

    if (a) {
        if (b == 0) {
            for (uint32_t e = e1b; e < e1e; ++e) {
                uint32_t c = G.idx(B.pidx[0][e], 0);
#pragma unroll
                for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], pl.I(k)[c] - P.ext[k]);
            }
            if (src1) {
#pragma unroll
                for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], -P.oe[k]);
            }
        } else {
            for (uint32_t e = e1b; e < e1e; ++e) {
                uint32_t c = G.idx(B.pidx[0][e], b);
                int32_t m = pl.M()[c];
#pragma unroll
                for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], imax(m - P.oe[k], pl.I(k)[c] - P.ext[k]));
            }
            if (src1) {
                int32_t m = pl.M()[G.idx(0, b)];
#pragma unroll
                for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], m - P.oe[k]);
            }
        }
    }
    if (b) {
        if (a == 0) {
            for (uint32_t f = e2b; f < e2e; ++f) {
                uint32_t c = G.idx(0, B.pidx[1][f]);
#pragma unroll
                for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], pl.D(k)[c] - P.ext[k]);
            }
            if (src2) {
#pragma unroll
                for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], -P.oe[k]);
            }
        } else {
            for (uint32_t f = e2b; f < e2e; ++f) {
                uint32_t c = G.idx(a, B.pidx[1][f]);
                int32_t m = pl.M()[c];
#pragma unroll
                for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], imax(m - P.oe[k], pl.D(k)[c] - P.ext[k]));
            }
            if (src2) {
                int32_t m = pl.M()[G.idx(a, 0)];
#pragma unroll
                for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], m - P.oe[k]);
            }
        }
    }
    if (a && b) {
        const int32_t s = ((l1 & 0x7f) == (l2 & 0x7f)) ? P.match : -P.mismatch;
        const uint32_t e1x = e1e + (src1 ? 1u : 0u), e2x = e2e + (src2 ? 1u : 0u);
        for (uint32_t e = e1b; e < e1x; ++e) {
            uint32_t pa = e < e1e ? B.pidx[0][e] : 0u;
            for (uint32_t f = e2b; f < e2x; ++f) {
                uint32_t pb = f < e2e ? B.pidx[1][f] : 0u;
                int32_t v = (pa | pb) ? pl.M()[G.idx(pa, pb)] : 0;
                M = imax(M, v + s);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        M = imax(M, imax(I[k], D[k]));
        pl.I(k)[self_idx] = I[k];
        pl.D(k)[self_idx] = D[k];
    }
    pl.M()[self_idx] = M;
}

// best sink pair + traceback by one lane; alignment.hpp:979-1138, rule for rule
template <int NPW>
__device__ void traceback(const ClDeviceBatch& B, const ClProbDesc& pd, const DiagGeom& G, const Planes<NPW>& pl,
                          const ClScoreParams& P, uint32_t prob) {
    const uint8_t* lab1 = B.lab[0] + pd.node_base[0];
    const uint8_t* lab2 = B.lab[1] + pd.node_base[1];
    const uint32_t* poff1 = B.poff[0] + pd.node_base[0];
    const uint32_t* poff2 = B.poff[1] + pd.node_base[1];
    const uint32_t* snk1 = B.snk[0] + pd.snk_base[0];
    const uint32_t* snk2 = B.snk[1] + pd.snk_base[1];
    const int32_t* Mp = pl.M();

    // first strictly better in the given sink order (:982-989)
    uint32_t a = 0, b = 0;
    int32_t best = 0;
    bool have = false;
    for (uint32_t i = 0; i < pd.snk_cnt[0]; ++i)
        for (uint32_t j = 0; j < pd.snk_cnt[1]; ++j) {
            int32_t v = Mp[G.idx(snk1[i], snk2[j])];
            if (!have || v > best) { have = true; best = v; a = snk1[i]; b = snk2[j]; }
        }
    B.out_score[prob] = have ? best : 0;
    uint32_t status = 0, len = 0;
    const uint32_t cap = pd.n1 + pd.n2;
    uint2* out = B.out_pairs + pd.out_base;
    int comp = 0;
    while (have) {
        if (len >= cap) { status = 2; break; }  // cannot happen on a DAG; keeps a corrupt input from overrunning
        const uint32_t c = G.idx(a, b);
        const int32_t Mv = Mp[c];
        if (comp == 0) {  // gap close test order I_0, D_0, I_1, D_1, ... (:1048-1066)
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                if (Mv == pl.I(k)[c]) { comp = k + 1; break; }
                if (Mv == pl.D(k)[c]) { comp = -k - 1; break; }
            }
        }
        uint32_t e1b = 0, e1e = 0, e2b = 0, e2e = 0, x1 = 0, x2 = 0, l1 = 0, l2 = 0;
        if (a) { e1b = poff1[a - 1]; e1e = poff1[a]; l1 = lab1[a - 1]; x1 = l1 >> 7; }
        if (b) { e2b = poff2[b - 1]; e2e = poff2[b]; l2 = lab2[b - 1]; x2 = l2 >> 7; }
        uint32_t na = 0xFFFFFFFFu, nb = 0xFFFFFFFFu;
        if (comp == 0) {
            if (!a || !b) { status = 3; break; }
            out[cap - 1 - len] = make_uint2(a, b);
            ++len;
            const int32_t s = ((l1 & 0x7f) == (l2 & 0x7f)) ? P.match : -P.mismatch;
            // the inner break leaves only the inner loop: LAST prev1 with a hit, its FIRST prev2 (:1091-1099)
            for (uint32_t e = e1b; e < e1e + x1; ++e) {
                uint32_t pa = e < e1e ? B.pidx[0][e] : 0u;
                for (uint32_t f = e2b; f < e2e + x2; ++f) {
                    uint32_t pb = f < e2e ? B.pidx[1][f] : 0u;
                    if (Mp[G.idx(pa, pb)] + s == Mv) { na = pa; nb = pb; break; }
                }
            }
        } else if (comp > 0) {
            if (!a) { status = 3; break; }
            out[cap - 1 - len] = make_uint2(a, 0u);
            ++len;
            const int k = comp - 1;
            const int32_t Iv = pl.I(k)[c];
            for (uint32_t e = e1b; e < e1e + x1; ++e) {  // open before extend, first predecessor wins (:1105-1118)
                uint32_t pa = e < e1e ? B.pidx[0][e] : 0u;
                uint32_t pc = G.idx(pa, b);
                if (Iv == Mp[pc] - P.oe[k]) { comp = 0; na = pa; nb = b; break; }
                if (Iv == pl.I(k)[pc] - P.ext[k]) { na = pa; nb = b; break; }
            }
        } else {
            if (!b) { status = 3; break; }
            out[cap - 1 - len] = make_uint2(0u, b);
            ++len;
            const int k = -comp - 1;
            const int32_t Dv = pl.D(k)[c];
            for (uint32_t f = e2b; f < e2e + x2; ++f) {  // (:1123-1136)
                uint32_t pb = f < e2e ? B.pidx[1][f] : 0u;
                uint32_t pc = G.idx(a, pb);
                if (Dv == Mp[pc] - P.oe[k]) { comp = 0; na = a; nb = pb; break; }
                if (Dv == pl.D(k)[pc] - P.ext[k]) { na = a; nb = pb; break; }
            }
        }
        if (na == 0xFFFFFFFFu) break;
        a = na;
        b = nb;
    }
    B.out_len[prob] = len;
    B.out_status[prob] = status;
}

// The same traceback walked by the first WAVE of the workgroup.  Rule for rule it is the function above (one lane executes exactly that
// code for a step that needs it); what the other 63 lanes add is look-ahead over the stretches where the walk has no choice to make:
//   * diagonal run (state "match"): lane i looks at cell (a - i, b - i).  While both nodes have exactly one predecessor, the node of
//     rank one less, and are not sources, the only candidate pair is (a - i - 1, b - i - 1); and while the cell's M equals none of its
//     I_k / D_k the walk stays in the match state and takes that pair (M = s + M(pair) then holds by construction).  The lanes test
//     this for 64 cells at once, a ballot finds the end of the run, the run is emitted in one go;
//   * gap run (state I_k or D_k): lane i looks at the cell i steps up (left); with a single predecessor the step either re-opens
//     (I_k == M(pred) - open - extend, tested first, alignment.hpp:1105-1118) — the run ends there, back in the match state — or extends.
// A whole-repeat deletion between two MSA graphs is a gap run of two thousand cells: thirty-odd round trips instead of two thousand
// dependent loads.
template <int NPW>
__device__ void traceback_wave(const ClDeviceBatch& B, const ClProbDesc& pd, const DiagGeom& G, const Planes<NPW>& pl,
                               const ClScoreParams& P, uint32_t prob) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t* lab1 = B.lab[0] + pd.node_base[0];
    const uint8_t* lab2 = B.lab[1] + pd.node_base[1];
    const uint32_t* poff1 = B.poff[0] + pd.node_base[0];
    const uint32_t* poff2 = B.poff[1] + pd.node_base[1];
    const uint32_t* snk1 = B.snk[0] + pd.snk_base[0];
    const uint32_t* snk2 = B.snk[1] + pd.snk_base[1];
    const int32_t* Mp = pl.M();
    uint32_t a = 0, b = 0, status = 0, len = 0;
    int32_t best = 0;
    int have_i = 0;
    if (lane == 0) {   // first strictly better in the given sink order (:982-989)
        bool have = false;
        for (uint32_t i = 0; i < pd.snk_cnt[0]; ++i)
            for (uint32_t j = 0; j < pd.snk_cnt[1]; ++j) {
                int32_t v = Mp[G.idx(snk1[i], snk2[j])];
                if (!have || v > best) { have = true; best = v; a = snk1[i]; b = snk2[j]; }
            }
        B.out_score[prob] = have ? best : 0;
        have_i = have ? 1 : 0;
    }
    a = __shfl(a, 0); b = __shfl(b, 0); have_i = __shfl(have_i, 0);
    const uint32_t cap = pd.n1 + pd.n2;
    uint2* out = B.out_pairs + pd.out_base;
    int comp = 0;
    // a node (1-based rank r >= 1) of graph g is "plain" when its only predecessor is rank r - 1 >= 1 ... or, for r == 1, when it
    // has no predecessor and is a source (its only predecessor is then the boundary index 0 = r - 1)
    auto plain1 = [&](uint32_t r) -> bool {
        const uint32_t e0 = poff1[r - 1], e1 = poff1[r];
        const bool src = lab1[r - 1] >> 7;
        if (r == 1) return e1 == e0 && src;
        return e1 - e0 == 1 && !src && B.pidx[0][e0] == r - 1;
    };
    auto plain2 = [&](uint32_t r) -> bool {
        const uint32_t e0 = poff2[r - 1], e1 = poff2[r];
        const bool src = lab2[r - 1] >> 7;
        if (r == 1) return e1 == e0 && src;
        return e1 - e0 == 1 && !src && B.pidx[1][e0] == r - 1;
    };
    while (have_i) {
        if (len >= cap) { status = 2; break; }
        // ---- look-ahead ------------------------------------------------------------------------------------------------
        uint32_t run = 0;
        bool run_opens = false;
        if (comp == 0) {
            bool ok = false;
            if (lane < a && lane < b) {
                const uint32_t ai = a - lane, bi = b - lane;
                const uint32_t c = G.idx(ai, bi);
                const int32_t Mv = Mp[c];
                bool closes = false;
#pragma unroll
                for (int k = 0; k < NPW; ++k) closes = closes || Mv == pl.I(k)[c] || Mv == pl.D(k)[c];
                // the step whose only candidate is the corner (0, 0) ends the walk (the corner's M is -inf in memory): not part of a run
                ok = !closes && !(ai == 1 && bi == 1) && plain1(ai) && plain2(bi);
            }
            const unsigned long long m = __ballot(ok);
            run = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
            run = min(run, cap - len);
            if (run) {
                if (lane < run) out[cap - 1 - (len + lane)] = make_uint2(a - lane, b - lane);
                len += run; a -= run; b -= run;
                // the pair reached is (a, b) with the boundary index 0 standing for "nothing left"; the walk ends when the state has no
                // predecessor, which the single step below finds out
                continue;
            }
        } else if (comp > 0) {
            const int k = comp - 1;
            bool ok = false, opens = false;
            if (lane < a) {   // (also along the boundary column b == 0: the same equalities decide there)
                const uint32_t ai = a - lane;
                if (plain1(ai)) {
                    const uint32_t c = G.idx(ai, b), pc = G.idx(ai - 1, b);
                    const int32_t Iv = pl.I(k)[c];
                    opens = Iv == Mp[pc] - P.oe[k];
                    ok = opens || Iv == pl.I(k)[pc] - P.ext[k];
                }
            }
            const unsigned long long m = __ballot(ok), mo = __ballot(opens);
            run = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
            const unsigned long long first_open = mo & (run >= 64 ? ~0ull : ((1ull << run) - 1));
            if (first_open) { run = (uint32_t)__builtin_ctzll(first_open) + 1; run_opens = true; }
            run = min(run, cap - len);
            if (run) {
                if (lane < run) out[cap - 1 - (len + lane)] = make_uint2(a - lane, 0u);
                len += run; a -= run;
                if (run_opens) comp = 0;
                continue;
            }
        } else {
            const int k = -comp - 1;
            bool ok = false, opens = false;
            if (lane < b) {
                const uint32_t bi = b - lane;
                if (plain2(bi)) {
                    const uint32_t c = G.idx(a, bi), pc = G.idx(a, bi - 1);
                    const int32_t Dv = pl.D(k)[c];
                    opens = Dv == Mp[pc] - P.oe[k];
                    ok = opens || Dv == pl.D(k)[pc] - P.ext[k];
                }
            }
            const unsigned long long m = __ballot(ok), mo = __ballot(opens);
            run = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);
            const unsigned long long first_open = mo & (run >= 64 ? ~0ull : ((1ull << run) - 1));
            if (first_open) { run = (uint32_t)__builtin_ctzll(first_open) + 1; run_opens = true; }
            run = min(run, cap - len);
            if (run) {
                if (lane < run) out[cap - 1 - (len + lane)] = make_uint2(0u, b - lane);
                len += run; b -= run;
                if (run_opens) comp = 0;
                continue;
            }
        }
        // ---- one step by the rules of the reference, on lane 0 ----------------------------------------------------------
        uint32_t na = 0xFFFFFFFFu, nb = 0xFFFFFFFFu;
        int ncomp = comp;
        uint32_t st = 0, emitted = 0;
        if (lane == 0) {
            const uint32_t c = G.idx(a, b);
            const int32_t Mv = Mp[c];
            if (ncomp == 0) {  // gap close test order I_0, D_0, I_1, D_1, ... (:1048-1066)
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    if (Mv == pl.I(k)[c]) { ncomp = k + 1; break; }
                    if (Mv == pl.D(k)[c]) { ncomp = -k - 1; break; }
                }
            }
            uint32_t e1b = 0, e1e = 0, e2b = 0, e2e = 0, x1 = 0, x2 = 0, l1 = 0, l2 = 0;
            if (a) { e1b = poff1[a - 1]; e1e = poff1[a]; l1 = lab1[a - 1]; x1 = l1 >> 7; }
            if (b) { e2b = poff2[b - 1]; e2e = poff2[b]; l2 = lab2[b - 1]; x2 = l2 >> 7; }
            if (ncomp == 0) {
                if (!a || !b) st = 3;
                else {
                    out[cap - 1 - len] = make_uint2(a, b);
                    emitted = 1;
                    const int32_t s = ((l1 & 0x7f) == (l2 & 0x7f)) ? P.match : -P.mismatch;
                    // the inner break leaves only the inner loop: LAST prev1 with a hit, its FIRST prev2 (:1091-1099)
                    for (uint32_t e = e1b; e < e1e + x1; ++e) {
                        uint32_t pa = e < e1e ? B.pidx[0][e] : 0u;
                        for (uint32_t f = e2b; f < e2e + x2; ++f) {
                            uint32_t pb = f < e2e ? B.pidx[1][f] : 0u;
                            if (Mp[G.idx(pa, pb)] + s == Mv) { na = pa; nb = pb; break; }
                        }
                    }
                }
            } else if (ncomp > 0) {
                if (!a) st = 3;
                else {
                    out[cap - 1 - len] = make_uint2(a, 0u);
                    emitted = 1;
                    const int k = ncomp - 1;
                    const int32_t Iv = pl.I(k)[c];
                    for (uint32_t e = e1b; e < e1e + x1; ++e) {  // open before extend, first predecessor wins (:1105-1118)
                        uint32_t pa = e < e1e ? B.pidx[0][e] : 0u;
                        uint32_t pc = G.idx(pa, b);
                        if (Iv == Mp[pc] - P.oe[k]) { ncomp = 0; na = pa; nb = b; break; }
                        if (Iv == pl.I(k)[pc] - P.ext[k]) { na = pa; nb = b; break; }
                    }
                }
            } else {
                if (!b) st = 3;
                else {
                    out[cap - 1 - len] = make_uint2(0u, b);
                    emitted = 1;
                    const int k = -ncomp - 1;
                    const int32_t Dv = pl.D(k)[c];
                    for (uint32_t f = e2b; f < e2e + x2; ++f) {  // (:1123-1136)
                        uint32_t pb = f < e2e ? B.pidx[1][f] : 0u;
                        uint32_t pc = G.idx(a, pb);
                        if (Dv == Mp[pc] - P.oe[k]) { ncomp = 0; na = a; nb = pb; break; }
                        if (Dv == pl.D(k)[pc] - P.ext[k]) { na = a; nb = pb; break; }
                    }
                }
            }
        }
        na = __shfl(na, 0); nb = __shfl(nb, 0); ncomp = __shfl(ncomp, 0); st = __shfl(st, 0); emitted = __shfl(emitted, 0);
        len += emitted;
        comp = ncomp;
        if (st) { status = st; break; }
        if (na == 0xFFFFFFFFu) break;
        a = na;
        b = nb;
    }
    if (lane == 0) {
        B.out_len[prob] = len;
        B.out_status[prob] = status;
    }
}

template <int NPW, int BLOCK>
__global__ void __launch_bounds__(BLOCK) popoa_general_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist,
                                                              ClScoreParams P) {
    cl_tick_start(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
    const uint32_t prob = plist[blockIdx.x];
    const ClProbDesc pd = B.desc[prob];
    const DiagGeom G(pd.n1, pd.n2);
    Planes<NPW> pl;
    pl.base = B.planes + pd.plane_base;
    pl.cells = (pd.n1 + 1) * (pd.n2 + 1);
    const uint32_t tid = threadIdx.x;

    if (tid == 0) {  // the corner stays -inf in memory; the diagonal term treats it as 0 (alignment.hpp:814-818)
        pl.M()[0] = CL_NEG_INF;
#pragma unroll
        for (int k = 0; k < NPW; ++k) { pl.I(k)[0] = CL_NEG_INF; pl.D(k)[0] = CL_NEG_INF; }
    }
    const uint32_t last = pd.n1 + pd.n2;
    uint32_t off = 1;  // G.off(1)
    for (uint32_t d = 1; d <= last; ++d) {
        const uint32_t lo = G.lo(d), cnt = G.hi(d) - lo + 1;
        for (uint32_t t = tid; t < cnt; t += BLOCK) {
            const uint32_t a = lo + t;
            compute_cell<NPW>(B, pd, G, pl, P, a, d - a, off + t);
        }
        off += cnt;
        __syncthreads();  // s_waitcnt vmcnt(0) + barrier: this anti-diagonal is visible to the whole workgroup
    }
    if (tid < 64 && !B.skip_traceback) traceback_wave<NPW>(B, pd, G, pl, P, prob);
    cl_tick_end(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// popoa_ring_kernel: the same sweep for subproblems whose working set fits LDS.  In LDS: a ring of the most recent anti-
// diagonals ([diagonal & (depth-1)][position][plane], depth a power of two, ClProbDesc::pad), one packed record per node
// ({first predecessor or list start, degree | label << 16 | source << 31}), the predecessor lists, and the boundary column
// and row.  The boundary (a == 0 or b == 0: chains of gap extensions only, alignment.hpp:832-894) is a 1-D recurrence per
// graph and is done up front by two lanes, so the sweep only meets interior cells — one code path, no per-diagonal
// divergence — whose reads are LDS reads; a read that reaches further back than the ring goes to the HBM planes, which are
// written as before (the traceback reads them) and drained every depth/2 anti-diagonals.
template <int NPW, int BLOCK>
__global__ void __launch_bounds__(BLOCK) popoa_ring_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist, ClScoreParams P) {
    extern __shared__ int32_t lds[];
    constexpr int PL = 1 + 2 * NPW, BP = 1 + NPW;
    cl_tick_start(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
    const uint32_t prob = plist[blockIdx.x];
    const ClProbDesc pd = B.desc[prob];
    const DiagGeom G(pd.n1, pd.n2);
    Planes<NPW> pl;
    pl.base = B.planes + pd.plane_base;
    pl.cells = (pd.n1 + 1) * (pd.n2 + 1);
    const uint32_t tid = threadIdx.x, n1 = pd.n1, n2 = pd.n2;
    // ClProbDesc::pad: ring depth (a power of two, <= 2^14) | 0x8000 when the ring serves every read of the subproblem
    const uint32_t depth = pd.pad & 0x7FFFu, width = (n1 < n2 ? n1 : n2) + 1, mask = depth - 1;
    const bool full_cover = pd.pad & 0x8000u;

    // LDS layout (int32 units): ring | node records 1 | node records 2 | predecessor lists 1 | 2 | boundary column | boundary row
    int32_t* ring = lds;
    uint2* rec1 = reinterpret_cast<uint2*>(lds + (((size_t)depth * width * PL + 1) & ~(size_t)1));   // 8-byte aligned
    uint2* rec2 = rec1 + n1;
    uint32_t* pl1 = reinterpret_cast<uint32_t*>(rec2 + n2);
    const uint32_t* gp1 = B.poff[0] + pd.node_base[0];
    const uint32_t* gp2 = B.poff[1] + pd.node_base[1];
    const uint32_t e10 = gp1[0], e11 = gp1[n1], e20 = gp2[0], e21 = gp2[n2];
    uint32_t* pl2 = pl1 + (e11 - e10);
    int32_t* bcol = reinterpret_cast<int32_t*>(pl2 + (e21 - e20));   // [(n1+1)][BP]: M, I_k of cell (a, 0)
    int32_t* brow = bcol + (size_t)(n1 + 1) * BP;                    // [(n2+1)][BP]: M, D_k of cell (0, b)
    for (uint32_t i = tid; i < e11 - e10; i += BLOCK) pl1[i] = B.pidx[0][e10 + i];
    for (uint32_t i = tid; i < e21 - e20; i += BLOCK) pl2[i] = B.pidx[1][e20 + i];
    for (int s = 0; s < 2; ++s) {
        const uint32_t n = s ? n2 : n1, e0 = s ? e20 : e10;
        const uint32_t* gp = s ? gp2 : gp1;
        const uint8_t* gl = B.lab[s] + pd.node_base[s];
        uint2* rec = s ? rec2 : rec1;
        for (uint32_t i = tid; i < n; i += BLOCK) {
            const uint32_t b0 = gp[i] - e0, deg = gp[i + 1] - gp[i], l = gl[i];
            rec[i] = make_uint2(deg == 1 ? B.pidx[s][gp[i]] : b0, (deg & 0xFFFFu) | ((l & 0x7Fu) << 16) | ((l >> 7) << 31));
        }
    }
    __syncthreads();
    // boundary column (lane 0) and row (lane 1): I_k(a,0) = max(source ? -(o_k+e_k) : -inf, max_p I_k(p,0) - e_k), M = max_k;
    // the corner stays -inf (the diagonal term treats it as 0, alignment.hpp:814-818)
    if (tid < 2) {
        const uint32_t n = tid ? n2 : n1;
        const uint2* rec = tid ? rec2 : rec1;
        const uint32_t* lst = tid ? pl2 : pl1;
        int32_t* bnd = tid ? brow : bcol;
        int32_t prev[NPW];   // the values of node v-1: a chain step (the usual case) needs nothing else
#pragma unroll
        for (int k = 0; k < NPW; ++k) prev[k] = CL_NEG_INF;
#pragma unroll
        for (int k = 0; k < BP; ++k) bnd[k] = CL_NEG_INF;
        uint2 r = n ? rec[0] : make_uint2(0u, 0u);
        for (uint32_t v = 1; v <= n; ++v) {
            const uint2 r_next = v < n ? rec[v] : make_uint2(0u, 0u);   // fetched while this node is worked on
            const uint32_t deg = r.y & 0xFFFFu;
            int32_t g[NPW], m = CL_NEG_INF;
#pragma unroll
            for (int k = 0; k < NPW; ++k) g[k] = (r.y >> 31) ? -P.oe[k] : CL_NEG_INF;
            if (deg == 1 && r.x == v - 1) {
#pragma unroll
                for (int k = 0; k < NPW; ++k) g[k] = imax(g[k], prev[k] - P.ext[k]);
            } else {
                for (uint32_t e = 0; e < deg; ++e) {
                    const uint32_t p = deg == 1 ? r.x : lst[r.x + e];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) g[k] = imax(g[k], bnd[(size_t)p * BP + 1 + k] - P.ext[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < NPW; ++k) { bnd[(size_t)v * BP + 1 + k] = g[k]; m = imax(m, g[k]); prev[k] = g[k]; }
            bnd[(size_t)v * BP] = m;
            r = r_next;
        }
    }
    __syncthreads();
    // the boundary cells also go to the HBM planes (the traceback reads them), all lanes at once
    for (uint32_t i = tid; i <= n1 + n2; i += BLOCK) {
        const bool row = i > n1;                       // i in [0, n1]: cell (i, 0); i in (n1, n1+n2]: cell (0, i - n1)
        const uint32_t v = row ? i - n1 : i;
        const int32_t* bnd = (row ? brow : bcol) + (size_t)v * BP;
        const uint32_t c = row ? G.idx(0, v) : G.idx(v, 0);
        pl.M()[c] = bnd[0];
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            pl.I(k)[c] = row ? CL_NEG_INF : bnd[1 + k];
            pl.D(k)[c] = row ? bnd[1 + k] : CL_NEG_INF;
        }
    }

    auto ring_cell = [&](uint32_t x, uint32_t y) -> int32_t* {
        const uint32_t d = x + y;
        return ring + ((size_t)(d & mask) * width + (x - G.lo(d))) * PL;
    };
    const uint32_t last = n1 + n2, drain_every = depth >= 2 ? depth / 2 : 1u;
    uint32_t off = 3;   // G.off(2): cells on anti-diagonals 0 and 1
    for (uint32_t d = 2; d <= last; ++d) {
        const uint32_t lo_full = G.lo(d), cnt_full = G.hi(d) - lo_full + 1;
        const uint32_t lo = lo_full ? lo_full : 1u, hi = (d - 1 < n1) ? d - 1 : n1;   // interior cells: a >= 1, b >= 1
        for (uint32_t a = lo + tid; a <= hi; a += BLOCK) {
            const uint32_t b = d - a;
            const uint2 r1 = rec1[a - 1], r2 = rec2[b - 1];
            const uint32_t deg1 = r1.y & 0xFFFFu, deg2 = r2.y & 0xFFFFu;
            const bool src1 = r1.y >> 31, src2 = r2.y >> 31;
            auto cellM = [&](uint32_t x, uint32_t y) -> int32_t { if (d - (x + y) < depth) return ring_cell(x, y)[0]; return pl.M()[G.idx(x, y)]; };
            int32_t M = CL_NEG_INF, I[NPW], D[NPW];
#pragma unroll
            for (int k = 0; k < NPW; ++k) { I[k] = CL_NEG_INF; D[k] = CL_NEG_INF; }
            for (uint32_t e = 0; e < deg1; ++e) {   // insertions: from (p, b)
                const uint32_t pa = deg1 == 1 ? r1.x : pl1[r1.x + e];
                if (d - (pa + b) < depth) {
                    const int32_t* c = ring_cell(pa, b);
                    const int32_t m = c[0];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], imax(m - P.oe[k], c[1 + k] - P.ext[k]));
                } else {
                    const uint32_t c = G.idx(pa, b);
                    const int32_t m = pl.M()[c];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], imax(m - P.oe[k], pl.I(k)[c] - P.ext[k]));
                }
            }
            if (src1) {
                const int32_t m = brow[(size_t)b * BP];
#pragma unroll
                for (int k = 0; k < NPW; ++k) I[k] = imax(I[k], m - P.oe[k]);
            }
            for (uint32_t f = 0; f < deg2; ++f) {   // deletions: from (a, q)
                const uint32_t pb = deg2 == 1 ? r2.x : pl2[r2.x + f];
                if (d - (a + pb) < depth) {
                    const int32_t* c = ring_cell(a, pb);
                    const int32_t m = c[0];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], imax(m - P.oe[k], c[1 + NPW + k] - P.ext[k]));
                } else {
                    const uint32_t c = G.idx(a, pb);
                    const int32_t m = pl.M()[c];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], imax(m - P.oe[k], pl.D(k)[c] - P.ext[k]));
                }
            }
            if (src2) {
                const int32_t m = bcol[(size_t)a * BP];
#pragma unroll
                for (int k = 0; k < NPW; ++k) D[k] = imax(D[k], m - P.oe[k]);
            }
            {   // diagonal: max over predecessor pairs, the boundary index standing in for a source
                const int32_t s = (((r1.y >> 16) & 0x7Fu) == ((r2.y >> 16) & 0x7Fu)) ? P.match : -P.mismatch;
                for (uint32_t e = 0; e < deg1; ++e) {
                    const uint32_t pa = deg1 == 1 ? r1.x : pl1[r1.x + e];
                    for (uint32_t f = 0; f < deg2; ++f) {
                        const uint32_t pb = deg2 == 1 ? r2.x : pl2[r2.x + f];
                        M = imax(M, cellM(pa, pb) + s);
                    }
                    if (src2) M = imax(M, bcol[(size_t)pa * BP] + s);
                }
                if (src1) {
                    for (uint32_t f = 0; f < deg2; ++f) {
                        const uint32_t pb = deg2 == 1 ? r2.x : pl2[r2.x + f];
                        M = imax(M, brow[(size_t)pb * BP] + s);
                    }
                    if (src2) M = imax(M, s);   // the corner counts as 0
                }
            }
            const uint32_t self_idx = off + (a - lo_full);
            int32_t* rc = ring + ((size_t)(d & mask) * width + (a - lo_full)) * PL;
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                M = imax(M, imax(I[k], D[k]));
                pl.I(k)[self_idx] = I[k];
                pl.D(k)[self_idx] = D[k];
                rc[1 + k] = I[k];
                rc[1 + NPW + k] = D[k];
            }
            pl.M()[self_idx] = M;
            rc[0] = M;
        }
        off += cnt_full;
        if (full_cover || (d & (drain_every - 1)) != 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        } else {
            __syncthreads();   // vmcnt(0): cells written depth/2 or more diagonals ago are in memory for the far reads
        }
    }
    __syncthreads();
    if (tid < 64 && !B.skip_traceback) traceback_wave<NPW>(B, pd, G, pl, P, prob);
    cl_tick_end(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// popoa_sys_kernel: any DAG pair whose SHORTER graph has at most BLOCK - 1 nodes, swept as a systolic array like the chain kernel
// (popoa_linear.hip) instead of anti-diagonal by anti-diagonal.  Thread r owns ROW r (rank r of the shorter graph, row 0 = the
// boundary "nothing consumed yet") for the whole sweep and at step t works on column t - r, so the cells a cell needs — (p, c), (r, q),
// (p, q) for predecessors p of its row and q of its column — were finished (r - p), (c - q) and (r - p) + (c - q) steps earlier.  Every
// row keeps its last H columns in an LDS ring, H a power of two above the row graph's predecessor span plus the column graph's NEAR
// predecessor distances; a column that some later column reaches from further away (the fork in front of a bubble whose branches
// differ by a whole repeat unit: one or two per subproblem) is a SAVED column: its cells are also written to a small LDS area of their
// own ([slot][row]) and read from there.  So every read of the sweep is an LDS read, and — this is the point — the loop contains no
// global load at all: on gfx9 loads and stores share one counter, a single load in the loop makes every step wait for the previous
// step's plane stores (≈1 µs); without one the stores are fire-and-forget.  The row's own topology sits in registers, the column's
// arrives as one 8-byte LDS record; up to two predecessors per side (a source's boundary index counted as one) take a straight-line
// path of eight independent LDS reads, more go through loops.
// The int32 planes go to HBM anti-diagonal-major as before: the traceback reads them.
// ClProbDesc::pad: bits 0-4 log2 H, bit 15: graph 2 is the shorter one (rows = graph 2: the roles of I and D swap);
// ClProbDesc::aux_base / aux_cnt: the saved columns.
template <int NPW, int BLOCK>
__global__ void __launch_bounds__(BLOCK) popoa_sys_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist, ClScoreParams P) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    // one cell: NumPW 1: {M, V0, H0, -}; NumPW 2, 3: {M, V0, V1, V2 | M, H0, H1, H2} — what a vertical or a horizontal read needs is one
    // aligned 16-byte LDS read
    constexpr int CW = NPW == 1 ? 4 : 8;
    cl_tick_start(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
    const uint32_t prob = plist[blockIdx.x];
    const ClProbDesc pd = B.desc[prob];
    const DiagGeom G(pd.n1, pd.n2);
    Planes<NPW> pl;
    pl.base = B.planes + pd.plane_base;
    pl.cells = (pd.n1 + 1) * (pd.n2 + 1);
    const uint32_t tid = threadIdx.x;
    const bool swap = pd.pad & 0x8000u;
    const uint32_t logH = pd.pad & 31u, H = 1u << logH, hm = H - 1;
    const uint32_t nR = swap ? pd.n2 : pd.n1, nC = swap ? pd.n1 : pd.n2;
    const uint32_t K = pd.aux_cnt;
    const uint32_t baseR = swap ? pd.node_base[1] : pd.node_base[0], baseC = swap ? pd.node_base[0] : pd.node_base[1];
    const uint32_t* const poffR = (swap ? B.poff[1] : B.poff[0]) + baseR;
    const uint32_t* const poffC = (swap ? B.poff[0] : B.poff[1]) + baseC;
    const uint32_t* const pidxR = swap ? B.pidx[1] : B.pidx[0];
    const uint32_t* const pidxC = swap ? B.pidx[0] : B.pidx[1];
    const uint8_t* const labRp = (swap ? B.lab[1] : B.lab[0]) + baseR;
    const uint8_t* const labCp = (swap ? B.lab[0] : B.lab[1]) + baseC;
    // LDS (int32 units): ring [nR + 1][row stride] | saved columns [K][nR + 1][CW] | column records [nC] (uint2) | row predecessor lists |
    // column predecessor lists | saved column numbers [K].
    // Lane r works on row r at column t - r, so neighbouring lanes' ring addresses differ by (row stride - CW) words: with a pad of 4 (8)
    // words for 8-word (4-word) cells that difference is an ODD multiple of four words — 4(2H - 1), 4(H + 1) — so the sixteen lanes of a
    // 16-byte LDS access fall on sixteen different 4-bank groups
    const uint32_t row_stride = H * CW + (CW == 8 ? 4u : 8u);
    int32_t* const ring = lds;
    int32_t* const saved = ring + (nR + 1) * row_stride;
    uint2* const recC = reinterpret_cast<uint2*>(saved + K * (nR + 1) * CW);
    const uint32_t eR0 = poffR[0], eR1 = poffR[nR], eC0 = poffC[0], eC1 = poffC[nC];
    uint32_t* const plR = reinterpret_cast<uint32_t*>(recC + nC);
    uint32_t* const plC = plR + (eR1 - eR0);
    uint32_t* const saved_col = plC + (eC1 - eC0);
    for (uint32_t i = tid; i < eR1 - eR0; i += BLOCK) plR[i] = pidxR[eR0 + i];
    for (uint32_t i = tid; i < eC1 - eC0; i += BLOCK) plC[i] = pidxC[eC0 + i];
    const uint32_t near_limit = B.aux[pd.aux_base];
    for (uint32_t i = tid; i < K; i += BLOCK) saved_col[i] = B.aux[pd.aux_base + 1 + i];
    __syncthreads();
    // column record, 8 bytes.  x: the straight-line cell's two predecessor columns, 12 bits each — bit 11 clear: the column is that many
    // columns back, in the ring; set: it is saved column number (low bits) — a source's boundary column 0 counted as a predecessor, a
    // missing second one repeating the first | bit 26: this column is itself saved, bits 27-31: in that slot.
    // y: list start (17 bits) | in-degree (6 bits) | bit 23: the straight-line cell applies (1 or 2 predecessors) | label (7 bits) | source
    auto slot_of = [&](uint32_t col) { uint32_t k = 0; while (k + 1 < K && saved_col[k] != col) ++k; return k; };
    auto pred_code = [&](uint32_t c, uint32_t q) { return c - q > near_limit ? 0x800u | slot_of(q) : c - q; };
    for (uint32_t i = tid; i < nC; i += BLOCK) {
        const uint32_t c = i + 1, b0 = poffC[i] - eC0, deg = poffC[i + 1] - poffC[i], l = labCp[i], src = l >> 7, nq = deg + src;
        uint32_t x = 0;
        if (nq >= 1 && nq <= 2) {
            const uint32_t q0 = deg ? plC[b0] : 0u, q1 = deg == 2 ? plC[b0 + 1] : (src ? 0u : q0);
            x = pred_code(c, q0) | (pred_code(c, q1) << 12);
        }
        bool is_saved = false;
        for (uint32_t k = 0; k < K; ++k) is_saved |= saved_col[k] == c;
        if (is_saved) x |= (1u << 26) | (slot_of(c) << 27);
        recC[i] = make_uint2(x, b0 | (deg << 17) | ((nq >= 1 && nq <= 2) ? 1u << 23 : 0u) | ((l & 0x7Fu) << 24) | (src << 31));
    }
    const bool save_col0 = K && saved_col[0] == 0;   // the list is ascending
    // this thread's row (rows beyond nR idle)
    const uint32_t r = tid;
    uint32_t degR = 0, firstR = 0, labR = 0;
    bool srcR = false;
    if (r >= 1 && r <= nR) {
        const uint32_t l = labRp[r - 1];
        degR = poffR[r] - poffR[r - 1];
        firstR = poffR[r - 1] - eR0;
        labR = l & 0x7Fu;
        srcR = l >> 7;
    }
    // up to two predecessors per side, a source's boundary index (row / column 0) counted as one: the straight-line cell below covers them
    // with no data-dependent branch (a missing second predecessor repeats the first: the maxima do not care).  The boundary row 0 takes
    // the same path with "predecessor rows" 0, its M and V forced to -inf afterwards
    const bool fastR = r == 0 || (r <= nR && degR + (srcR ? 1u : 0u) <= 2 && degR + (srcR ? 1u : 0u) >= 1);
    uint32_t rp0 = 0, rp1 = 0;
    if (r && fastR && degR) {
        rp0 = plR[firstR];
        rp1 = degR == 2 ? plR[firstR + 1] : (srcR ? 0u : rp0);
    }
    const uint32_t rp0_rs = rp0 * row_stride, rp1_rs = rp1 * row_stride, r_rs = r * row_stride, rp0_cw = rp0 * CW, rp1_cw = rp1 * CW, r_cw = r * CW;
    __syncthreads();
    const uint32_t last = nR + nC;
    uint32_t off = 0;   // G.off(t): cells on the anti-diagonals before t
    uint32_t t = 0;
    // where cell (row, col) lives: in the row's ring while the row has not moved H columns past col (at step t it is at column t - row),
    // else in the saved area (the host saved every column that is read from further away)
    auto where = [&](uint32_t row, uint32_t col) -> const int32_t* {
        if (t - row - col < H) return ring + (row * row_stride + (col & hm) * CW);
        return saved + (slot_of(col) * (nR + 1) + row) * CW;
    };
    // one cell: NumPW 1: {M, V0, H0, M'}; NumPW 2, 3: {M, V0, V1, V2 | M', H0, H1, H2}.  M' is what a HORIZONTAL step reads as the cell's M:
    // M itself, except on the boundary row, where it is -inf (the row only extends its gap: alignment.hpp:864-877) and 0 at the corner
    // (where a source column opens it); the corner's M reads 0 as well (:814-818)
    auto get_mv = [&](const int32_t* cell, int32_t& m, int32_t (&v)[NPW]) {   // M and the vertical gap values of a cell
        const int4 x = reinterpret_cast<const int4*>(cell)[0];
        m = x.x; v[0] = x.y;
        if (NPW > 1) v[1] = x.z;
        if (NPW > 2) v[2] = x.w;
    };
    auto get_mh = [&](const int32_t* cell, int32_t& m, int32_t (&h)[NPW]) {   // M' and the horizontal gap values
        if (NPW == 1) { const int4 x = reinterpret_cast<const int4*>(cell)[0]; m = x.w; h[0] = x.z; }
        else {
            const int4 x = reinterpret_cast<const int4*>(cell)[1];
            m = x.x; h[0] = x.y;
            if (NPW > 1) h[1] = x.z;
            if (NPW > 2) h[2] = x.w;
        }
    };
    auto get_m = [&](uint32_t row, uint32_t col) -> int32_t { return where(row, col)[0]; };
    const uint32_t saved_off = (uint32_t)(saved - ring), slot_stride = (nR + 1) * CW;
    int32_t* const my_row = ring + r * row_stride;
    int32_t* const plane0 = pl.M();
    const size_t plane_stride = pl.cells;
    // the planes in graph-1 / graph-2 terms (I consumes a graph-1 node, D a graph-2 node): which of them take the row gaps and which the column gaps is the
    // same for the whole launch, so the pointers are chosen once and a store is a uniform base plus one 32-bit index
    g_i32* pV[NPW];
    g_i32* pH[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        pV[k] = uniform_plane(plane0 + (size_t)(swap ? 1 + NPW + k : 1 + k) * plane_stride);
        pH[k] = uniform_plane(plane0 + (size_t)(swap ? 1 + k : 1 + NPW + k) * plane_stride);
    }
    g_i32* const pM = uniform_plane(plane0);
    for (; t <= last; ++t) {
        const uint32_t lo_d = G.lo(t), cnt_d = G.hi(t) - lo_d + 1;
        if (r <= nR && t >= r && t - r <= nC) {
            const uint32_t c = t - r;
            int32_t M = CL_NEG_INF, V[NPW], Hh[NPW];
#pragma unroll
            for (int k = 0; k < NPW; ++k) { V[k] = CL_NEG_INF; Hh[k] = CL_NEG_INF; }
            uint32_t degC = 0, firstC = 0, labC = 0, keep_slot = 0;
            bool srcC = false, keep = !c && save_col0, fastC = false;
            uint2 rc = make_uint2(0, 0);
            if (c) {
                rc = recC[c - 1];
                firstC = rc.y & 0x1FFFFu; degC = (rc.y >> 17) & 63u; fastC = (rc.y >> 23) & 1u; labC = (rc.y >> 24) & 0x7Fu; srcC = rc.y >> 31;
                keep = (rc.x >> 26) & 1u; keep_slot = rc.x >> 27;
            }
            if (c && fastC && fastR) {
                // the usual cell, straight-line: two predecessors per side (the second may repeat the first), eight independent LDS reads
                // at addresses that are selects and multiply-adds of the column record
                // (round 4: the row parts of the addresses are per-thread constants selected by the code's flag — six quarter-rate 32-bit multiplies and two
                // 64-bit multiply-adds per step before — and the slot offset is a 24-bit multiply)
                const uint32_t e0 = rc.x & 0xFFFu, e1 = (rc.x >> 12) & 0xFFFu;
                const bool f0 = e0 & 0x800u, f1 = e1 & 0x800u;
                const uint32_t o0 = f0 ? saved_off + __umul24(e0 & 0x7FFu, slot_stride) : ((c - e0) & hm) * CW;
                const uint32_t o1 = f1 ? saved_off + __umul24(e1 & 0x7FFu, slot_stride) : ((c - e1) & hm) * CW;
                const uint32_t oc = (c & hm) * CW;
                int32_t mv0, mv1, mh0, mh1, vv0[NPW], vv1[NPW], hh0[NPW], hh1[NPW];
                get_mv(ring + (rp0_rs + oc), mv0, vv0);
                get_mv(ring + (rp1_rs + oc), mv1, vv1);
                get_mh(ring + (o0 + (f0 ? r_cw : r_rs)), mh0, hh0);
                get_mh(ring + (o1 + (f1 ? r_cw : r_rs)), mh1, hh1);
                const int32_t d00 = ring[o0 + (f0 ? rp0_cw : rp0_rs)], d01 = ring[o1 + (f1 ? rp0_cw : rp0_rs)];
                const int32_t d10 = ring[o0 + (f0 ? rp1_cw : rp1_rs)], d11 = ring[o1 + (f1 ? rp1_cw : rp1_rs)];
                M = imax(imax(d00, d01), imax(d10, d11)) + ((labR == labC) ? P.match : -P.mismatch);
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    V[k] = imax(imax(mv0 - P.oe[k], vv0[k] - P.ext[k]), imax(mv1 - P.oe[k], vv1[k] - P.ext[k]));
                    Hh[k] = imax(imax(mh0 - P.oe[k], hh0[k] - P.ext[k]), imax(mh1 - P.oe[k], hh1[k] - P.ext[k]));
                }
                if (!r) {   // the boundary row has no M and no vertical gap (what was read for them is its own unwritten cell)
                    M = CL_NEG_INF;
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = CL_NEG_INF;
                }
            } else if (r && !c) {          // boundary column: gap extensions down the row graph (alignment.hpp:832-845)
                for (uint32_t e = 0; e < degR; ++e) {
                    const uint32_t p = plR[firstR + e];
                    int32_t m, vv[NPW];
                    get_mv(where(p, 0), m, vv);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], vv[k] - P.ext[k]);
                }
                if (srcR) {
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], -P.oe[k]);
                }
            } else if (!r && c) {   // boundary row (:864-877)
                for (uint32_t f = 0; f < degC; ++f) {
                    const uint32_t q = plC[firstC + f];
                    int32_t m, hh[NPW];
                    get_mh(where(0, q), m, hh);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], hh[k] - P.ext[k]);
                }
                if (srcC) {
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], -P.oe[k]);
                }
            } else if (r && c) {    // interior, any degrees (:897-938 in pull form, see compute_cell)
                for (uint32_t e = 0; e < degR; ++e) {
                    const uint32_t p = plR[firstR + e];
                    int32_t m, vv[NPW];
                    get_mv(where(p, c), m, vv);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], imax(m - P.oe[k], vv[k] - P.ext[k]));
                }
                if (srcR) {
                    const int32_t m = get_m(0, c);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], m - P.oe[k]);
                }
                const int32_t s = (labR == labC) ? P.match : -P.mismatch;
                for (uint32_t f = 0; f < degC; ++f) {
                    const uint32_t q = plC[firstC + f];
                    int32_t m, hh[NPW];
                    get_mh(where(r, q), m, hh);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], imax(m - P.oe[k], hh[k] - P.ext[k]));
                    for (uint32_t e = 0; e < degR; ++e) {
                        const uint32_t p = plR[firstR + e];
                        M = imax(M, get_m(p, q) + s);
                    }
                    if (srcR) M = imax(M, get_m(0, q) + s);
                }
                if (srcC) {
                    const int32_t m = get_m(r, 0);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], m - P.oe[k]);
                    for (uint32_t e = 0; e < degR; ++e) {
                        const uint32_t p = plR[firstR + e];
                        M = imax(M, get_m(p, 0) + s);
                    }
                    if (srcR) M = imax(M, s);   // the corner counts as 0 (:814-818)
                }
            }
#pragma unroll
            for (int k = 0; k < NPW; ++k) M = imax(M, imax(V[k], Hh[k]));
            // the LDS copy: the corner's M reads 0, the boundary row's M' -inf (see get_mh)
            const int32_t Ml = (r | c) ? M : 0, Mh = r ? M : (c ? CL_NEG_INF : 0);
            const int4 w0 = NPW == 1 ? make_int4(Ml, V[0], Hh[0], Mh) : make_int4(Ml, V[0], V[NPW > 1 ? 1 : 0], V[NPW > 2 ? 2 : 0]);
            const int4 w1 = make_int4(Mh, Hh[0], Hh[NPW > 1 ? 1 : 0], Hh[NPW > 2 ? 2 : 0]);
            int4* w = reinterpret_cast<int4*>(my_row + (c & hm) * CW);
            w[0] = w0;
            if (NPW > 1) w[1] = w1;
            if (keep) {   // a saved column: its cells stay available for the far reads
                int4* sw = reinterpret_cast<int4*>(saved + ((c ? keep_slot : 0u) * (nR + 1) + r) * CW);
                sw[0] = w0;
                if (NPW > 1) sw[1] = w1;
            }
            const uint32_t pidx = off + ((swap ? c : r) - lo_d);
            if (!(B.skip_traceback & 2)) {
                const uint32_t pb = pidx * 4u;
                plane_store(pM, pb, M);
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    plane_store(pV[k], pb, V[k]);
                    plane_store(pH[k], pb, Hh[k]);
                }
            }
        }
        off += cnt_d;
        if (BLOCK > 64) {   // LDS traffic only: nothing in the sweep reads the planes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    }
    __syncthreads();   // vmcnt(0): every plane value is in memory
    if (tid < 64 && !B.skip_traceback) traceback_wave<NPW>(B, pd, G, pl, P, prob);
    cl_tick_end(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// popoa_strip_kernel: the systolic sweep of popoa_sys_kernel for branching pairs whose rows do NOT fit one workgroup's LDS (round 4; SURVEY §7
// step 5; the reference's ceiling is 40 M cells per pair, src/parameters.cpp:79).  The rows are cut into STRIPS of S consecutive rows (S a multiple
// of 64), one workgroup per strip, all strips of a pair in flight at once on different compute units: strip j runs behind strip j - 1 and reads
// that strip's last rows — every row one of its own rows has a predecessor in: its GHOST rows — from a hand-off area in HBM which strip j - 1
// fills as it goes.  Waves 0-3 of a workgroup hold the ghost rows and nothing else: lane i of wave w stands for ghost row i at the steps t = w
// (mod 4): it puts the cell it asked for four steps earlier into the row's LDS ring — exactly when a computing row would have written it — and
// asks for the cell four columns on.  A wave's loads so have four steps (2-3 us) to arrive and wait for nothing but one another: no register
// rotation, no stores in these waves.  The computing rows start at wave 4 and are the sys kernel's, with two differences: the column records (16 bytes, written by the host: the straight-line cell's two predecessor distances, and up to six
// predecessor distances for the general cell, so that no cell of the sweep needs a global load) pass through a small LDS ring which wave 0
// refills sixteen columns at a time, thirty-odd columns ahead of the first row — 16 bytes x columns do not fit LDS beside the rings .  Columns that a later column reaches from further away than the ring holds (the fork in front of a long bubble; column 0 for a
// late source) are SAVED columns as in the sys kernel, at most eight per pair: their cells are also written to an LDS area of their own ([slot][row], ghost
// rows included) and read from there.  The hand-off cells are written and read as 64-bit write-through / cache-bypassing atomics (agent scope, relaxed), and the
// producer publishes "columns done" after a vmcnt(0) of the one wave that holds all its hand-off rows: the consumer polls that word before it
// asks for a column, so no fence is ever needed while the sweep runs.  At its end a strip makes its planes visible (one fence) and marks itself
// done; the last strip waits for all marks and runs the traceback.  Every wait is bounded: a strip that gives up marks itself failed, the marks
// propagate, and the pair reports status 9.
constexpr uint32_t kStripDone = 0x7FFFFFFFu, kStripFailed = 0xFFFFFFFFu;

template <int NPW>
__global__ void __launch_bounds__(1024) popoa_strip_kernel(ClDeviceBatch B, ClStripDevice SD, const uint32_t* __restrict__ slist, ClScoreParams P) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    constexpr int CW = NPW == 1 ? 4 : 8;
    constexpr int WW = CW / 2;   // 64-bit words of a cell
    constexpr uint32_t GW = 4;   // ghost waves = steps a hand-off cell has to arrive
    cl_tick_start(B, true);
    const ClStripDesc sd = SD.strips[slist[blockIdx.x]];
    const uint32_t prob = sd.prob;
    const ClProbDesc pd = B.desc[prob];
    const DiagGeom G(pd.n1, pd.n2);
    Planes<NPW> pl;
    pl.base = B.planes + pd.plane_base;
    pl.cells = (pd.n1 + 1) * (pd.n2 + 1);
    const uint32_t tid = threadIdx.x;
    const bool swap = pd.pad & 0x8000u;
    const uint32_t H = 1u << sd.logH, hm = H - 1;
    const uint32_t nC = swap ? pd.n1 : pd.n2;
    const uint32_t baseR = swap ? pd.node_base[1] : pd.node_base[0], baseC = swap ? pd.node_base[0] : pd.node_base[1];
    const uint32_t* const poffR = (swap ? B.poff[1] : B.poff[0]) + baseR;
    const uint32_t* const poffC = (swap ? B.poff[0] : B.poff[1]) + baseC;
    const uint32_t* const pidxR = swap ? B.pidx[1] : B.pidx[0];
    const uint32_t* const pidxC = swap ? B.pidx[0] : B.pidx[1];
    const uint8_t* const labRp = (swap ? B.lab[1] : B.lab[0]) + baseR;
    const uint4* const recs = SD.recs + sd.rec_base;
    const uint32_t g = sd.n_ghost, S = sd.n_real, n_loc = g + S;
    const bool is_ghost = tid < 64 * GW;
    const uint32_t gw = tid >> 6, gi = tid & 63u;
    const uint32_t L = is_ghost ? gi : g + (tid - 64 * GW);       // timing index: this row works on column t - L at step t
    const bool live = is_ghost ? gi < g : (tid - 64 * GW) < S;
    const uint32_t r = sd.row_base + L;                           // matrix row
    const uint32_t row_stride = H * CW + (CW == 8 ? 4u : 8u);     // (see popoa_sys_kernel: the lanes' 16-byte accesses fall on different bank groups)
    int32_t* const ring = lds;
    const uint32_t K = pd.aux_cnt;                                                  // saved columns
    int32_t* const saved = ring + n_loc * row_stride;                               // [K][n_loc][CW]
    const uint32_t saved_off = n_loc * row_stride, slot_stride = n_loc * CW;
    const uint32_t rmask = (1u << sd.logRW) - 1;
    uint4* const rec_ring = reinterpret_cast<uint4*>(saved + K * slot_stride);     // record of column c in slot (c - 1) & rmask
    uint32_t* const plR = reinterpret_cast<uint32_t*>(rec_ring + (rmask + 1));
    const uint32_t r_lo = sd.row_base + g, r_hi = sd.row_base + n_loc - 1;   // computing rows
    const uint32_t node_lo = r_lo ? r_lo : 1u;
    const uint32_t eR0 = poffR[node_lo - 1], eR1 = r_hi >= node_lo ? poffR[r_hi] : eR0;
    for (uint32_t i = tid; i < eR1 - eR0; i += blockDim.x) plR[i] = pidxR[eR0 + i] - sd.row_base;   // as timing indices
    uint32_t* const saved_col = plR + (eR1 - eR0);
    for (uint32_t i = tid; i < K; i += blockDim.x) saved_col[i] = B.aux[pd.aux_base + 1 + i];
    const bool save_col0 = K && B.aux[pd.aux_base + 1] == 0;   // the list is ascending
    int* const fail_flag = reinterpret_cast<int*>(saved_col + K);
    if (tid == 0) *fail_flag = 0;
    for (uint32_t i = tid; i < 32 && i < nC; i += blockDim.x) rec_ring[i & rmask] = recs[i];   // columns 1 .. 32; the loop below goes on from 33
    // this thread's row
    uint32_t degR = 0, firstR = 0, labR = 0;
    bool srcR = false;
    if (!is_ghost && live && r >= 1) {
        const uint32_t l = labRp[r - 1];
        degR = poffR[r] - poffR[r - 1];
        firstR = poffR[r - 1] - eR0;
        labR = l & 0x7Fu;
        srcR = l >> 7;
    }
    __syncthreads();
    // the straight-line cell takes up to THREE predecessors per side here (a source's boundary index counted as one, missing ones repeating the
    // first): fifteen independent LDS reads; graphs with many three-way joins would otherwise send a row through the loops at every step
    const bool fastR = !is_ghost && live && (r == 0 || (degR + (srcR ? 1u : 0u) <= 3 && degR + (srcR ? 1u : 0u) >= 1));
    uint32_t rp0 = 0, rp1 = 0, rp2 = 0;
    if (r && fastR && degR) {
        rp0 = plR[firstR];
        rp1 = degR >= 2 ? plR[firstR + 1] : (srcR ? 0u : rp0);
        rp2 = degR == 3 ? plR[firstR + 2] : (degR == 2 && srcR ? 0u : rp0);
    }
    const uint32_t rp_rs[3] = {rp0 * row_stride, rp1 * row_stride, rp2 * row_stride}, rp_cw[3] = {rp0 * CW, rp1 * CW, rp2 * CW}, L_rs = L * row_stride, L_cw = L * CW;
    // predecessor f of column c: up to six distances ride in the record (z, w: twelve bits each, in list order), longer lists stay in HBM
    // (the record's words are passed one by one: a reference to the uint4 made the compiler index it in scratch memory — a scratch store and several loads per step)
    auto col_pred4 = [&](uint32_t rx, uint32_t ry, uint32_t rz, uint32_t rw, uint32_t c, uint32_t fc, uint32_t f) -> uint32_t {
        if (!((ry >> 16) & 1u)) return pidxC[fc + f];
        uint32_t word = rz;
        word = f >= 2 ? rw : word;
        word = f >= 4 ? rx : word;
        const uint32_t e = (word >> (12 * (f & 1u))) & 0xFFFu;
        return (e & 0x80u) ? saved_col[e & 0x7Fu] : c - e;      // bit 7: saved column number (low bits), else that many columns back
    };
    // up to four row predecessors in registers (a missing one repeats the first: the maxima do not care), so that the general cell's reads do not
    // hang on list reads
    uint32_t rpl[4] = {0, 0, 0, 0};
    const bool few_rows = degR >= 1 && degR <= 4;
    if (!is_ghost && live && few_rows) {
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) rpl[e] = plR[firstR + (e < degR ? e : 0u)];
    }
    auto slot_of = [&](uint32_t col) { uint32_t s2 = 0; while (s2 + 1 < K && saved_col[s2] != col) ++s2; return s2; };
    // where cell (row with timing index row_l, column col) lives at step t: in the row's ring while that row has not moved H columns past col, else in the
    // saved area (the host saved every column that is read from further away)
    auto cell_at_t = [&](uint32_t t, uint32_t row_l, uint32_t col) -> const int32_t* {
        if (t - row_l - col < H) return ring + (row_l * row_stride + (col & hm) * CW);
        return saved + (slot_of(col) * n_loc + row_l) * CW;
    };
    auto get_mv = [&](const int32_t* cell, int32_t& m, int32_t (&v)[NPW]) {
        const int4 x = reinterpret_cast<const int4*>(cell)[0];
        m = x.x; v[0] = x.y;
        if (NPW > 1) v[1] = x.z;
        if (NPW > 2) v[2] = x.w;
    };
    auto get_mh = [&](const int32_t* cell, int32_t& m, int32_t (&h)[NPW]) {
        if (NPW == 1) { const int4 x = reinterpret_cast<const int4*>(cell)[0]; m = x.w; h[0] = x.z; }
        else {
            const int4 x = reinterpret_cast<const int4*>(cell)[1];
            m = x.x; h[0] = x.y;
            if (NPW > 1) h[1] = x.z;
            if (NPW > 2) h[2] = x.w;
        }
    };
    int32_t* const my_row = ring + L * row_stride;
    int32_t* const plane0 = pl.M();
    const size_t plane_stride = pl.cells;
    g_i32* pV[NPW];
    g_i32* pH[NPW];   // (see popoa_sys_kernel: the planes that take the row gaps / the column gaps, chosen once; scalar base + 32-bit offset per store)
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        pV[k] = uniform_plane(plane0 + (size_t)(swap ? 1 + NPW + k : 1 + k) * plane_stride);
        pH[k] = uniform_plane(plane0 + (size_t)(swap ? 1 + k : 1 + NPW + k) * plane_stride);
    }
    g_i32* const pM = uniform_plane(plane0);
    const uint32_t last = n_loc - 1 + nC;
    uint32_t off = G.off(sd.row_base);   // cells on the anti-diagonals before the one of step t (anti-diagonal row_base + t)
    // ---- ghost rows: the hand-off pipeline ----
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;   // (scalars, not an array passed by reference: that one lived in scratch memory)
    const unsigned long long* const hin = SD.handoff + sd.hand_in + (size_t)gi * (nC + 1) * WW;
    uint32_t* const prog_in = SD.progress + (sd.prog - 1);   // the strip in front (ghost lanes only: strip > 0)
    uint32_t avail = 0, avail_next = 0;                       // columns of the incoming rows known to be complete; the word as asked for at the last active step
    bool failed = false;
    auto fetch = [&](uint32_t col) {
        if (col >= avail) {
            unsigned spins = 0;
            while (true) {
                avail = __hip_atomic_load(prog_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (avail > col) break;
                if (++spins > (1u << 20)) { failed = true; avail = kStripFailed; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (avail == kStripFailed) failed = true;
        }
        const unsigned long long* src = hin + (size_t)col * WW;
        st0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (WW > 2) {
            st2 = __hip_atomic_load(src + (WW > 2 ? 2 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st3 = __hip_atomic_load(src + (WW > 2 ? 3 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (is_ghost && live && gw >= L && gw - L <= nC) fetch(gw - L);   // what this lane delivers at its first step, t = gw
    // ---- wave 0, lanes 0-15: the column records, sixteen at a time (asked for at t = 0 (mod 16), stored four steps later) ----
    uint4 rec_reg = make_uint4(0, 0, 0, 0);
    uint32_t rec_col = 0;   // column whose record rec_reg holds (0: none)
    const uint32_t out_first = S - sd.n_out;   // computing rows from here on are handed to the next strip
    unsigned long long* const hout = SD.handoff + sd.hand_out;
    for (uint32_t t = 0; t <= last; ++t) {
        auto cell_at = [&](uint32_t row_l, uint32_t col) -> const int32_t* { return cell_at_t(t, row_l, col); };
        const uint32_t d = sd.row_base + t;
        const uint32_t lo_d = G.lo(d), cnt_d = G.hi(d) - lo_d + 1;
        if (is_ghost) {
            if ((t & (GW - 1)) == gw && live) {
                if (t >= L && t - L <= nC) {
                    const uint32_t c = t - L;
                    int4* w = reinterpret_cast<int4*>(my_row + (c & hm) * CW);
                    w[0] = make_int4((int)(unsigned)st0, (int)(unsigned)(st0 >> 32), (int)(unsigned)st1, (int)(unsigned)(st1 >> 32));
                    if (NPW > 1) w[1] = make_int4((int)(unsigned)st2, (int)(unsigned)(st2 >> 32), (int)(unsigned)st3, (int)(unsigned)(st3 >> 32));
                    if (K) {   // a saved column keeps its cells of the ghost rows too
                        const uint32_t ky = c ? rec_ring[(c - 1) & rmask].y : (save_col0 ? 0x8000u : 0u);
                        if (ky & 0x8000u) {
                            int4* sw = reinterpret_cast<int4*>(saved + (((ky >> 12) & 7u) * n_loc + L) * CW);
                            sw[0] = w[0];
                            if (NPW > 1) sw[1] = w[1];
                        }
                    }
                }
                // the progress word asked for four steps ago has arrived with the cell: a blocking poll (a round trip to L2 with the whole workgroup waiting
                // at the barrier) is left for the case that the strip in front really is not there yet
                if (avail_next == kStripFailed) failed = true;
                avail = avail_next > avail ? avail_next : avail;
                const uint32_t tn = t + GW;
                if (tn >= L && tn - L <= nC) fetch(tn - L);
                avail_next = __hip_atomic_load(prog_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (gw == 0 && gi < 16 && (t & (GW - 1)) == 0) {
                if (rec_col) { rec_ring[(rec_col - 1) & rmask] = rec_reg; rec_col = 0; }
                if ((t & 15u) == 0 && t + 33 + gi <= nC) { rec_col = t + 33 + gi; rec_reg = recs[rec_col - 1]; }
            }
        } else if (live) {
            if (t >= L && t - L <= nC) {
            const uint32_t c = t - L;
            const uint4 rc = c ? rec_ring[(c - 1) & rmask] : make_uint4(0, 0, 0, 0);
            int32_t M = CL_NEG_INF, V[NPW], Hh[NPW];
#pragma unroll
            for (int k = 0; k < NPW; ++k) { V[k] = CL_NEG_INF; Hh[k] = CL_NEG_INF; }
            uint32_t degC = 0, labC = 0;
            bool srcC = false, fastC = false, inlineC = false;
            if (c) { degC = (rc.y >> 17) & 63u; fastC = (rc.y >> 23) & 1u; labC = (rc.y >> 24) & 0x7Fu; srcC = rc.y >> 31; inlineC = (rc.y >> 16) & 1u; }
            if (c && fastC && fastR) {
                const uint32_t e0 = rc.x & 0xFFu, e1 = (rc.x >> 8) & 0xFFu, e2 = (rc.x >> 16) & 0xFFu;
                // bit 7 of a predecessor code: saved column number (low bits) instead of a distance — a cell of row x then sits at saved_off + slot * slot_stride + x * CW
                const bool f0 = e0 & 0x80u, f1 = e1 & 0x80u, f2 = e2 & 0x80u;
                const uint32_t o0 = f0 ? saved_off + __umul24(e0 & 0x7Fu, slot_stride) : ((c - e0) & hm) * CW;
                const uint32_t o1 = f1 ? saved_off + __umul24(e1 & 0x7Fu, slot_stride) : ((c - e1) & hm) * CW;
                const uint32_t o2 = f2 ? saved_off + __umul24(e2 & 0x7Fu, slot_stride) : ((c - e2) & hm) * CW, oc = (c & hm) * CW;
                const uint32_t b0 = rp_rs[0], b1 = rp_rs[1], b2 = rp_rs[2];
                int32_t mv0, mv1, mv2, mh0, mh1, mh2, vv0[NPW], vv1[NPW], vv2[NPW], hh0[NPW], hh1[NPW], hh2[NPW];
                get_mv(ring + (b0 + oc), mv0, vv0);
                get_mv(ring + (b1 + oc), mv1, vv1);
                get_mv(ring + (b2 + oc), mv2, vv2);
                // (the row parts of the addresses are per-thread constants selected by the code's flag: no multiplies in the step)
                get_mh(ring + (o0 + (f0 ? L_cw : L_rs)), mh0, hh0);
                get_mh(ring + (o1 + (f1 ? L_cw : L_rs)), mh1, hh1);
                get_mh(ring + (o2 + (f2 ? L_cw : L_rs)), mh2, hh2);
                const int32_t d00 = ring[o0 + (f0 ? rp_cw[0] : rp_rs[0])], d01 = ring[o1 + (f1 ? rp_cw[0] : rp_rs[0])], d02 = ring[o2 + (f2 ? rp_cw[0] : rp_rs[0])];
                const int32_t d10 = ring[o0 + (f0 ? rp_cw[1] : rp_rs[1])], d11 = ring[o1 + (f1 ? rp_cw[1] : rp_rs[1])], d12 = ring[o2 + (f2 ? rp_cw[1] : rp_rs[1])];
                const int32_t d20 = ring[o0 + (f0 ? rp_cw[2] : rp_rs[2])], d21 = ring[o1 + (f1 ? rp_cw[2] : rp_rs[2])], d22 = ring[o2 + (f2 ? rp_cw[2] : rp_rs[2])];
                M = imax(imax(imax(imax(d00, d01), imax(d10, d11)), imax(imax(d02, d12), imax(d20, d21))), d22) + ((labR == labC) ? P.match : -P.mismatch);
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    V[k] = imax(imax(imax(mv0 - P.oe[k], vv0[k] - P.ext[k]), imax(mv1 - P.oe[k], vv1[k] - P.ext[k])), imax(mv2 - P.oe[k], vv2[k] - P.ext[k]));
                    Hh[k] = imax(imax(imax(mh0 - P.oe[k], hh0[k] - P.ext[k]), imax(mh1 - P.oe[k], hh1[k] - P.ext[k])), imax(mh2 - P.oe[k], hh2[k] - P.ext[k]));
                }
                if (!r) {
                    M = CL_NEG_INF;
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = CL_NEG_INF;
                }
            } else if (r && !c) {          // boundary column (alignment.hpp:832-845)
                for (uint32_t e = 0; e < degR; ++e) {
                    int32_t m, vv[NPW];
                    get_mv(cell_at(plR[firstR + e], 0), m, vv);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], vv[k] - P.ext[k]);
                }
                if (srcR) {
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], -P.oe[k]);
                }
            } else if (!r && c) {   // boundary row (:864-877)
                const uint32_t fc = inlineC ? 0u : poffC[c - 1];
                for (uint32_t f = 0; f < degC; ++f) {
                    int32_t m, hh[NPW];
                    get_mh(cell_at(0, col_pred4(rc.x, rc.y, rc.z, rc.w, c, fc, f)), m, hh);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], hh[k] - P.ext[k]);
                }
                if (srcC) {
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], -P.oe[k]);
                }
            } else if (r && c && (few_rows || !degR)) {    // interior, up to four row predecessors: every read of a column predecessor's cells at once
                if (degR) {
                    int32_t m[4], vv[4][NPW];
#pragma unroll
                    for (int e = 0; e < 4; ++e) get_mv(cell_at(rpl[e], c), m[e], vv[e]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], imax(m[e] - P.oe[k], vv[e][k] - P.ext[k]));
                }
                if (srcR) {
                    const int32_t m = cell_at(0, c)[0];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], m - P.oe[k]);
                }
                const int32_t sc = (labR == labC) ? P.match : -P.mismatch;
                const uint32_t fc = inlineC ? 0u : poffC[c - 1];
                for (uint32_t f = 0; f < degC; ++f) {
                    const uint32_t q = col_pred4(rc.x, rc.y, rc.z, rc.w, c, fc, f);
                    int32_t m, hh[NPW];
                    get_mh(cell_at(L, q), m, hh);
                    const int32_t d0 = degR ? cell_at(rpl[0], q)[0] : CL_NEG_INF, d1 = degR ? cell_at(rpl[1], q)[0] : CL_NEG_INF;
                    const int32_t d2 = degR ? cell_at(rpl[2], q)[0] : CL_NEG_INF, d3 = degR ? cell_at(rpl[3], q)[0] : CL_NEG_INF;
                    const int32_t d4 = srcR ? cell_at(0, q)[0] : CL_NEG_INF;
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], imax(m - P.oe[k], hh[k] - P.ext[k]));
                    M = imax(M, imax(imax(imax(d0, d1), imax(d2, d3)), d4) + sc);
                }
                if (srcC) {
                    const int32_t m = cell_at(L, 0)[0];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], m - P.oe[k]);
                    for (uint32_t e = 0; e < degR; ++e) M = imax(M, cell_at(plR[firstR + e], 0)[0] + sc);
                    if (srcR) M = imax(M, sc);   // the corner counts as 0 (:814-818)
                }
            } else if (r && c) {    // interior, any degrees (:897-938 in pull form)
                for (uint32_t e = 0; e < degR; ++e) {
                    int32_t m, vv[NPW];
                    get_mv(cell_at(plR[firstR + e], c), m, vv);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], imax(m - P.oe[k], vv[k] - P.ext[k]));
                }
                if (srcR) {
                    const int32_t m = cell_at(0, c)[0];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], m - P.oe[k]);
                }
                const int32_t sc = (labR == labC) ? P.match : -P.mismatch;
                const uint32_t fc = inlineC ? 0u : poffC[c - 1];
                for (uint32_t f = 0; f < degC; ++f) {
                    const uint32_t q = col_pred4(rc.x, rc.y, rc.z, rc.w, c, fc, f);
                    int32_t m, hh[NPW];
                    get_mh(cell_at(L, q), m, hh);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], imax(m - P.oe[k], hh[k] - P.ext[k]));
                    for (uint32_t e = 0; e < degR; ++e) M = imax(M, cell_at(plR[firstR + e], q)[0] + sc);
                    if (srcR) M = imax(M, cell_at(0, q)[0] + sc);
                }
                if (srcC) {
                    const int32_t m = cell_at(L, 0)[0];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) Hh[k] = imax(Hh[k], m - P.oe[k]);
                    for (uint32_t e = 0; e < degR; ++e) M = imax(M, cell_at(plR[firstR + e], 0)[0] + sc);
                    if (srcR) M = imax(M, sc);   // the corner counts as 0 (:814-818)
                }
            }
#pragma unroll
            for (int k = 0; k < NPW; ++k) M = imax(M, imax(V[k], Hh[k]));
            const int32_t Ml = (r | c) ? M : 0, Mh = r ? M : (c ? CL_NEG_INF : 0);
            const int4 w0 = NPW == 1 ? make_int4(Ml, V[0], Hh[0], Mh) : make_int4(Ml, V[0], V[NPW > 1 ? 1 : 0], V[NPW > 2 ? 2 : 0]);
            const int4 w1 = make_int4(Mh, Hh[0], Hh[NPW > 1 ? 1 : 0], Hh[NPW > 2 ? 2 : 0]);
            int4* w = reinterpret_cast<int4*>(my_row + (c & hm) * CW);
            w[0] = w0;
            if (NPW > 1) w[1] = w1;
            if (c ? (rc.y >> 15) & 1u : (uint32_t)save_col0) {   // a saved column: its cells stay available for the far reads
                int4* sw = reinterpret_cast<int4*>(saved + ((c ? (rc.y >> 12) & 7u : 0u) * n_loc + L) * CW);
                sw[0] = w0;
                if (NPW > 1) sw[1] = w1;
            }
            const uint32_t k_row = tid - 64 * GW;
            if (k_row >= out_first) {   // a hand-off row: the cell goes to the next strip as well (write-through)
                unsigned long long* dst = hout + ((size_t)(k_row - out_first) * (nC + 1) + c) * WW;
                __hip_atomic_store(dst, (unsigned long long)(unsigned)w0.x | ((unsigned long long)(unsigned)w0.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, (unsigned long long)(unsigned)w0.z | ((unsigned long long)(unsigned)w0.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (NPW > 1) {
                    __hip_atomic_store(dst + WW - 2, (unsigned long long)(unsigned)w1.x | ((unsigned long long)(unsigned)w1.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + WW - 1, (unsigned long long)(unsigned)w1.z | ((unsigned long long)(unsigned)w1.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!(B.skip_traceback & 2)) {
                const uint32_t pb = (off + ((swap ? c : r) - lo_d)) * 4u;
                plane_store(pM, pb, M);
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    plane_store(pV[k], pb, V[k]);
                    plane_store(pH[k], pb, Hh[k]);
                }
            }
            // the last row of the strip has finished column c: every hand-off row (all of them lanes of this wave, and ahead of this one) has too
            if (sd.n_out && k_row == S - 1 && (((c + 1) & 15u) == 0 || c == nC)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(SD.progress + sd.prog, c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            }
        }
        off += cnt_d;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    if (failed || (SD.debug_fail && (prob & 1u) && tid == 0)) *fail_flag = 1;
    __threadfence();     // the planes of this strip, for the workgroup that runs the traceback
    __syncthreads();
    const bool strip_failed = *fail_flag != 0;
    if (tid == 0) __hip_atomic_store(SD.progress + sd.prog, strip_failed ? kStripFailed : kStripDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sd.strip + 1 == sd.n_strips && tid < 64) {
        bool bad = strip_failed;
        for (uint32_t j = tid; j + 1 < sd.n_strips; j += 64) {
            const uint32_t* word = SD.progress + (sd.prog - sd.strip + j);
            unsigned spins = 0;
            uint32_t v;
            while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < kStripDone) {
                if (++spins > (1u << 20)) { v = kStripFailed; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            bad |= v == kStripFailed;
        }
        bad = __any(bad);
        __threadfence();
        if (bad) {
            if (tid == 0) { B.out_status[prob] = 9; B.out_len[prob] = 0; }
        } else if (!B.skip_traceback) traceback_wave<NPW>(B, pd, G, pl, P, prob);
    }
    cl_tick_end(B, true);
}

template <int NPW>
void launch_sys_npw(int block, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const uint32_t* plist, const ClScoreParams& P,
                    hipStream_t stream) {
    if (block <= 64) hipLaunchKernelGGL((popoa_sys_kernel<NPW, 64>), dim3(n_blocks), dim3(64), lds_bytes, stream, B, plist, P);
    else if (block <= 128) hipLaunchKernelGGL((popoa_sys_kernel<NPW, 128>), dim3(n_blocks), dim3(128), lds_bytes, stream, B, plist, P);
    else if (block <= 256) hipLaunchKernelGGL((popoa_sys_kernel<NPW, 256>), dim3(n_blocks), dim3(256), lds_bytes, stream, B, plist, P);
    else hipLaunchKernelGGL((popoa_sys_kernel<NPW, 1024>), dim3(n_blocks), dim3(1024), lds_bytes, stream, B, plist, P);
}

template <int NPW>
void launch_general_npw(int block, uint32_t n_blocks, uint32_t ring_bytes, const ClDeviceBatch& B, const uint32_t* plist,
                        const ClScoreParams& P, hipStream_t stream) {
    if (ring_bytes) {
        if (block <= 64)
            hipLaunchKernelGGL((popoa_ring_kernel<NPW, 64>), dim3(n_blocks), dim3(64), ring_bytes, stream, B, plist, P);
        else if (block <= 256)
            hipLaunchKernelGGL((popoa_ring_kernel<NPW, 256>), dim3(n_blocks), dim3(256), ring_bytes, stream, B, plist, P);
        else
            hipLaunchKernelGGL((popoa_ring_kernel<NPW, 1024>), dim3(n_blocks), dim3(1024), ring_bytes, stream, B, plist, P);
        return;
    }
    if (block <= 64)
        hipLaunchKernelGGL((popoa_general_kernel<NPW, 64>), dim3(n_blocks), dim3(64), 0, stream, B, plist, P);
    else if (block <= 256)
        hipLaunchKernelGGL((popoa_general_kernel<NPW, 256>), dim3(n_blocks), dim3(256), 0, stream, B, plist, P);
    else
        hipLaunchKernelGGL((popoa_general_kernel<NPW, 1024>), dim3(n_blocks), dim3(1024), 0, stream, B, plist, P);
}

#include "popoa_lane.h"

}  // namespace

// host-callable launcher (C++ linkage, used by cl_api.cpp only)
// ring_bytes > 0: the LDS-ring variant (every problem of the launch has its ring depth in ClProbDesc::pad and fits ring_bytes)
hipError_t cl_launch_popoa_lane(int W, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const uint32_t* plist, const ClScoreParams& P, uint32_t* lane_sync, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    if (lds_bytes > 150 * 1024) return hipErrorInvalidValue;   // (the planner keeps the hand-off window + the saved columns of a pair below that)
    static ClDeviceOnce attr_once;   // more than 64 KB of dynamic LDS needs the opt-in once per function and device
    attr_once([] {
        const int cap = 159 * 1024;   // (the kernel has a few bytes of static LDS as well: 160 KB of dynamic LDS is refused, and the refusal would surface as the NEXT launch's error)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_lane_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_lane_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_lane_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_lane_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipGetLastError();
    });
    if (lane_sync) {   // wide pairs: a workgroup per group of eight strips
        if (W != 4) return hipErrorInvalidValue;
        hipLaunchKernelGGL((popoa_lane_kernel<4, true>), dim3(n_blocks), dim3(256), lds_bytes, stream, B, plist, P, lane_sync);
        return hipGetLastError();
    }
    switch (W) {
    case 1: hipLaunchKernelGGL((popoa_lane_kernel<1, false>), dim3(n_blocks), dim3(64), lds_bytes, stream, B, plist, P, (uint32_t*)nullptr); break;
    case 3: hipLaunchKernelGGL((popoa_lane_kernel<3, false>), dim3(n_blocks), dim3(192), lds_bytes, stream, B, plist, P, (uint32_t*)nullptr); break;
    case 4: hipLaunchKernelGGL((popoa_lane_kernel<4, false>), dim3(n_blocks), dim3(256), lds_bytes, stream, B, plist, P, (uint32_t*)nullptr); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t cl_launch_popoa_sys(int npw, int block, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const uint32_t* plist,
                               const ClScoreParams& P, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    static ClDeviceOnce attr_once;   // more than 64 KB of dynamic LDS needs the opt-in once per function (worker threads launch concurrently)
    attr_once([] {
        const int cap = 160 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<1, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<1, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<2, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<2, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<2, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<3, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<3, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<3, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_sys_kernel<3, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    });
    switch (npw) {
    case 1: launch_sys_npw<1>(block, n_blocks, lds_bytes, B, plist, P, stream); break;
    case 2: launch_sys_npw<2>(block, n_blocks, lds_bytes, B, plist, P, stream); break;
    case 3: launch_sys_npw<3>(block, n_blocks, lds_bytes, B, plist, P, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// threads = 256 (the four ghost waves) + the largest strip's rows (a multiple of 64)
hipError_t cl_launch_popoa_strip(int npw, uint32_t threads, uint32_t n_blocks, uint32_t lds_bytes, const ClDeviceBatch& B, const ClStripDevice& SD, const uint32_t* slist,
                                 const ClScoreParams& P, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    if (threads < 320 || threads > 1024 || (threads & 63u)) return hipErrorInvalidValue;
    static ClDeviceOnce attr_once;
    attr_once([] {
        const int cap = 160 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_strip_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_strip_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_strip_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    });
    switch (npw) {
    case 1: hipLaunchKernelGGL((popoa_strip_kernel<1>), dim3(n_blocks), dim3(threads), lds_bytes, stream, B, SD, slist, P); break;
    case 2: hipLaunchKernelGGL((popoa_strip_kernel<2>), dim3(n_blocks), dim3(threads), lds_bytes, stream, B, SD, slist, P); break;
    case 3: hipLaunchKernelGGL((popoa_strip_kernel<3>), dim3(n_blocks), dim3(threads), lds_bytes, stream, B, SD, slist, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t cl_launch_popoa_general(int npw, int block, uint32_t n_blocks, uint32_t ring_bytes, const ClDeviceBatch& B,
                                   const uint32_t* plist, const ClScoreParams& P, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    static ClDeviceOnce attr_once;
    if (ring_bytes > 64 * 1024) attr_once([] {
        const int cap = 160 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<1, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<1, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<2, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<2, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<3, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<3, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&popoa_ring_kernel<3, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    });
    switch (npw) {
    case 1: launch_general_npw<1>(block, n_blocks, ring_bytes, B, plist, P, stream); break;
    case 2: launch_general_npw<2>(block, n_blocks, ring_bytes, B, plist, P, stream); break;
    case 3: launch_general_npw<3>(block, n_blocks, ring_bytes, B, plist, P, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
