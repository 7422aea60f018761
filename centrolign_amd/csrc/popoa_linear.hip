// popoa_linear.hip — register-resident wavefront kernel for chain x chain subproblems (every subproblem
// of a pairwise alignment; SURVEY.md §6: 100 % of the 2 x 1 Mbp batch).
//
// Same recurrences and tie-breaks as popoa_general_kernel / the reference's po_poa_internal
// (include/centrolign/alignment.hpp:753-1151), specialised to graphs whose nodes form one path
// (previous(r) == {r-1}, one source = rank 0, one sink = last rank):
//
//   * the shorter graph is laid across the lanes ("rows"; SWAP when that is graph 2): a wave owns a STRIP of
//     64*R consecutive rows, lane l holds rows l*R .. l*R+R-1;
//   * it sweeps the columns (the longer graph) as a systolic array: at step t lane l works on column
//     t - l + 1, so the cell above (lane l-1, one step earlier) and the diagonal cell (two steps earlier)
//     arrive through a single wave_shr:1 DPP move per value; graph-2 labels travel the same way;
//   * scores never leave registers.  What goes to HBM is one TRACEBACK CODE per cell (1 byte for
//     NumPW <= 2, 2 bytes for NumPW == 3) holding exactly the decisions the reference's traceback would
//     re-derive from its int32 planes with == tests (alignment.hpp:1048-1136):
//         bits 0-2  gap-close choice at this cell in the order I_0, D_0, I_1, D_1, I_2, D_2 (0 = none/diagonal)
//         bit 3+k   I_k(a,b) was reached by OPENING from Mf(a-1,b) (tested before extend, :1107-1117)
//         bit 3+NumPW+k  same for D_k
//     written as one coalesced run per wave and step;
//   * strips of one matrix are pipelined over the W waves of the workgroup: strip s+1 trails strip s by kLag
//     kChunk-step chunks and receives the last row of strip s (Mf, I_k per column) through a small HBM/L2
//     buffer, one coalesced load per chunk;
//   * the boundary row/column are closed forms for a chain (-(open_k) - len*extend_k, alignment.hpp:832-894),
//     so they are never stored; wave 0 walks the traceback over the codes, 64 cells per memory round trip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "popoa_device.h"

// The systolic moves below are the whole-wave DPP controls wave_shr:1 / wave_shl:1, which exist on the GFX9 family with 64-wide waves
// only; built for anything else the kernel would be rejected by the backend or, worse, compute wrong DP values without a diagnostic.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "popoa_linear.hip is written for gfx950 (MI355X): build with --offload-arch=gfx950"
#endif

namespace {

__device__ __forceinline__ int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

// lane l <- lane l-1 ; lane 0 <- fill (wave_shr:1, DPP control 0x138 on gfx9-family)
__device__ __forceinline__ int32_t shift_in(int32_t v, int32_t fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}

template <int NPW>
struct CodeT { using type = uint8_t; };
template <>
struct CodeT<3> { using type = uint16_t; };

template <int BYTES> struct PackT;
template <> struct PackT<1> { using type = uint8_t; };
template <> struct PackT<2> { using type = uint16_t; };
template <> struct PackT<4> { using type = uint32_t; };
template <> struct PackT<8> { using type = uint64_t; };

// Mf of a boundary cell at distance len >= 1 from the corner: max_k -(open_k + len*extend_k)
template <int NPW>
__device__ __forceinline__ int32_t boundary_m(const ClScoreParams& P, uint32_t len) {
    int32_t m = -P.oe[0] - (int32_t)(len - 1) * P.ext[0];
#pragma unroll
    for (int k = 1; k < NPW; ++k) m = imax(m, -P.oe[k] - (int32_t)(len - 1) * P.ext[k]);
    return m;
}

template <int NPW>
__device__ __forceinline__ int32_t boundary_gap(const ClScoreParams& P, int k, uint32_t len) {
    return -P.oe[k] - (int32_t)(len - 1) * P.ext[k];
}

// A strip's sweep is cut into chunks of kChunk steps; the workgroup barriers once per chunk.  Strip s+1 may read
// the columns of chunk c from strip s only after strip s has finished chunk c + kLag - 1 (its lane 63 reaches
// column C(c+1) at step C(c+1)+62), hence the start offset of kLag chunks between consecutive strips.
constexpr uint32_t kChunk = 32;
constexpr uint32_t kLag = 2 + 62 / kChunk;

// rows = the graph laid across the lanes (graph 1, or graph 2 when SWAP), cols = the graph that is swept
struct LinearGeom {
    uint32_t nr, nc, S, Cn, steps;
    template <int R>
    __device__ void init(uint32_t nr_, uint32_t nc_) {
        nr = nr_; nc = nc_;
        S = (nr + 64 * R - 1) / (64 * R);
        Cn = (nc + 63 + kChunk - 1) / kChunk;  // steps 0 .. nc+62
        steps = Cn * kChunk;
    }
};

// Wave-cooperative traceback over the codes; same walk as alignment.hpp:1036-1138 for a chain pair, but the
// 64 lanes of wave 0 read the codes of the next 64 cells ALONG THE CURRENT DIRECTION (diagonal while in the
// match state, along the gap while in an I_k / D_k state) with one load, find with a ballot where the run
// ends, and emit the whole run at once.  A traceback of length L with g gap events costs about L/64 + 3g
// dependent memory round trips instead of L.
// Works in kernel space (ra = row index, cb = column index, 1-based, 0 = boundary).  A "vertical" gap consumes
// a row node: it is the reference's I_k when rows are graph 1, its D_k when SWAP.
template <int NPW, int R, bool SWAP>
__device__ void linear_traceback(const ClDeviceBatch& B, const ClProbDesc& pd, const LinearGeom& G,
                                 const typename CodeT<NPW>::type* codes, const ClScoreParams& P, uint32_t prob,
                                 uint32_t lane) {
    uint32_t ra = G.nr, cb = G.nc, len = 0, status = 0;
    const uint32_t cap = pd.n1 + pd.n2;
    uint2* out = B.out_pairs + pd.out_base;
    int comp = 0;  // reference convention: > 0 in I_{comp-1}, < 0 in D_{-comp-1}
    auto code_at = [&](uint32_t r1, uint32_t c1) -> uint32_t {
        const uint32_t row = r1 - 1, s = row / (64 * R), rr = row - s * 64 * R, l = rr / R, r = rr - l * R;
        const uint32_t t = (c1 - 1) + l;
        return codes[(((size_t)s * G.steps + t) * 64 + l) * R + r];
    };
    auto pair_of = [&](uint32_t r1, uint32_t c1) { return SWAP ? make_uint2(c1, r1) : make_uint2(r1, c1); };
    while (true) {
        if (len > cap) { status = 2; break; }
        if (ra && cb) {
            if (comp == 0) {
                const bool valid = lane < ra && lane < cb;
                const uint32_t code = valid ? code_at(ra - lane, cb - lane) : 0u;
                const uint32_t cc = code & 7u;
                const unsigned long long stop = __ballot(!valid || cc != 0u);
                const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                if (lane < first) out[cap - 1 - (len + lane)] = pair_of(ra - lane, cb - lane);
                len += first;
                ra -= first; cb -= first;
                if (first < 64u && ra && cb) {
                    // a gap closes at this cell (first hit in the order I_0, D_0, I_1, ... ; :1048-1066)
                    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)cc, (int)first);
                    comp = (c1 & 1u) ? (int)((c1 + 1) >> 1) : -(int)(c1 >> 1);
                } else if (ra == 0 && cb == 0) {
                    break;  // emitted the (1,1) cell: the stored corner is -inf, no predecessor matches (:1091-1099)
                }
            } else {
                const bool vert = SWAP ? comp < 0 : comp > 0;
                const int k = comp > 0 ? comp - 1 : -comp - 1;
                const uint32_t x = vert ? ra : cb;
                const bool valid = lane < x;
                const uint32_t code = valid ? (vert ? code_at(ra - lane, cb) : code_at(ra, cb - lane)) : 0u;
                // the run ends at the first cell that was reached by OPENING (tested before extend, :1105-1136)
                const bool open = valid && ((code >> (vert ? 3 + k : 3 + NPW + k)) & 1u);
                const unsigned long long mo = __ballot(open), mi = __ballot(!valid);
                const uint32_t fo = mo ? (uint32_t)__builtin_ctzll(mo) : 64u, fi = mi ? (uint32_t)__builtin_ctzll(mi) : 64u;
                const uint32_t n = fo < fi ? fo + 1 : fi;
                if (lane < n) out[cap - 1 - (len + lane)] = vert ? pair_of(ra - lane, 0u) : pair_of(0u, cb - lane);
                len += n;
                if (vert) ra -= n; else cb -= n;
                if (fo < fi) comp = 0;
            }
        } else if (ra == 0 && cb == 0) {
            status = 3; break;
        } else {
            // boundary row (ra == 0) or column (cb == 0): closed forms, M = max_k G_k, G_k = -(open_k) - x*extend_k,
            // the other gap family is -inf.  Walk exactly as :1048-1066 / :1101-1137 do on those cells.
            const bool vert = cb == 0;             // boundary column: gaps consume row nodes
            const bool is_i = vert != SWAP;        // ... which are I_k unless the graphs are swapped
            uint32_t x = vert ? ra : cb;
            if (comp == 0) {
                const int32_t Mv = boundary_m<NPW>(P, x);
#pragma unroll
                for (int k = NPW - 1; k >= 0; --k) if (Mv == boundary_gap<NPW>(P, k, x)) comp = is_i ? k + 1 : -k - 1;
            }
            if (is_i ? comp <= 0 : comp >= 0) { status = 3; break; }
            const int k = is_i ? comp - 1 : -comp - 1;
            // the run ends after the first cell whose gap value also equals an OPEN from the previous boundary cell,
            // or at x == 1 (its predecessor is the corner: nothing matches)
            const bool valid = lane < x;
            const uint32_t xj = x - lane;
            const bool open = valid && xj >= 2 && boundary_gap<NPW>(P, k, xj) == boundary_m<NPW>(P, xj - 1) - P.oe[k];
            const unsigned long long mo = __ballot(open), mi = __ballot(!valid);
            const uint32_t fo = mo ? (uint32_t)__builtin_ctzll(mo) : 64u, fi = mi ? (uint32_t)__builtin_ctzll(mi) : 64u;
            const uint32_t n = fo < fi ? fo + 1 : fi;
            if (lane < n) out[cap - 1 - (len + lane)] = vert ? pair_of(xj, 0u) : pair_of(0u, xj);
            len += n; x -= n;
            if (vert) ra = x; else cb = x;
            if (fo < fi) comp = 0;
            else if (x == 0) break;  // emitted the cell next to the corner
        }
    }
    if (lane == 0) {
        B.out_len[prob] = len > cap ? cap : len;
        B.out_status[prob] = status;
    }
}

// lane l <- lane l+1 (wave_shl:1); lane 63 keeps its value
__device__ __forceinline__ int32_t rotate_down(int32_t v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
}

template <int NPW, int R, int W, bool SWAP>
__device__ __forceinline__ void linear_body(const ClDeviceBatch& B, const ClProbDesc& pd, uint32_t prob,
                                            const ClScoreParams& P) {
    using code_t = typename CodeT<NPW>::type;
    using pack_t = typename PackT<R * sizeof(code_t)>::type;
    constexpr uint32_t C = kChunk;
    LinearGeom G;
    G.init<R>(SWAP ? pd.n2 : pd.n1, SWAP ? pd.n1 : pd.n2);
    const uint32_t nr = G.nr, nc = G.nc, steps = G.steps;
    const uint8_t* labR = B.lab[SWAP ? 1 : 0] + pd.node_base[SWAP ? 1 : 0];
    const uint8_t* labC = B.lab[SWAP ? 0 : 1] + pd.node_base[SWAP ? 0 : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    code_t* codes = reinterpret_cast<code_t*>(B.planes + pd.plane_base);
    // hand-off rows between strips: [strip][1 + NPW][steps] int32, after the codes (16-byte aligned by the packer)
    const size_t code_bytes = ((size_t)G.S * steps * 64 * R * sizeof(code_t) + 15) & ~(size_t)15;
    int32_t* brow = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(codes) + code_bytes);

    const uint32_t Pm = G.Cn > kLag * W ? G.Cn : kLag * W;  // macro-step period of one round of W strips
    const uint32_t total = ((G.S - 1) / W) * Pm + kLag * ((G.S - 1) % W) + G.Cn;

    // per-strip register state.  V_k = gap family that consumes row nodes (vertical), H_k = column nodes.
    int32_t Mleft[R], Hleft[R][NPW], labr[R];
    int32_t lastM = 0, lastV[NPW], prevUpM = 0, c2 = 0xff;
    int32_t bM = 0, bV[NPW], myc2 = 0xff;
#pragma unroll
    for (int k = 0; k < NPW; ++k) { lastV[k] = CL_NEG_INF; bV[k] = CL_NEG_INF; }

    for (uint32_t m = 0; m < total; ++m) {
        const int32_t mm = (int32_t)m - (int32_t)(kLag * wave);
        if (mm >= 0) {
            const uint32_t j = (uint32_t)mm / Pm, c = (uint32_t)mm - j * Pm, s = j * W + wave;
            if (s < G.S && c < G.Cn) {
                const uint32_t row0 = s * 64 * R + lane * R;  // 0-based first row of this lane; row index = row0 + r + 1
                if (c == 0) {
                    // boundary column: Mf closed form, H_k = -inf (alignment.hpp:832-862 / :864-894)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t a = row0 + r + 1;
                        Mleft[r] = boundary_m<NPW>(P, a);
#pragma unroll
                        for (int k = 0; k < NPW; ++k) Hleft[r][k] = CL_NEG_INF;
                        labr[r] = a <= nr ? (int32_t)(labR[a - 1] & 0x7f) : 0xfe;
                    }
                    lastM = Mleft[R - 1];
                    // diagonal term of the lane's first row at column 1: Mf(row-1, 0); the corner counts as 0 (:814-818)
                    prevUpM = row0 == 0 ? 0 : boundary_m<NPW>(P, row0);
                    c2 = 0xff;
                }
                const uint32_t t0 = c * C;
                {   // this chunk's C columns as seen by lane 0: labels and the row above the strip
                    const uint32_t colb = t0 + lane + 1;
                    const bool v = lane < C && colb <= nc;
                    myc2 = v ? (int32_t)(labC[colb - 1] & 0x7f) : 0xff;
                    if (s == 0) {
                        bM = v ? boundary_m<NPW>(P, colb) : CL_NEG_INF;  // boundary row: Mf, V_k = -inf
                    } else {
                        const int32_t* src = brow + (size_t)(s - 1) * (1 + NPW) * steps + (colb - 1);
                        bM = v ? src[0] : CL_NEG_INF;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) bV[k] = v ? src[(size_t)(1 + k) * steps] : CL_NEG_INF;
                    }
                }
                pack_t* cw = reinterpret_cast<pack_t*>(codes) + ((size_t)s * steps + t0) * 64 + lane;
                int32_t* bout = brow + (size_t)s * (1 + NPW) * steps;
                const bool hand_off_lane = s + 1 < G.S && lane == 63;
#pragma unroll 2
                for (uint32_t jj = 0; jj < C; ++jj) {
                    const uint32_t t = t0 + jj;
                    // lane 0 takes the next column of the row above the strip (kept rotating in bM/bV/myc2),
                    // every other lane takes what its upper neighbour produced one step ago
                    const int32_t upM = shift_in(lastM, bM);
                    bM = rotate_down(bM);
                    int32_t upV[NPW];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) { upV[k] = shift_in(lastV[k], bV[k]); bV[k] = rotate_down(bV[k]); }
                    c2 = shift_in(c2, myc2);
                    myc2 = rotate_down(myc2);
                    const uint32_t b = t - lane + 1;  // this lane's column (1-based); wraps when not started
                    if ((uint32_t)(b - 1) < nc && row0 < nr) {  // lanes below the last row have nothing to do
                        int32_t diag = prevUpM, uM = upM, uV[NPW];
#pragma unroll
                        for (int k = 0; k < NPW; ++k) uV[k] = upV[k];
                        uint64_t pack = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const int32_t sc = labr[r] == c2 ? P.match : -P.mismatch;
                            int32_t Mf = diag + sc;
                            int32_t V[NPW], H[NPW];
                            uint32_t code = 0;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) {
                                const int32_t vo = uM - P.oe[k], ho = Mleft[r] - P.oe[k];
                                V[k] = imax(vo, uV[k] - P.ext[k]);
                                H[k] = imax(ho, Hleft[r][k] - P.ext[k]);
                                code |= (V[k] == vo ? 1u : 0u) << (3 + k);
                                code |= (H[k] == ho ? 1u : 0u) << (3 + NPW + k);
                                Mf = imax(Mf, imax(V[k], H[k]));
                            }
                            uint32_t cc = 0;
#pragma unroll
                            for (int k = NPW - 1; k >= 0; --k) {  // lowest k, I before D, wins (:1048-1066)
                                if (SWAP) {
                                    cc = Mf == V[k] ? 2u * k + 2u : cc;  // rows are graph 2: vertical gaps are D_k
                                    cc = Mf == H[k] ? 2u * k + 1u : cc;
                                } else {
                                    cc = Mf == H[k] ? 2u * k + 2u : cc;
                                    cc = Mf == V[k] ? 2u * k + 1u : cc;
                                }
                            }
                            code |= cc;
                            pack |= (uint64_t)code << (8 * sizeof(code_t) * r);
                            diag = Mleft[r];
                            Mleft[r] = Mf;
                            uM = Mf;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) { Hleft[r][k] = H[k]; uV[k] = V[k]; }
                        }
                        lastM = uM;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) lastV[k] = uV[k];
                        cw[(size_t)jj * 64] = (pack_t)pack;
                        if (hand_off_lane) {  // last row of a full strip: Mf and V_k of this column feed the next strip
                            bout[b - 1] = uM;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) bout[(size_t)(1 + k) * steps + (b - 1)] = uV[k];
                        }
                    }
                    prevUpM = upM;
                }
                if (s + 1 == G.S && c + 1 == G.Cn) {
                    // Mf(n1, n2): the lane holding the last row has it in Mleft after the last column
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (row0 + r + 1 == nr) B.out_score[prob] = Mleft[r];
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) linear_traceback<NPW, R, SWAP>(B, pd, G, codes, P, prob, lane);
}

// ---- large chain pairs over SEVERAL workgroups (round 6) -------------------------------------------------------------------------------------------------------
// A chain pair of thousands of rows on ONE workgroup sweeps its strips in rounds of W (2 048 x 2 048 on sixteen waves: two rounds, 0.43 us per step at sixteen waves a
// barrier; 6 300 x 6 300 was handed to the DAG strip kernel at 1.0 us per step: 13.3 ms).  Here the strips of a pair are dealt over GROUPS of kSpanW = 4 strips, one
// workgroup per group, every group in ONE round, all groups of the pair in flight on different compute units: inside a group the strips follow one another through the
// area behind the codes exactly as in linear_body (kLag chunks apart, one barrier per chunk); the FIRST strip of group g follows the LAST strip of group g - 1 the same way
// across compute units.  That hand-off row (Mf, V_k per column) is written through to memory (agent-scope relaxed atomic stores) and read past the caches (agent-scope
// relaxed atomic loads), and a PROGRESS word per group — "chunks my last strip has finished", stored after a vmcnt(0) of the one wave that holds the group's hand-off row —
// is what the next group's wave 0 polls before a chunk: no fence while the sweep runs (the protocol of popoa_strip_kernel and of the WIDE popoa_lane_kernel).  Codes and
// hand-off layout are linear_body's, so linear_traceback walks them unchanged and a pair whose groups gave up waiting for one another (status 9; every wait is bounded)
// is simply run again by popoa_linear_kernel<16> on the same workspace (cl_stitch_plan_collect).  The wavefront's slope stays one row per step: (columns + 1.5 x rows)
// steps instead of rounds x (columns + 1 024): 6 300 x 6 300 in 15 800 steps of ~0.25 us.
constexpr uint32_t kSpanW = 4;                 // strips (waves) per group (measured: 2 048^2 x 10 / 6 300^2 in 1.65 / 5.03 ms with two, 1.48 / 4.54 with four, 1.79 / 5.66 with eight)
constexpr uint32_t kSpanFailed = 0xFFFFFFFFu;
constexpr uint32_t kSpanPolls = 1u << 20;      // x ~1 us
constexpr uint32_t kSpanPublish = 2;           // chunks between two progress stores of a group (each costs its last wave a vmcnt(0))

template <int NPW, bool SWAP>
__device__ __forceinline__ void linear_span_body(const ClDeviceBatch& B, const ClProbDesc& pd, uint32_t prob, const ClScoreParams& P, uint32_t grp, uint32_t* sync) {
    using code_t = typename CodeT<NPW>::type;
    constexpr uint32_t C = kChunk, W = kSpanW;
    LinearGeom G;
    G.init<1>(SWAP ? pd.n2 : pd.n1, SWAP ? pd.n1 : pd.n2);
    const uint32_t nr = G.nr, nc = G.nc, steps = G.steps;
    const uint8_t* labR = B.lab[SWAP ? 1 : 0] + pd.node_base[SWAP ? 1 : 0];
    const uint8_t* labC = B.lab[SWAP ? 0 : 1] + pd.node_base[SWAP ? 0 : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    code_t* codes = reinterpret_cast<code_t*>(B.planes + pd.plane_base);
    const size_t code_bytes = ((size_t)G.S * steps * 64 * sizeof(code_t) + 15) & ~(size_t)15;
    int32_t* brow = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(codes) + code_bytes);
    const uint32_t n_groups = pd.aux_cnt;
    uint32_t* const progress = sync + pd.aux_base;
    uint32_t* const done = progress + n_groups;
    const uint32_t s = grp * W + wave;                                          // this wave's strip (none if s >= S)
    const uint32_t strips_here = G.S - grp * W < W ? G.S - grp * W : W;
    const uint32_t total = kLag * (strips_here - 1) + G.Cn;
    __shared__ uint32_t gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    __syncthreads();
    bool dead = false;
    uint32_t seen = 0;

    int32_t Mleft, Hleft[NPW], labr = 0xfe;
    int32_t lastM = 0, lastV[NPW], prevUpM = 0, c2 = 0xff;
    int32_t bM = 0, bV[NPW], myc2 = 0xff;
    Mleft = 0;
#pragma unroll
    for (int k = 0; k < NPW; ++k) { lastV[k] = CL_NEG_INF; bV[k] = CL_NEG_INF; Hleft[k] = CL_NEG_INF; }

    for (uint32_t m = 0; m < total; ++m) {
        const int32_t mm = (int32_t)m - (int32_t)(kLag * wave);
        if (mm >= 0 && s < G.S && (uint32_t)mm < G.Cn && !dead) {
            const uint32_t c = (uint32_t)mm;
            const uint32_t row0 = s * 64 + lane;   // 0-based row of this lane; row index = row0 + 1
            if (c == 0) {
                const uint32_t a = row0 + 1;
                Mleft = boundary_m<NPW>(P, a);
#pragma unroll
                for (int k = 0; k < NPW; ++k) Hleft[k] = CL_NEG_INF;
                labr = a <= nr ? (int32_t)(labR[a - 1] & 0x7f) : 0xfe;
                lastM = Mleft;
                prevUpM = row0 == 0 ? 0 : boundary_m<NPW>(P, row0);
                c2 = 0xff;
            }
            const uint32_t t0 = c * C;
            bool starved = false;
            const bool across = wave == 0 && grp > 0;   // the row above this strip belongs to another workgroup
            if (across) {
                // chunk c needs the upper group's last strip to have finished chunk c + kLag - 1 (see kLag); the first chunk waits for two more, so that the word read
                // is usually ahead of the need afterwards and one read serves several chunks
                const uint32_t want = c == 0 ? c + kLag + kSpanPublish : c + kLag;
                const uint32_t need = want < G.Cn ? want : G.Cn;
                uint32_t polls = 0;
                while (seen < need && seen != kSpanFailed) {
                    seen = __hip_atomic_load(progress + (grp - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen >= need || ++polls > kSpanPolls) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                starved = seen == kSpanFailed || seen < need;
                if (starved && lane == 0) gave_up = 1;
            }
            if (!starved) {
                {   // this chunk's C columns as seen by lane 0: labels and the row above the strip
                    const uint32_t colb = t0 + lane + 1;
                    const bool v = lane < C && colb <= nc;
                    myc2 = v ? (int32_t)(labC[colb - 1] & 0x7f) : 0xff;
                    if (s == 0) {
                        bM = v ? boundary_m<NPW>(P, colb) : CL_NEG_INF;
                    } else {
                        const int32_t* src = brow + (size_t)(s - 1) * (1 + NPW) * steps + (colb - 1);
                        if (across) {
                            bM = v ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) bV[k] = v ? __hip_atomic_load(src + (size_t)(1 + k) * steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : CL_NEG_INF;
                        } else {
                            bM = v ? src[0] : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) bV[k] = v ? src[(size_t)(1 + k) * steps] : CL_NEG_INF;
                        }
                    }
                }
                code_t* cw = codes + ((size_t)s * steps + t0) * 64 + lane;
                int32_t* bout = brow + (size_t)s * (1 + NPW) * steps;
                const bool hand_off_lane = s + 1 < G.S && lane == 63;
                const bool hand_across = wave + 1 == W;   // (the next strip, if any, is another workgroup's first)
#pragma unroll 2
                for (uint32_t jj = 0; jj < C; ++jj) {
                    const uint32_t t = t0 + jj;
                    const int32_t upM = shift_in(lastM, bM);
                    bM = rotate_down(bM);
                    int32_t upV[NPW];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) { upV[k] = shift_in(lastV[k], bV[k]); bV[k] = rotate_down(bV[k]); }
                    c2 = shift_in(c2, myc2);
                    myc2 = rotate_down(myc2);
                    const uint32_t b = t - lane + 1;
                    if ((uint32_t)(b - 1) < nc && row0 < nr) {
                        const int32_t sc = labr == c2 ? P.match : -P.mismatch;
                        int32_t Mf = prevUpM + sc;
                        int32_t V[NPW], H[NPW];
                        uint32_t code = 0;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) {
                            const int32_t vo = upM - P.oe[k], ho = Mleft - P.oe[k];
                            V[k] = imax(vo, upV[k] - P.ext[k]);
                            H[k] = imax(ho, Hleft[k] - P.ext[k]);
                            code |= (V[k] == vo ? 1u : 0u) << (3 + k);
                            code |= (H[k] == ho ? 1u : 0u) << (3 + NPW + k);
                            Mf = imax(Mf, imax(V[k], H[k]));
                        }
                        uint32_t cc = 0;
#pragma unroll
                        for (int k = NPW - 1; k >= 0; --k) {
                            if (SWAP) { cc = Mf == V[k] ? 2u * k + 2u : cc; cc = Mf == H[k] ? 2u * k + 1u : cc; }
                            else { cc = Mf == H[k] ? 2u * k + 2u : cc; cc = Mf == V[k] ? 2u * k + 1u : cc; }
                        }
                        code |= cc;
                        Mleft = Mf;
                        lastM = Mf;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) { Hleft[k] = H[k]; lastV[k] = V[k]; }
                        cw[(size_t)jj * 64] = (code_t)code;
                        if (hand_off_lane) {
                            if (hand_across) {
                                __hip_atomic_store(bout + (b - 1), Mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                                for (int k = 0; k < NPW; ++k) __hip_atomic_store(bout + (size_t)(1 + k) * steps + (b - 1), V[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                bout[b - 1] = Mf;
#pragma unroll
                                for (int k = 0; k < NPW; ++k) bout[(size_t)(1 + k) * steps + (b - 1)] = V[k];
                            }
                        }
                    }
                    prevUpM = upM;
                }
                if (s + 1 == G.S && c + 1 == G.Cn && row0 + 1 == nr) B.out_score[prob] = Mleft;
            }
        }
        if (grp + 1 < n_groups && wave + 1 == strips_here && mm >= 0 && (uint32_t)mm < G.Cn && !dead && (((uint32_t)mm + 1) % kSpanPublish == 0 || (uint32_t)mm + 1 == G.Cn)) {
            // the group's last strip has finished chunk mm: its hand-off row becomes visible to the other compute units, then the count
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(progress + grp, (uint32_t)mm + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (grp > 0 && !dead && gave_up) {
            dead = true;
            if (threadIdx.x == 0) __hip_atomic_store(progress + grp, kSpanFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // every wave's codes out to the other compute units, then this group's mark; the last group collects the marks and walks back
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done + grp, dead ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (grp + 1 != n_groups || threadIdx.x >= 64) return;
    bool ok = !dead;
    for (uint32_t g2 = 0; g2 + 1 < n_groups && ok; ++g2) {
        uint32_t v = 0, polls = 0;
        while ((v = __hip_atomic_load(done + g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++polls <= kSpanPolls) __builtin_amdgcn_s_sleep(8);
        ok = v == 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok || B.debug_span_fail) {
        if (lane == 0) { B.out_len[prob] = 0; B.out_status[prob] = 9; }
        return;
    }
    linear_traceback<NPW, 1, SWAP>(B, pd, G, codes, P, prob, lane);
}

// a pair's groups are consecutive workgroups of the launch (plist repeats the pair once per group); sync: the plan's progress / done words
__global__ void __launch_bounds__(64 * kSpanW) popoa_linear_span_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist, ClScoreParams P, uint32_t* sync) {
    cl_tick_start(B, true);
    const uint32_t prob = plist[blockIdx.x];
    const ClProbDesc pd = B.desc[prob];
    uint32_t grp = 0;
    while (grp < blockIdx.x && plist[blockIdx.x - grp - 1] == prob) ++grp;
    const bool swap = pd.pad & 1u;
    switch (pd.npw) {
    case 1: if (swap) linear_span_body<1, true>(B, pd, prob, P, grp, sync); else linear_span_body<1, false>(B, pd, prob, P, grp, sync); break;
    case 2: if (swap) linear_span_body<2, true>(B, pd, prob, P, grp, sync); else linear_span_body<2, false>(B, pd, prob, P, grp, sync); break;
    default: if (swap) linear_span_body<3, true>(B, pd, prob, P, grp, sync); else linear_span_body<3, false>(B, pd, prob, P, grp, sync); break;
    }
    cl_tick_end(B, true);
}

// One kernel per workgroup shape (W waves).  The (NumPW, rows per lane, orientation) variant of each subproblem
// is picked at run time from its descriptor, so a whole stitch pass is three launches that start together
// instead of a dozen that queue behind each other on the hardware queues.
template <int NPW, int W>
__device__ __forceinline__ void linear_dispatch(const ClDeviceBatch& B, const ClProbDesc& pd, uint32_t prob,
                                                const ClScoreParams& P) {
    const bool swap = pd.pad & 1u;
    const uint32_t rows = (pd.pad >> 1) & 7u;
    if (W == 1 && rows == 2) {
        if (swap) linear_body<NPW, (W == 1 ? 2 : 1), W, true>(B, pd, prob, P);
        else linear_body<NPW, (W == 1 ? 2 : 1), W, false>(B, pd, prob, P);
    } else {
        if (swap) linear_body<NPW, 1, W, true>(B, pd, prob, P);
        else linear_body<NPW, 1, W, false>(B, pd, prob, P);
    }
}

template <int W>
__global__ void __launch_bounds__(64 * W) popoa_linear_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist,
                                                              ClScoreParams P) {
    cl_tick_start(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
    const uint32_t prob = plist[blockIdx.x];
    const ClProbDesc pd = B.desc[prob];
    switch (pd.npw) {
    case 1: linear_dispatch<1, W>(B, pd, prob, P); break;
    case 2: linear_dispatch<2, W>(B, pd, prob, P); break;
    default: linear_dispatch<3, W>(B, pd, prob, P); break;
    }
    cl_tick_end(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
}

// ---- four small chain pairs per wave -------------------------------------------------------------------------------------------------------------------
// A stitch pass of a progressive MSA is bound by VALU issue over the whole device, and half of its subproblems are chain pairs whose shorter side has at most
// 16 nodes (10 x 1 Mbp: 57 000 of 107 000 pairs, 2 % of the cells, a quarter of all wave-steps): one such pair per wave leaves 48 of 64 lanes idle for n + 63
// steps.  Here a wave takes FOUR of them: lanes 16 g .. 16 g + 15 are pair g's rows, the systolic moves are the ROW forms of the same DPP controls
// (row_shr:1 / row_shl:1 act inside rows of 16 lanes: lane 0 of a row keeps the fill), the column feed is a chunk of 16 columns per segment.  Same cell, same
// codes at the same addresses as linear_body<NPW, 1, 1, SWAP> (a pair's code buffer is indexed by ITS step and ITS lane), so linear_traceback walks them
// unchanged, one pair after the other.  The quad's pairs have the same NumPW (host: cl_api.cpp sorts them by NumPW and length); 0xFFFFFFFF = no pair.
// SEG = lanes of a segment = rows a pair may have: 16 (four pairs per wave, the ROW forms of the DPP moves) or 32 (two pairs per wave — round 5, second half: the
// 17 000 pairs of 17-32 rows of a 10 x 1 Mbp step cost a wave 110 steps each alone, 80 steps per two here; the wave forms of the moves, and the one lane that would
// take its neighbour segment's value — lane 32 — takes the fill by a select).  rotate_down needs no such care: what enters a segment's last lane from the next
// segment reaches lane 0 only after SEG more rotations, and a chunk of SEG columns is reloaded by then
template <int SEG>
__device__ __forceinline__ int32_t seg_shift_in(int32_t v, int32_t fill, uint32_t l) {
    if (SEG == 16) return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    const int32_t r = __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);          // wave_shr:1
    return l == 0 ? fill : r;
}
template <int SEG>
__device__ __forceinline__ int32_t seg_rotate_down(int32_t v) {
    return SEG == 16 ? __builtin_amdgcn_update_dpp(v, v, 0x101, 0xf, 0xf, false)             // row_shl:1
                     : __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);            // wave_shl:1
}

template <int NPW, int SEG>
__device__ __forceinline__ void linear_quad(const ClDeviceBatch& B, const uint32_t* __restrict__ quad, const ClScoreParams& P) {
    using code_t = typename CodeT<NPW>::type;
    constexpr uint32_t CQ = SEG;   // columns per chunk = lanes of a segment
    constexpr int NSEG = 64 / SEG;
    const uint32_t lane = threadIdx.x & 63u, seg = lane / SEG, l = lane % SEG;
    const uint32_t prob = quad[seg];
    const bool have = prob != 0xFFFFFFFFu;
    ClProbDesc pd = B.desc[have ? prob : quad[0]];
    const bool swp = pd.pad & 1u;
    const uint32_t nr = have ? (swp ? pd.n2 : pd.n1) : 0u, nc = have ? (swp ? pd.n1 : pd.n2) : 0u;
    const uint8_t* labR = B.lab[swp ? 1 : 0] + pd.node_base[swp ? 1 : 0];
    const uint8_t* labC = B.lab[swp ? 0 : 1] + pd.node_base[swp ? 0 : 1];
    code_t* codes = reinterpret_cast<code_t*>(B.planes + pd.plane_base);
    // the longest pair of the wave sets the number of steps (its last row reaches its last column at step nc + nr - 2)
    uint32_t last = 0;
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)nc, g * SEG), r = (uint32_t)__builtin_amdgcn_readlane((int)nr, g * SEG);
        if (c && r && c + r > last) last = c + r;
    }
    const uint32_t a = l + 1;   // this lane's row (1-based)
    int32_t Mleft = boundary_m<NPW>(P, a), Hleft[NPW], lastM, lastV[NPW], prevUpM = l == 0 ? 0 : boundary_m<NPW>(P, l), c2 = 0xff;
    const int32_t labr = a <= nr ? (int32_t)(labR[a - 1] & 0x7f) : 0xfe;
    int32_t bM = CL_NEG_INF, myc2 = 0xff;
    lastM = Mleft;
#pragma unroll
    for (int k = 0; k < NPW; ++k) { Hleft[k] = CL_NEG_INF; lastV[k] = CL_NEG_INF; }
    for (uint32_t t0 = 0; t0 + 1 < last; t0 += CQ) {
        {   // this chunk's columns as seen by the segment's lane 0: label and the boundary row's Mf (V_k of the boundary row = -inf)
            const uint32_t colb = t0 + l + 1;
            const bool v = colb <= nc;
            myc2 = v ? (int32_t)(labC[colb - 1] & 0x7f) : 0xff;
            bM = v ? boundary_m<NPW>(P, colb) : CL_NEG_INF;
        }
        for (uint32_t jj = 0; jj < CQ; ++jj) {
            const uint32_t t = t0 + jj;
            const int32_t upM = seg_shift_in<SEG>(lastM, bM, l);
            bM = seg_rotate_down<SEG>(bM);
            int32_t upV[NPW];
#pragma unroll
            for (int k = 0; k < NPW; ++k) upV[k] = seg_shift_in<SEG>(lastV[k], CL_NEG_INF, l);
            c2 = seg_shift_in<SEG>(c2, myc2, l);
            myc2 = seg_rotate_down<SEG>(myc2);
            const uint32_t b = t - l + 1;   // this lane's column (1-based); wraps when not started
            if ((uint32_t)(b - 1) < nc && l < nr) {
                const int32_t sc = labr == c2 ? P.match : -P.mismatch;
                int32_t Mf = prevUpM + sc;
                int32_t V[NPW], H[NPW];
                uint32_t code = 0;
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    const int32_t vo = upM - P.oe[k], ho = Mleft - P.oe[k];
                    V[k] = imax(vo, upV[k] - P.ext[k]);
                    H[k] = imax(ho, Hleft[k] - P.ext[k]);
                    code |= (V[k] == vo ? 1u : 0u) << (3 + k);
                    code |= (H[k] == ho ? 1u : 0u) << (3 + NPW + k);
                    Mf = imax(Mf, imax(V[k], H[k]));
                }
                uint32_t cc = 0;
#pragma unroll
                for (int k = NPW - 1; k >= 0; --k) {  // lowest k, I before D, wins (alignment.hpp:1048-1066); rows are graph 2 when swapped: vertical gaps are D_k then
                    const uint32_t vcode = swp ? 2u * k + 2u : 2u * k + 1u, hcode = swp ? 2u * k + 1u : 2u * k + 2u;
                    if (swp) { cc = Mf == V[k] ? vcode : cc; cc = Mf == H[k] ? hcode : cc; }
                    else { cc = Mf == H[k] ? hcode : cc; cc = Mf == V[k] ? vcode : cc; }
                }
                code |= cc;
                codes[((size_t)t * 64 + l)] = (code_t)code;
                Mleft = Mf;
                lastM = Mf;
#pragma unroll
                for (int k = 0; k < NPW; ++k) { Hleft[k] = H[k]; lastV[k] = V[k]; }
            }
            prevUpM = upM;
        }
    }
    if (have && a == nr) B.out_score[prob] = Mleft;
    __syncthreads();   // the codes are in memory
    for (int g = 0; g < NSEG; ++g) {
        const uint32_t pg = quad[g];
        if (pg == 0xFFFFFFFFu) continue;
        const ClProbDesc pdg = B.desc[pg];
        const bool sw = pdg.pad & 1u;
        LinearGeom G;
        G.init<1>(sw ? pdg.n2 : pdg.n1, sw ? pdg.n1 : pdg.n2);
        const code_t* cg = reinterpret_cast<const code_t*>(B.planes + pdg.plane_base);
        if (sw) linear_traceback<NPW, 1, true>(B, pdg, G, cg, P, pg, lane);
        else linear_traceback<NPW, 1, false>(B, pdg, G, cg, P, pg, lane);
    }
}

template <int SEG>
__device__ __forceinline__ void linear_pack_entry(const ClDeviceBatch& B, const uint32_t* __restrict__ plist, const ClScoreParams& P) {
    cl_tick_start(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
    const uint32_t* quad = plist + (64u / SEG) * blockIdx.x;
    switch (B.desc[quad[0]].npw) {
    case 1: linear_quad<1, SEG>(B, quad, P); break;
    case 2: linear_quad<2, SEG>(B, quad, P); break;
    default: linear_quad<3, SEG>(B, quad, P); break;
    }
    cl_tick_end(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
}
__global__ void __launch_bounds__(64) popoa_linear_quad_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist, ClScoreParams P) { linear_pack_entry<16>(B, plist, P); }
__global__ void __launch_bounds__(64) popoa_linear_duo_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist, ClScoreParams P) { linear_pack_entry<32>(B, plist, P); }

}  // namespace

// bytes of workspace a chain problem needs (codes + strip hand-off rows); mirrors the kernel's layout.
// nr / nc = nodes of the graph laid across lanes / swept
size_t cl_linear_workspace_bytes(uint32_t nr, uint32_t nc, int npw, int R) {
    const size_t S = (nr + 64 * (size_t)R - 1) / (64 * (size_t)R);
    const size_t steps = ((nc + 63 + kChunk - 1) / kChunk) * (size_t)kChunk;
    const size_t csize = npw == 3 ? 2 : 1;
    const size_t code_bytes = (S * steps * 64 * R * csize + 15) & ~(size_t)15;
    const size_t brow_bytes = S * (size_t)(1 + npw) * steps * sizeof(int32_t);
    return code_bytes + brow_bytes;
}

// the chain pairs that span several workgroups (popoa_linear_span_kernel): n_blocks = groups over all pairs of the launch, sync = the plan's progress / done words
hipError_t cl_launch_popoa_linear_span(uint32_t n_blocks, const ClDeviceBatch& B, const uint32_t* plist, const ClScoreParams& P, uint32_t* sync, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(popoa_linear_span_kernel, dim3(n_blocks), dim3(64 * kSpanW), 0, stream, B, plist, P, sync);
    return hipGetLastError();
}
uint32_t cl_linear_span_groups(uint32_t nr) { return ((nr + 63u) / 64u + kSpanW - 1u) / kSpanW; }

hipError_t cl_launch_popoa_linear(int W, uint32_t n_blocks, const ClDeviceBatch& B, const uint32_t* plist,
                                  const ClScoreParams& P, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    switch (W) {
    case 0: hipLaunchKernelGGL(popoa_linear_quad_kernel, dim3(n_blocks / 4), dim3(64), 0, stream, B, plist, P); break;   // four small pairs per wave: n_blocks list entries, four per workgroup
    case -2: hipLaunchKernelGGL(popoa_linear_duo_kernel, dim3(n_blocks / 2), dim3(64), 0, stream, B, plist, P); break;   // two pairs of 17-32 rows per wave
    case 1: hipLaunchKernelGGL((popoa_linear_kernel<1>), dim3(n_blocks), dim3(64), 0, stream, B, plist, P); break;
    case 3: hipLaunchKernelGGL((popoa_linear_kernel<3>), dim3(n_blocks), dim3(192), 0, stream, B, plist, P); break;
    case 4: hipLaunchKernelGGL((popoa_linear_kernel<4>), dim3(n_blocks), dim3(256), 0, stream, B, plist, P); break;
    case 8: hipLaunchKernelGGL((popoa_linear_kernel<8>), dim3(n_blocks), dim3(512), 0, stream, B, plist, P); break;
    case 16: hipLaunchKernelGGL((popoa_linear_kernel<16>), dim3(n_blocks), dim3(1024), 0, stream, B, plist, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
