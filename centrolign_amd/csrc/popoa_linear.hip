// popoa_linear.hip — register-resident wavefront kernel for chain x chain subproblems (every subproblem
// of a pairwise alignment; SURVEY.md §6: 100 % of the 2 x 1 Mbp batch).
//
// Same recurrences and tie-breaks as popoa_general_kernel / the reference's po_poa_internal
// (include/centrolign/alignment.hpp:753-1151), specialised to graphs whose nodes form one path
// (previous(r) == {r-1}, one source = rank 0, one sink = last rank):
//
//   * a wave owns a STRIP of 64*R consecutive rows (graph-1 nodes), lane l holds rows l*R .. l*R+R-1;
//   * it sweeps the columns (graph-2 nodes) as a systolic array: at step t lane l works on column
//     t - l + 1, so the cell above (lane l-1, one step earlier) and the diagonal cell (two steps earlier)
//     arrive through a single wave_shr:1 DPP move per value; graph-2 labels travel the same way;
//   * scores never leave registers.  What goes to HBM is one TRACEBACK CODE per cell (1 byte for
//     NumPW <= 2, 2 bytes for NumPW == 3) holding exactly the decisions the reference's traceback would
//     re-derive from its int32 planes with == tests (alignment.hpp:1048-1136):
//         bits 0-2  gap-close choice at this cell in the order I_0, D_0, I_1, D_1, I_2, D_2 (0 = none/diagonal)
//         bit 3+k   I_k(a,b) was reached by OPENING from Mf(a-1,b) (tested before extend, :1107-1117)
//         bit 3+NumPW+k  same for D_k
//     written as one coalesced run per wave and step;
//   * strips of one matrix are pipelined over the W waves of the workgroup: strip s+1 trails strip s by two
//     64-step chunks and receives the last row of strip s (Mf, I_k per column) through a small HBM/L2
//     buffer, one coalesced 64-column load per chunk;
//   * the boundary row/column are closed forms for a chain (-(open_k) - len*extend_k, alignment.hpp:832-894),
//     so they are never stored; lane 0 of wave 0 walks the traceback over the codes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "popoa_device.h"

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

struct LinearGeom {
    uint32_t n1, n2, S, Cn;
    template <int R>
    __device__ void init(uint32_t n1_, uint32_t n2_) {
        n1 = n1_; n2 = n2_;
        S = (n1 + 64 * R - 1) / (64 * R);
        Cn = (n2 + 63 + 63) / 64;  // steps 0 .. n2+62
    }
};

// Wave-cooperative traceback over the codes; same walk as alignment.hpp:1036-1138 for a chain pair, but the
// 64 lanes of wave 0 read the codes of the next 64 cells ALONG THE CURRENT DIRECTION (diagonal while in the
// match state, up while in an I_k gap, left while in a D_k gap) with one coalesced-by-direction load, find
// with a ballot where the run ends, and emit the whole run at once.  A traceback of length L with g gap
// events costs about L/64 + 3g dependent memory round trips instead of L.
template <int NPW, int R>
__device__ void linear_traceback(const ClDeviceBatch& B, const ClProbDesc& pd, const LinearGeom& G,
                                 const typename CodeT<NPW>::type* codes, const ClScoreParams& P, uint32_t prob,
                                 uint32_t lane) {
    uint32_t a = pd.n1, b = pd.n2, len = 0, status = 0;
    const uint32_t cap = pd.n1 + pd.n2;
    uint2* out = B.out_pairs + pd.out_base;
    int comp = 0;
    auto code_at = [&](uint32_t ca, uint32_t cb) -> uint32_t {
        const uint32_t row = ca - 1, s = row / (64 * R), rr = row - s * 64 * R, l = rr / R, r = rr - l * R;
        const uint32_t t = (cb - 1) + l;
        return codes[(((size_t)s * G.Cn * 64 + t) * 64 + l) * R + r];
    };
    while (true) {
        if (len > cap) { status = 2; break; }
        if (a && b) {
            if (comp == 0) {
                // diagonal run: lanes look at (a-j, b-j)
                const bool valid = lane < a && lane < b;
                const uint32_t code = valid ? code_at(a - lane, b - lane) : 0u;
                const uint32_t cc = code & 7u;
                const unsigned long long stop = __ballot(!valid || cc != 0u);
                const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                if (lane < first) out[cap - 1 - (len + lane)] = make_uint2(a - lane, b - lane);
                len += first;
                a -= first; b -= first;
                if (first < 64u && a && b) {
                    // a gap closes at this cell (first hit in the order I_0, D_0, I_1, ... ; :1048-1066)
                    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)cc, (int)first);
                    comp = (c1 & 1u) ? (int)((c1 + 1) >> 1) : -(int)(c1 >> 1);
                } else if (a == 0 && b == 0) {
                    break;  // emitted (1,1): the stored corner is -inf, no predecessor matches (:1091-1099)
                }
            } else if (comp > 0) {
                // I_k run: lanes look at (a-j, b); the run ends at the first cell reached by OPENING (:1105-1118)
                const bool valid = lane < a;
                const uint32_t code = valid ? code_at(a - lane, b) : 0u;
                const bool open = valid && ((code >> (3 + comp - 1)) & 1u);
                const unsigned long long mo = __ballot(open), mi = __ballot(!valid);
                const uint32_t fo = mo ? (uint32_t)__builtin_ctzll(mo) : 64u, fi = mi ? (uint32_t)__builtin_ctzll(mi) : 64u;
                const uint32_t n = fo < fi ? fo + 1 : fi;
                if (lane < n) out[cap - 1 - (len + lane)] = make_uint2(a - lane, 0u);
                len += n; a -= n;
                if (fo < fi) comp = 0;
            } else {
                const bool valid = lane < b;
                const uint32_t code = valid ? code_at(a, b - lane) : 0u;
                const bool open = valid && ((code >> (3 + NPW - comp - 1)) & 1u);
                const unsigned long long mo = __ballot(open), mi = __ballot(!valid);
                const uint32_t fo = mo ? (uint32_t)__builtin_ctzll(mo) : 64u, fi = mi ? (uint32_t)__builtin_ctzll(mi) : 64u;
                const uint32_t n = fo < fi ? fo + 1 : fi;
                if (lane < n) out[cap - 1 - (len + lane)] = make_uint2(0u, b - lane);
                len += n; b -= n;
                if (fo < fi) comp = 0;
            }
        } else if (a == 0 && b == 0) {
            status = 3; break;
        } else {
            // boundary row (a == 0) or column (b == 0): closed forms, M = max_k G_k, G_k = -(open_k) - x*extend_k,
            // the other gap family is -inf.  Walk exactly as :1048-1066 / :1101-1137 do on those cells.
            const bool row = a == 0;
            uint32_t x = row ? b : a;
            if (comp == 0) {
                const int32_t Mv = boundary_m<NPW>(P, x);
#pragma unroll
                for (int k = NPW - 1; k >= 0; --k) if (Mv == boundary_gap<NPW>(P, k, x)) comp = row ? -k - 1 : k + 1;
            }
            if (row ? comp >= 0 : comp <= 0) { status = 3; break; }
            const int k = row ? -comp - 1 : comp - 1;
            // lanes look at x-j; the run ends after the first cell whose gap value also equals an OPEN from the
            // previous boundary cell, or at x == 1 (its predecessor is the corner: nothing matches)
            const bool valid = lane < x;
            const uint32_t xj = x - lane;
            const bool open = valid && xj >= 2 && boundary_gap<NPW>(P, k, xj) == boundary_m<NPW>(P, xj - 1) - P.oe[k];
            const unsigned long long mo = __ballot(open), mi = __ballot(!valid);
            const uint32_t fo = mo ? (uint32_t)__builtin_ctzll(mo) : 64u, fi = mi ? (uint32_t)__builtin_ctzll(mi) : 64u;
            const uint32_t n = fo < fi ? fo + 1 : fi;
            if (lane < n) out[cap - 1 - (len + lane)] = row ? make_uint2(0u, xj) : make_uint2(xj, 0u);
            len += n; x -= n;
            if (row) b = x; else a = x;
            if (fo < fi) comp = 0;
            else if (x == 0) break;  // emitted the cell next to the corner
        }
    }
    if (lane == 0) {
        B.out_len[prob] = len > cap ? cap : len;
        B.out_status[prob] = status;
    }
}

template <int NPW, int R, int W>
__global__ void __launch_bounds__(64 * W) popoa_linear_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist,
                                                              ClScoreParams P) {
    using code_t = typename CodeT<NPW>::type;
    using pack_t = typename PackT<R * sizeof(code_t)>::type;
    const uint32_t prob = plist[blockIdx.x];
    const ClProbDesc pd = B.desc[prob];
    LinearGeom G;
    G.init<R>(pd.n1, pd.n2);
    const uint32_t n1 = pd.n1, n2 = pd.n2;
    const uint8_t* lab1 = B.lab[0] + pd.node_base[0];
    const uint8_t* lab2 = B.lab[1] + pd.node_base[1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t steps = G.Cn * 64;  // steps per strip, padded to whole chunks
    code_t* codes = reinterpret_cast<code_t*>(B.planes + pd.plane_base);
    // hand-off rows between strips: [strip][1 + NPW][steps] int32, after the codes (16-byte aligned by the packer)
    const size_t code_bytes = ((size_t)G.S * steps * 64 * R * sizeof(code_t) + 15) & ~(size_t)15;
    int32_t* brow = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(codes) + code_bytes);

    const uint32_t Pm = G.Cn > 2u * W ? G.Cn : 2u * W;  // macro-step period of one round of W strips
    const uint32_t total = ((G.S - 1) / W) * Pm + 2 * ((G.S - 1) % W) + G.Cn;

    // per-strip register state
    int32_t Mleft[R], Dleft[R][NPW], lab1r[R];
    int32_t lastM = 0, lastI[NPW], prevUpM = 0, c2 = 0xff;
    int32_t outv[1 + NPW];
    int32_t bM = 0, bI[NPW], myc2 = 0xff;
    int32_t my_score = 0;
#pragma unroll
    for (int k = 0; k < NPW; ++k) { lastI[k] = CL_NEG_INF; bI[k] = CL_NEG_INF; }
#pragma unroll
    for (int i = 0; i < 1 + NPW; ++i) outv[i] = 0;

    for (uint32_t m = 0; m < total; ++m) {
        const int32_t mm = (int32_t)m - 2 * (int32_t)wave;
        if (mm >= 0) {
            const uint32_t j = (uint32_t)mm / Pm, c = (uint32_t)mm - j * Pm, s = j * W + wave;
            if (s < G.S && c < G.Cn) {
                const uint32_t row0 = s * 64 * R + lane * R;  // 0-based first row of this lane; a = row + 1
                if (c == 0) {
                    // boundary column (b == 0): Mf(a,0) closed form, D_k(a,0) = -inf (alignment.hpp:832-862)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t a = row0 + r + 1;
                        Mleft[r] = boundary_m<NPW>(P, a);
#pragma unroll
                        for (int k = 0; k < NPW; ++k) Dleft[r][k] = CL_NEG_INF;
                        lab1r[r] = a <= n1 ? (int32_t)(lab1[a - 1] & 0x7f) : 0xfe;
                    }
                    lastM = Mleft[R - 1];
                    // diagonal term of the lane's first row at column 1: Mf(a-1, 0); the corner counts as 0 (:814-818)
                    prevUpM = row0 == 0 ? 0 : boundary_m<NPW>(P, row0);
                    c2 = 0xff;
                }
                const uint32_t t0 = c * 64;
                {   // this chunk's 64 columns as seen by lane 0: labels and the row above the strip
                    const uint32_t colb = t0 + lane + 1;
                    const bool v = colb <= n2;
                    myc2 = v ? (int32_t)(lab2[colb - 1] & 0x7f) : 0xff;
                    if (s == 0) {
                        bM = v ? boundary_m<NPW>(P, colb) : CL_NEG_INF;  // boundary row: Mf(0,b), I_k(0,b) = -inf
                    } else {
                        const int32_t* src = brow + (size_t)(s - 1) * (1 + NPW) * steps + (colb - 1);
                        bM = v ? src[0] : CL_NEG_INF;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) bI[k] = v ? src[(size_t)(1 + k) * steps] : CL_NEG_INF;
                    }
                }
                pack_t* cw = reinterpret_cast<pack_t*>(codes) + ((size_t)s * steps + t0) * 64 + lane;
                int32_t* bout = brow + (size_t)s * (1 + NPW) * steps;
                const bool hand_off = s + 1 < G.S;
#pragma unroll 2
                for (uint32_t jj = 0; jj < 64; ++jj) {
                    const uint32_t t = t0 + jj;
                    const int32_t upM = shift_in(lastM, __builtin_amdgcn_readlane(bM, jj));
                    int32_t upI[NPW];
#pragma unroll
                    for (int k = 0; k < NPW; ++k) upI[k] = shift_in(lastI[k], __builtin_amdgcn_readlane(bI[k], jj));
                    c2 = shift_in(c2, __builtin_amdgcn_readlane(myc2, jj));
                    const uint32_t b = t - lane + 1;  // this lane's column (1-based); wraps when not started
                    if ((uint32_t)(b - 1) < n2) {
                        int32_t diag = prevUpM, uM = upM, uI[NPW];
#pragma unroll
                        for (int k = 0; k < NPW; ++k) uI[k] = upI[k];
                        uint64_t pack = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const int32_t sc = lab1r[r] == c2 ? P.match : -P.mismatch;
                            int32_t Mf = diag + sc;
                            int32_t I[NPW], D[NPW];
                            uint32_t code = 0;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) {
                                const int32_t io = uM - P.oe[k], dopen = Mleft[r] - P.oe[k];
                                I[k] = imax(io, uI[k] - P.ext[k]);
                                D[k] = imax(dopen, Dleft[r][k] - P.ext[k]);
                                code |= (I[k] == io ? 1u : 0u) << (3 + k);
                                code |= (D[k] == dopen ? 1u : 0u) << (3 + NPW + k);
                                Mf = imax(Mf, imax(I[k], D[k]));
                            }
                            uint32_t cc = 0;
#pragma unroll
                            for (int k = NPW - 1; k >= 0; --k) {  // lowest k, I before D, wins (:1048-1066)
                                cc = Mf == D[k] ? 2u * k + 2u : cc;
                                cc = Mf == I[k] ? 2u * k + 1u : cc;
                            }
                            code |= cc;
                            pack |= (uint64_t)code << (8 * sizeof(code_t) * r);
                            if (row0 + r + 1 == n1 && b == n2) my_score = Mf;
                            diag = Mleft[r];
                            Mleft[r] = Mf;
                            uM = Mf;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) { Dleft[r][k] = D[k]; uI[k] = I[k]; }
                        }
                        lastM = uM;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) lastI[k] = uI[k];
                        cw[(size_t)jj * 64] = (pack_t)pack;
                    }
                    prevUpM = upM;
                    if (hand_off) {
                        // last row of the strip (lane 63, row R-1) has just finished column t-62
                        const uint32_t b63 = t - 62;
                        if ((uint32_t)(b63 - 1) < n2) {
                            const uint32_t slot = (b63 - 1) & 63u;
                            const int32_t vM = __builtin_amdgcn_readlane(lastM, 63);
                            outv[0] = lane == slot ? vM : outv[0];
#pragma unroll
                            for (int k = 0; k < NPW; ++k) {
                                const int32_t vI = __builtin_amdgcn_readlane(lastI[k], 63);
                                outv[1 + k] = lane == slot ? vI : outv[1 + k];
                            }
                            if (slot == 63u || b63 == n2) {
                                const uint32_t col = (b63 - 1) - slot + lane;  // 0-based column held by this lane
                                if (lane <= slot) {
#pragma unroll
                                    for (int i = 0; i < 1 + NPW; ++i) bout[(size_t)i * steps + col] = outv[i];
                                }
                            }
                        }
                    }
                }
                if (s + 1 == G.S && c + 1 == G.Cn) {
                    const uint32_t rr = (n1 - 1) - s * 64 * R;
                    if (lane == rr / R) B.out_score[prob] = my_score;
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) linear_traceback<NPW, R>(B, pd, G, codes, P, prob, lane);
}

template <int NPW, int R>
hipError_t launch_w(int W, uint32_t n_blocks, const ClDeviceBatch& B, const uint32_t* plist, const ClScoreParams& P,
                    hipStream_t stream) {
    switch (W) {
    case 1: hipLaunchKernelGGL((popoa_linear_kernel<NPW, R, 1>), dim3(n_blocks), dim3(64), 0, stream, B, plist, P); break;
    case 4: hipLaunchKernelGGL((popoa_linear_kernel<NPW, R, 4>), dim3(n_blocks), dim3(256), 0, stream, B, plist, P); break;
    case 16: hipLaunchKernelGGL((popoa_linear_kernel<NPW, R, 16>), dim3(n_blocks), dim3(1024), 0, stream, B, plist, P); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int NPW>
hipError_t launch_r(int R, int W, uint32_t n_blocks, const ClDeviceBatch& B, const uint32_t* plist,
                    const ClScoreParams& P, hipStream_t stream) {
    switch (R) {
    case 1: return launch_w<NPW, 1>(W, n_blocks, B, plist, P, stream);
    case 2: return launch_w<NPW, 2>(W, n_blocks, B, plist, P, stream);
    case 4: return launch_w<NPW, 4>(W, n_blocks, B, plist, P, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

// bytes of workspace a linear problem needs (codes + strip hand-off rows); mirrors the kernel's layout
size_t cl_linear_workspace_bytes(uint32_t n1, uint32_t n2, int npw, int R) {
    const size_t S = (n1 + 64 * (size_t)R - 1) / (64 * (size_t)R);
    const size_t steps = ((n2 + 63 + 63) / 64) * (size_t)64;
    const size_t csize = npw == 3 ? 2 : 1;
    const size_t code_bytes = (S * steps * 64 * R * csize + 15) & ~(size_t)15;
    const size_t brow_bytes = S * (size_t)(1 + npw) * steps * sizeof(int32_t);
    return code_bytes + brow_bytes;
}

hipError_t cl_launch_popoa_linear(int npw, int R, int W, uint32_t n_blocks, const ClDeviceBatch& B,
                                  const uint32_t* plist, const ClScoreParams& P, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    switch (npw) {
    case 1: return launch_r<1>(R, W, n_blocks, B, plist, P, stream);
    case 2: return launch_r<2>(R, W, n_blocks, B, plist, P, stream);
    case 3: return launch_r<3>(R, W, n_blocks, B, plist, P, stream);
    default: return hipErrorInvalidValue;
    }
}
