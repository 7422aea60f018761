// popoa_lane.h — popoa_lane_kernel: the register / DPP systolic sweep of popoa_linear_kernel (popoa_linear.hip) for NEAR-CHAIN graph pairs.  Included by
// popoa_kernels.hip inside its anonymous namespace (DiagGeom, Planes, plane_store, traceback_wave are that file's).
//
// Same recurrences as every other PO-POA kernel here — the reference's po_poa_internal in pull form (include/centrolign/alignment.hpp:813-938; SURVEY.md
// Appendix A) — for the pairs that bound a stitch pass of a progressive MSA: the long sweeps of BASELINE configs[2] are all pairs of graphs that are chains
// but for a few SNP / short-indel bubbles (predecessors 1-3 ranks back, in-degree 2, rarely 3) and one or two forks in front of a bubble as long as a repeat
// unit (a predecessor 2 059 ranks back).  popoa_sys_kernel gives such a pair a thread per row, an LDS ring per row and a workgroup barrier per step: 0.6-0.8 us
// per step, 70 % of a wave's cycles parked on the barrier / LDS round trip (profiles/r04_pmc_summary.json).  Here nothing but registers and DPP moves is touched
// in a step of the usual cell, and waves never wait for one another inside a chunk of 32 steps:
//   * rows (the shorter graph) lie across the lanes, strips of 64 rows, one row per lane; lane l of a strip works on column t - l + 1 at step t;
//   * ROW predecessors (1 .. DR ranks back) arrive on a CONVEYOR: stage d of lane l holds M and the row-consuming gap values V_k of row a - d at the lane's
//     current column — stage 1 is wave_shr:1 of the upper lane's last cell, stage d wave_shr:1 of the upper lane's stage d - 1 one step earlier;
//   * COLUMN predecessors 1 .. 3 columns back are the lane's own history (M, H_k), the diagonal terms the history of the conveyor's M: RINGS of four registers whose
//     slot numbers are compile-time constants (the step loop is unrolled by four), so that nothing is moved to age a value — a conveyor move writes straight into the
//     ring slot of its step;
//   * the usual step — every active lane on a chain row at a chain column: one predecessor one back on both sides, no source, no saved column — is decided per step by
//     one ballot and takes a straight-line cell of ~25 instructions (popoa_linear_kernel's); any other step takes the general cell below (masks over the distances);
//   * nothing in a chunk of 32 steps loads from global memory: the chunk's column records are fetched one chunk ahead, and a strip's last DR rows reach the next strip
//     of the same round through a window of 128 columns in LDS (the next ROUND's first strip, and a WIDE pair's next group, read them from the area behind the planes);
//     on gfx9 loads and stores share one counter, a load in the loop would wait for every plane store in front of it;
//     a column that a later column reaches from further away is a SAVED column as in popoa_sys_kernel: its cells go to LDS ([slot][row]: M, H_k), and only
//     lanes whose column has such a predecessor read them (a wave-divergent branch taken a few dozen steps per pair);
//   * the boundary row and column are closed forms: a boundary cell's gap value is -(open_k + extend_k * L) with L the number of nodes on the shortest walk
//     from a source (alignment.hpp:832-894 on a DAG: the maximum over walks of a value that only depends on the walk's length); the host packs L per node
//     (topology, no DP arithmetic), the prologue writes those cells' planes;
//   * a node's predecessors are a MASK over the distances (any subset of 1 .. DR / 1 .. 3), a source flag (the boundary index is a predecessor) and, for
//     columns, up to two saved columns: every maximum below runs over all distances with the absent ones forced to -inf, so the usual cell is branch-free;
//   * strips are pipelined over the W waves of the workgroup exactly as in popoa_linear_kernel: chunks of 32 steps, one barrier per chunk, strip s + 1
//     three chunks behind strip s, the last DR rows of a strip handed on through a small area behind the planes (M, V_k per column);
//   * the int32 planes go to HBM anti-diagonal-major, fire and forget, as in popoa_sys_kernel: traceback_wave reads them afterwards.
//   * WIDE pairs (more rows than one workgroup's eight waves take in a round or two — chain pairs of 4 096 rows and more, branching pairs above 1 024 rows): the strips are
//     dealt to GROUPS of eight, one workgroup per group, all groups of a pair consecutive workgroups of one launch on different compute units.  Group g's first strip
//     follows group g - 1's last strip exactly as a strip follows its upper neighbour inside a workgroup, three chunks behind — across compute units that takes a
//     PROGRESS word per group ("chunks my last strip has finished": release fence at agent scope, then a relaxed store) which wave 0 of the next group polls before every
//     chunk (then an agent-scope acquire: the compute unit's L1 is not refreshed by another unit's stores); the saved-column cells of a group's last DR rows travel
//     through a small area behind the hand-off rows.  Every wait is bounded: a group that gives up marks itself failed, the marks propagate, the pair reports status 9
//     and cl_stitch_plan_collect runs it again on the anti-diagonal kernel (as for popoa_strip_kernel).  The last group waits for every group's "done" mark and runs
//     the traceback.
// Model of the data movement, checked against the plain pull-form DP: scripts/dev/nearchain_model.py.
//
// Records (host: cl_api.cpp), uint32 each, at ClDeviceBatch::aux + ClProbDesc::aux_base: sync | rowrec[nR] | rowdist[nR] | colrec[nC] | coldist[nC]
//   sync  : a WIDE pair's first progress word in the plan's lane_sync array (progress[groups] | done[groups]); 0 otherwise
//   rowrec: bits 0-3 predecessor distances (bit d - 1) | bit 4 source | bits 8-14 label
//   colrec: bits 0-2 near predecessor distances (1 .. 3 columns back) | bit 4 source | bits 5-6 number of saved-column predecessors | bits 8-14 label | bit 15 this column is saved,
//           bits 16-19 in that slot | bits 20-23, 24-27 the slots of its saved-column predecessors
//   *dist : nodes on the shortest walk from a source to the node, the node included
// ClProbDesc::pad: bit 15 rows = graph 2 | bits 0-3 DR needed | bits 4-7 DC needed | bits 8-14 groups - 1 (0: one workgroup);
// aux_cnt = saved columns (LDS: aux_cnt * (DR + rows of a workgroup + 1) * (1 + NumPW) ints)

constexpr uint32_t kLaneChunk = 32;
constexpr uint32_t kLaneLag = 2 + 62 / kLaneChunk;
constexpr uint32_t kLaneGroupStrips = 3;   // WIDE: strips (= active waves) per workgroup.  Measured on one pair alone: one, two or three active waves of a workgroup run a step
                                           // in 0.24 / 0.30 / 0.30 us, FOUR in 0.55 us (two of them end up sharing a SIMD's issue slots): a group is three strips
constexpr uint32_t kLaneWindow = 128;      // columns of hand-off rows kept in LDS per strip boundary: the consumer reads chunk c while the producer writes chunk c + lag
constexpr uint32_t kLaneFailed = 0xFFFFFFFFu;
constexpr uint32_t kLanePolls = 1u << 20;   // x ~1 us: a group waits about a second for its upper neighbour before it gives the pair up

// lane l <- lane l-1 ; lane 0 <- fill (wave_shr:1) / lane l <- lane l+1 (wave_shl:1): gfx9-family whole-wave DPP controls, as in popoa_linear.hip
__device__ __forceinline__ int32_t lane_shift_in(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t lane_rotate_down(int32_t v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }

template <int NPW>
__device__ __forceinline__ int32_t lane_bnd_gap(const ClScoreParams& P, int k, uint32_t len) { return -P.oe[k] - (int32_t)(len - 1) * P.ext[k]; }
template <int NPW>
__device__ __forceinline__ int32_t lane_bnd_m(const ClScoreParams& P, uint32_t len) {
    int32_t m = lane_bnd_gap<NPW>(P, 0, len);
#pragma unroll
    for (int k = 1; k < NPW; ++k) m = imax(m, lane_bnd_gap<NPW>(P, k, len));
    return m;
}

template <int Q> struct LaneQ { static constexpr int value = Q; };

// LDS of a workgroup: hand-off window [W - 1][DR][1 + NPW][kLaneWindow] ints (W > 1) | saved columns [aux_cnt][DR + rows + 1][1 + NPW]
template <int NPW, int DR, int W, bool WIDE>
__device__ __forceinline__ void lane_body(const ClDeviceBatch& B, const ClProbDesc& pd, uint32_t prob, const ClScoreParams& P, int32_t* __restrict__ lds,
                                          uint32_t grp, uint32_t* __restrict__ lane_sync) {
    constexpr uint32_t C = kLaneChunk;
    constexpr int CW = 1 + NPW;   // a handed-on / saved cell: M, then V_k (hand-off) or H_k (saved column)
    const bool swap = pd.pad & 0x8000u;
    const uint32_t nR = swap ? pd.n2 : pd.n1, nC = swap ? pd.n1 : pd.n2;
    const uint32_t* const rowrec = B.aux + pd.aux_base + 1;
    const uint32_t* const rowdist = rowrec + nR;
    const uint32_t* const colrec = rowdist + nR;
    const uint32_t* const coldist = colrec + nC;
    const DiagGeom G(pd.n1, pd.n2);
    Planes<NPW> pl;
    pl.base = B.planes + pd.plane_base;
    pl.cells = (pd.n1 + 1) * (pd.n2 + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // V = the gap family that consumes ROW nodes, H = column nodes: I / D of the reference when the rows are graph 1, D / I when they are graph 2
    g_i32* pV[NPW];
    g_i32* pH[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        pV[k] = uniform_plane(pl.base + (size_t)(swap ? 1 + NPW + k : 1 + k) * pl.cells);
        pH[k] = uniform_plane(pl.base + (size_t)(swap ? 1 + k : 1 + NPW + k) * pl.cells);
    }
    g_i32* const pM = uniform_plane(pl.base);
    // hand-off rows between strips that are NOT neighbours in one round, behind the planes: [strip][DR][1 + NPW][nC] int32
    int32_t* const brow = pl.base + (((size_t)pl.cells * (1 + 2 * NPW) + 3) & ~(size_t)3);
    const uint32_t S = (nR + 63u) / 64u, Cn = (nC + 63u + C - 1) / C;
    // WIDE: this workgroup is group `grp` of n_groups and takes the strips grp * W .. in one round; rows_here = the rows an LDS slot of saved cells spans
    const uint32_t n_groups = WIDE ? ((pd.pad >> 8) & 0x7Fu) + 1u : 1u;
    constexpr uint32_t GS = WIDE ? kLaneGroupStrips : (uint32_t)W;   // strips of a round / group (WIDE: one wave of the workgroup stays idle)
    const uint32_t rowbase = WIDE ? grp * GS * 64u : 0u;
    const uint32_t area = (DR + (WIDE ? GS * 64u : nR) + 1u) * CW;     // ints per saved column: [0] the boundary row's Mf, [DR - d] ghost rows (unused), [DR + local row]
    int32_t* const hand = lds;                                        // the LDS hand-off window
    int32_t* const saved = lds + (W > 1 ? (W - 1) * DR * CW * kLaneWindow : 0);
    uint32_t* const progress = WIDE ? lane_sync + B.aux[pd.aux_base] : nullptr;
    uint32_t* const done = WIDE ? progress + n_groups : nullptr;
    // saved-column cells (M) of a group's last DR rows, for the next group's first rows: behind the hand-off rows, [group][slot][DR]
    // (kept in the GLOBAL address space: a generic pointer here lets the optimiser fold the "ghost row or LDS" choice below into ONE flat load from a selected address,
    // and the backend then emits an illegal compare against src_shared_base)
    g_i32* const sx = (g_i32*)(brow + (size_t)(S > 0 ? S - 1 : 0) * DR * CW * nC);
    __shared__ uint32_t gave_up;
    if (WIDE) { if (tid == 0) gave_up = 0; __syncthreads(); }
    auto cell_index = [&](uint32_t row, uint32_t col) { return swap ? G.idx(col, row) : G.idx(row, col); };

    // ---- prologue: the boundary cells' planes (closed forms; the traceback reads them) ----
    if (!(B.skip_traceback & 2) && (!WIDE || grp == 0)) {
        for (uint32_t i = tid; i <= nR; i += 64 * W) {
            const uint32_t pb = cell_index(i, 0) * 4u;
            const uint32_t len = i ? rowdist[i - 1] : 0u;
            plane_store(pM, pb, i ? lane_bnd_m<NPW>(P, len) : CL_NEG_INF);
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                plane_store(pV[k], pb, i ? lane_bnd_gap<NPW>(P, k, len) : CL_NEG_INF);
                plane_store(pH[k], pb, CL_NEG_INF);
            }
        }
        for (uint32_t j = tid + 1; j <= nC; j += 64 * W) {
            const uint32_t pb = cell_index(0, j) * 4u;
            const uint32_t len = coldist[j - 1];
            plane_store(pM, pb, lane_bnd_m<NPW>(P, len));
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                plane_store(pV[k], pb, CL_NEG_INF);
                plane_store(pH[k], pb, lane_bnd_gap<NPW>(P, k, len));
            }
        }
    }

    const uint32_t Pm = Cn > kLaneLag * W ? Cn : kLaneLag * W;   // macro-step period of one round of W strips
    const uint32_t strips_here = WIDE ? (S - grp * GS < GS ? S - grp * GS : GS) : 0u;
    const uint32_t total = WIDE ? kLaneLag * (strips_here - 1) + Cn : ((S - 1) / W) * Pm + kLaneLag * ((S - 1) % W) + Cn;
    bool dead = false;   // WIDE: this group has given the pair up (its upper neighbour did not deliver in time, or failed itself)
    uint32_t seen = 0;   // WIDE, wave 0: the upper neighbour's progress as last read (no new read while it already covers the chunk)

    // ---- per-strip register state.  Rings: slot Q = the value of the current step (Q = step & 3), slot (Q - e) & 3 the value e steps ago ----
    int32_t Mh[4], Hh[NPW][4];        // the lane's own cells: Mf and H_k of the last columns
    int32_t cMh[DR][4];               // conveyor stage d: M of row (own - d - 1) at the lane's current column and at the last columns
    int32_t cV[DR][NPW], lastV[NPW];  // ... its V_k (current column only); the lane's own V_k of the last column
    int32_t bMr[4];                   // the boundary row's Mf at the lane's column, travelling with the column record
    int32_t crec = 0;
    int32_t frec = 0, fbm = CL_NEG_INF, nrec = 0, nbm = CL_NEG_INF, fM[DR], fV[DR][NPW];
    uint32_t rrec = 0, labR = 0, row = 0;      // this lane's row (1-based; 0 = none)
    uint32_t pidx = 0;                         // its current cell in the planes
    int32_t ownBnd = CL_NEG_INF, predBnd[DR];
    bool rowslow = false, strip_dr1 = true;
    auto reset_strip = [&]() {
        crec = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Mh[q] = CL_NEG_INF; bMr[q] = CL_NEG_INF;
#pragma unroll
            for (int k = 0; k < NPW; ++k) Hh[k][q] = CL_NEG_INF;
#pragma unroll
            for (int d = 0; d < DR; ++d) cMh[d][q] = CL_NEG_INF;
        }
#pragma unroll
        for (int k = 0; k < NPW; ++k) lastV[k] = CL_NEG_INF;
#pragma unroll
        for (int d = 0; d < DR; ++d) {
            fM[d] = CL_NEG_INF; predBnd[d] = CL_NEG_INF;
#pragma unroll
            for (int k = 0; k < NPW; ++k) { cV[d][k] = CL_NEG_INF; fV[d][k] = CL_NEG_INF; }
        }
    };
    reset_strip();

    for (uint32_t m = 0; m < total; ++m) {
        const int32_t mm = (int32_t)m - (int32_t)(kLaneLag * wave);
        if (mm >= 0 && !dead) {
            const uint32_t j = WIDE ? 0u : (uint32_t)mm / Pm, c = (uint32_t)mm - j * Pm, s = (WIDE ? grp * GS : j * W) + wave;
            if (s < S && c < Cn && (!WIDE || wave < GS)) {
                const bool real = (s * 64u + lane + 1u) <= nR;
                const uint32_t t0 = c * C;
                auto fetch_columns = [&](uint32_t first, int32_t& rec, int32_t& bm) {   // lanes 0 .. C - 1: the records of columns first + 1 ..
                    const uint32_t colb = first + lane + 1;
                    const bool v = lane < C && colb <= nC;
                    rec = v ? (int32_t)colrec[colb - 1] : 0;
                    bm = v ? lane_bnd_m<NPW>(P, coldist[colb - 1]) : CL_NEG_INF;
                };
                if (c == 0) {
                    // a new strip: this lane's row and what is fixed for it
                    reset_strip();
                    row = s * 64u + lane + 1u;
                    rrec = real ? rowrec[row - 1] : 0u;
                    labR = (rrec >> 8) & 0x7Fu;
                    ownBnd = real ? lane_bnd_m<NPW>(P, rowdist[row - 1]) : CL_NEG_INF;
#pragma unroll
                    for (int d = 0; d < DR; ++d) predBnd[d] = (real && row > (uint32_t)(d + 1)) ? lane_bnd_m<NPW>(P, rowdist[row - 2 - d]) : CL_NEG_INF;
                    if (row == 1u && real) {
                        // the first row's only predecessor is the boundary row (it is a source and nothing lies above it): for strip 0 the boundary row IS the row above
                        // the strip — lane 0's conveyor feed carries Mf(0, column) with no gap values to extend (alignment.hpp:907-916 with p == n1 opens only) — so the
                        // row is an ordinary chain row for the cell below, its corner term Mf(0, 0) counting 0 (:814-818)
                        rrec = (rrec & ~0x1Fu) | 1u;
                        predBnd[0] = 0;
                    }
                    rowslow = real && (rrec & 0x1Fu) != 1u;                           // anything but "one predecessor, the row above"
                    strip_dr1 = __ballot(real && (rrec & 0xEu)) == 0ull;               // no row of the strip reaches further than one row up: the conveyor's first stage does
                    fetch_columns(0, nrec, nbm);
                }
                // this chunk's columns were fetched a chunk ago; the next chunk's are asked for now and not waited for before the chunk is over
                frec = nrec; fbm = nbm;
                fetch_columns(t0 + C, nrec, nbm);
                const bool from_lds = W > 1 && s > 0 && (WIDE ? wave > 0 : (s % W) != 0);   // the rows above the strip: the previous wave's, through the LDS window ...
                bool starved = false;
                if (WIDE && grp > 0 && wave == 0) {
                    // chunk c of this group's first strip needs the columns of that chunk from the last rows of the group above: its last strip must have finished
                    // c + lag chunks (what the barrier per macro-step guarantees between neighbouring waves of one workgroup).  The hand-off words are written
                    // through to memory (sc1 stores) and read past the caches (sc1 loads): no fence runs while the sweep runs (as in popoa_strip_kernel)
                    // (the group STARTS two chunks later than it must: both groups advance at the same rate, so the word it reads is then usually two or three chunks
                    // ahead of the need, one read serves several chunks and hardly ever has to wait — a group that starts the moment it can finds its upper neighbour
                    // exactly at the need in every chunk and pays the publish + poll round trip, ~15 us, per chunk)
                    const uint32_t want = c == 0 ? c + kLaneLag + 2 : c + kLaneLag;
                    const uint32_t need = want < Cn ? want : Cn;
                    uint32_t polls = 0;
                    while (seen < need && seen != kLaneFailed) {
                        seen = __hip_atomic_load(progress + (grp - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen >= need || ++polls > kLanePolls) break;
                        __builtin_amdgcn_s_sleep(4);
                    }
                    starved = seen == kLaneFailed || seen < need;
                    if (starved && lane == 0) gave_up = 1;
                }
                if (!starved) {
                if (s == 0) fM[0] = fbm;   // ... for the first strip the boundary row (its gap values stay -inf: nothing extends out of it)
                if (s > 0) {
                    const uint32_t colb = t0 + lane + 1;
                    const bool v = lane < C && colb <= nC;
                    if (from_lds) {
                        const int32_t* src = hand + (size_t)(wave - 1) * DR * CW * kLaneWindow + ((colb - 1) & (kLaneWindow - 1));
#pragma unroll
                        for (int d = 0; d < DR; ++d) {
                            fM[d] = v ? src[(d * CW) * kLaneWindow] : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) fV[d][k] = v ? src[(d * CW + 1 + k) * kLaneWindow] : CL_NEG_INF;
                        }
                    } else {   // ... or the previous round's / group's last strip, from the area behind the planes
                        const int32_t* src = brow + (size_t)(s - 1) * DR * CW * nC + (colb - 1);
                        auto ld = [&](const int32_t* q) { return WIDE ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q; };   // (another compute unit wrote them)
#pragma unroll
                        for (int d = 0; d < DR; ++d) {
                            fM[d] = v ? ld(src + (size_t)(d * CW) * nC) : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) fV[d][k] = v ? ld(src + (size_t)(d * CW + 1 + k) * nC) : CL_NEG_INF;
                        }
                    }
                }
                int32_t* const bout = brow + (size_t)s * DR * CW * nC;
                const bool hands_on = s + 1 < S && lane >= 64u - DR;
                const bool to_lds = W > 1 && s + 1 < S && (WIDE ? wave + 1 < GS : ((s + 1) % W) != 0);
                int32_t* const hout = hand + (size_t)wave * DR * CW * kLaneWindow + (size_t)((63u - lane) * CW) * kLaneWindow;
                const uint32_t rmask = rrec & 0xFu;
                const bool rsrc = (rrec >> 4) & 1u;

                // one step of the sweep; Q = step & 3 names the ring slots.  DRS = conveyor stages that are run (1 when no row of the strip needs more)
                auto step = [&](auto qc, auto drs, uint32_t t) {
                    constexpr int Q = decltype(qc)::value, P1 = (Q + 3) & 3, P2 = (Q + 2) & 3, P3 = (Q + 1) & 3, DRS = decltype(drs)::value;
                    // the conveyor moves one lane: stage d takes the upper lane's stage d - 1 of the last step, stage 1 its last cell; lane 0 the rows above the strip
#pragma unroll
                    for (int d = DRS - 1; d > 0; --d) {
                        cMh[d][Q] = lane_shift_in(cMh[d - 1][P1], fM[d]);
#pragma unroll
                        for (int k = 0; k < NPW; ++k) cV[d][k] = lane_shift_in(cV[d - 1][k], fV[d][k]);
                    }
                    cMh[0][Q] = lane_shift_in(Mh[P1], fM[0]);
#pragma unroll
                    for (int k = 0; k < NPW; ++k) cV[0][k] = lane_shift_in(lastV[k], fV[0][k]);
                    fM[0] = lane_rotate_down(fM[0]);
                    if (s > 0) {
#pragma unroll
                        for (int k = 0; k < NPW; ++k) fV[0][k] = lane_rotate_down(fV[0][k]);
#pragma unroll
                        for (int d = 1; d < DRS; ++d) {
                            fM[d] = lane_rotate_down(fM[d]);
#pragma unroll
                            for (int k = 0; k < NPW; ++k) fV[d][k] = lane_rotate_down(fV[d][k]);
                        }
                    }
                    crec = lane_shift_in(crec, frec);
                    frec = lane_rotate_down(frec);
                    bMr[Q] = lane_shift_in(bMr[P1], fbm);
                    fbm = lane_rotate_down(fbm);
                    const uint32_t b = t - lane + 1;   // this lane's column (1-based); wraps while the lane has not started
                    const bool active = (uint32_t)(b - 1) < nC && real;
                    const uint32_t cr = (uint32_t)crec;
                    const bool slow = active && (rowslow || (cr & 0x7Fu) != 1u);
                    int32_t Mf = CL_NEG_INF, V[NPW], H[NPW];
                    if (__ballot(slow) == 0ull) {
                        // the usual step: every active lane on a chain row at a chain column (alignment.hpp:907-936 with one predecessor each)
                        const int32_t sc = (labR == ((cr >> 8) & 0x7Fu)) ? P.match : -P.mismatch;
                        Mf = cMh[0][P1] + sc;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) {
                            V[k] = imax(cMh[0][Q] - P.oe[k], cV[0][k] - P.ext[k]);
                            H[k] = imax(Mh[P1] - P.oe[k], Hh[k][P1] - P.ext[k]);
                            Mf = imax(Mf, imax(V[k], H[k]));
                        }
                    } else if (active) {
                        const uint32_t cmask = cr & 0x7u, nfar = (cr >> 5) & 3u;
                        const bool csrc = (cr >> 4) & 1u;
                        const int32_t sc = (labR == ((cr >> 8) & 0x7Fu)) ? P.match : -P.mismatch;
                        int32_t Md = CL_NEG_INF;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) { V[k] = CL_NEG_INF; H[k] = CL_NEG_INF; }
                        const bool c1 = cmask & 1u, c2 = cmask & 2u, c3 = cmask & 4u;
                        // row predecessors: M and V_k at this column from the conveyor, the diagonal terms from its history
#pragma unroll
                        for (int d = 0; d < DRS; ++d) {
                            const bool on = (rmask >> d) & 1u;
                            const int32_t mu = on ? cMh[d][Q] : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], imax(mu - P.oe[k], (on ? cV[d][k] : CL_NEG_INF) - P.ext[k]));
                            Md = imax(Md, (on && c1) ? cMh[d][P1] : CL_NEG_INF);
                            Md = imax(Md, (on && c2) ? cMh[d][P2] : CL_NEG_INF);
                            Md = imax(Md, (on && c3) ? cMh[d][P3] : CL_NEG_INF);
                            Md = imax(Md, (on && csrc) ? predBnd[d] : CL_NEG_INF);
                        }
                        // the boundary row as a predecessor: opens only (alignment.hpp:907-916 with p == n1), diagonal from Mf(0, q), the corner counts 0
                        {
                            const int32_t bnow = rsrc ? bMr[Q] : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) V[k] = imax(V[k], bnow - P.oe[k]);
                            Md = imax(Md, (rsrc && c1) ? bMr[P1] : CL_NEG_INF);
                            Md = imax(Md, (rsrc && c2) ? bMr[P2] : CL_NEG_INF);
                            Md = imax(Md, (rsrc && c3) ? bMr[P3] : CL_NEG_INF);
                            Md = imax(Md, (rsrc && csrc) ? 0 : CL_NEG_INF);
                        }
                        // column predecessors: the lane's own history; the boundary column opens only
                        {
                            const int32_t m1 = c1 ? Mh[P1] : CL_NEG_INF, m2 = c2 ? Mh[P2] : CL_NEG_INF, m3 = c3 ? Mh[P3] : CL_NEG_INF, m0 = csrc ? ownBnd : CL_NEG_INF;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) {
                                const int32_t ho = imax(imax(m1, m2), imax(m3, m0)) - P.oe[k];
                                const int32_t he = imax(imax(c1 ? Hh[k][P1] : CL_NEG_INF, c2 ? Hh[k][P2] : CL_NEG_INF), c3 ? Hh[k][P3] : CL_NEG_INF) - P.ext[k];
                                H[k] = imax(ho, he);
                            }
                        }
                        if (nfar) {   // saved-column predecessors (the fork in front of a long bubble): LDS, a few dozen steps per pair
                            for (uint32_t f = 0; f < nfar; ++f) {
                                const uint32_t slot = (cr >> (20 + 4 * f)) & 0xFu;
                                const int32_t* col = saved + (size_t)slot * area;
                                const int32_t* mine = col + (size_t)(DR + row - rowbase) * CW;
                                const int32_t ml = mine[0];
#pragma unroll
                                for (int k = 0; k < NPW; ++k) H[k] = imax(H[k], imax(ml - P.oe[k], mine[1 + k] - P.ext[k]));
#pragma unroll
                                for (int d = 0; d < DR; ++d)
                                    if ((rmask >> d) & 1u) {
                                        const uint32_t pr = row - 1 - d;   // the predecessor row: this workgroup's, or one of the last DR rows of the group above
                                        int32_t far_m;   // (two loads under a branch: a select between a global and an LDS address makes the backend emit an illegal compare)
                                        if (WIDE && grp > 0 && pr <= rowbase) far_m = __hip_atomic_load((const int32_t*)sx + ((size_t)(grp - 1) * pd.aux_cnt + slot) * DR + (rowbase - pr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        else far_m = col[(size_t)(DR + pr - rowbase) * CW];
                                        Md = imax(Md, far_m);
                                    }
                                if (rsrc) Md = imax(Md, col[0]);
                            }
                        }
                        Mf = Md + sc;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) Mf = imax(Mf, imax(V[k], H[k]));
                    }
                    if (active) {
                        // where the cell lies in the anti-diagonal-major planes: closed form at the lane's first column, then one step along the column axis —
                        // idx(a1, a2 + 1) - idx(a1, a2) = hi(d) + 1 - lo(d + 1) with d = a1 + a2, one more when the columns are graph 1 (DiagGeom)
                        if (b == 1) pidx = cell_index(row, 1);
                        if (!(B.skip_traceback & 2)) {
                            const uint32_t pb = pidx * 4u;
                            plane_store(pM, pb, Mf);
#pragma unroll
                            for (int k = 0; k < NPW; ++k) {
                                plane_store(pV[k], pb, V[k]);
                                plane_store(pH[k], pb, H[k]);
                            }
                        }
                        {
                            const uint32_t dg = row + b;
                            pidx += (dg < pd.n1 ? dg : pd.n1) + (swap ? 2u : 1u) - (dg + 1 > pd.n2 ? dg + 1 - pd.n2 : 0u);
                        }
                        // the new cell is this step's ring entry; the lane below takes M / V_k on its next step
                        Mh[Q] = Mf;
#pragma unroll
                        for (int k = 0; k < NPW; ++k) { Hh[k][Q] = H[k]; lastV[k] = V[k]; }
                        if ((cr >> 15) & 1u) {   // a saved column: its cells stay available for the far reads
                            const uint32_t slot = (cr >> 16) & 0xFu;
                            int32_t* w = saved + (size_t)slot * area + (size_t)(DR + row - rowbase) * CW;
                            w[0] = Mf;
#pragma unroll
                            for (int k = 0; k < NPW; ++k) w[1 + k] = H[k];
                            if (row == rowbase + 1) {
                                saved[(size_t)slot * area] = bMr[Q];                        // the boundary row's Mf at this column: for source rows ...
                                if (rowbase == 0) saved[(size_t)slot * area + DR * CW] = bMr[Q];   // ... and as "row 0" for row 1, whose row predecessor it is
                            }
                            if (WIDE && hands_on && wave + 1 == strips_here) __hip_atomic_store((int32_t*)sx + ((size_t)grp * pd.aux_cnt + slot) * DR + (63u - lane), Mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next group's first rows
                        }
                        if (hands_on) {   // the last DR rows of a full strip feed the next strip's conveyor
                            if (to_lds) {
                                int32_t* o = hout + ((b - 1) & (kLaneWindow - 1));
                                o[0] = Mf;
#pragma unroll
                                for (int k = 0; k < NPW; ++k) o[(1 + k) * kLaneWindow] = V[k];
                            } else {
                                int32_t* o = bout + (size_t)((63u - lane) * CW) * nC + (b - 1);
                                if (WIDE) {
                                    __hip_atomic_store(o, Mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                                    for (int k = 0; k < NPW; ++k) __hip_atomic_store(o + (size_t)(1 + k) * nC, V[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                } else {
                                    o[0] = Mf;
#pragma unroll
                                    for (int k = 0; k < NPW; ++k) o[(size_t)(1 + k) * nC] = V[k];
                                }
                            }
                        }
                    }
                };
                auto run_chunk = [&](auto drs) {
                    for (uint32_t jj = 0; jj < C; jj += 4) {
                        step(LaneQ<0>{}, drs, t0 + jj);
                        step(LaneQ<1>{}, drs, t0 + jj + 1);
                        step(LaneQ<2>{}, drs, t0 + jj + 2);
                        step(LaneQ<3>{}, drs, t0 + jj + 3);
                    }
                };
                if (DR > 1 && !strip_dr1) run_chunk(LaneQ<DR>{}); else run_chunk(LaneQ<1>{});
                }   // (not starved)
            }
        }
        if (WIDE && grp + 1 < n_groups && wave + 1 == strips_here && mm >= 0 && (uint32_t)mm < Cn && !dead) {
            // the last strip of the group has finished chunk mm: its hand-off rows (and saved-column cells) become visible to the other compute units, then the count
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave holds every hand-off row of the group: its write-through stores have arrived
            if (lane == 0) __hip_atomic_store(progress + grp, (uint32_t)mm + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (W > 1) __syncthreads();
        if (WIDE && grp > 0 && !dead && gave_up) {   // wave 0 did not get its columns in time (or the group above failed): the whole group gives the pair up
            dead = true;
            if (tid == 0) __hip_atomic_store(progress + grp, kLaneFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();   // vmcnt(0): every plane value is in memory
    if (!WIDE) {
        if (tid < 64 && !B.skip_traceback) traceback_wave<NPW>(B, pd, G, pl, P, prob);
        return;
    }
    // WIDE: every wave's planes out to the other compute units, then this group's mark; the last group collects the marks and walks back
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done + grp, dead ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (grp + 1 != n_groups || tid >= 64) return;
    bool ok = !dead;
    for (uint32_t g2 = 0; g2 + 1 < n_groups && ok; ++g2) {
        uint32_t v = 0, polls = 0;
        while ((v = __hip_atomic_load(done + g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++polls <= kLanePolls) __builtin_amdgcn_s_sleep(8);
        ok = v == 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!ok) {
        if (lane == 0) { B.out_len[prob] = 0; B.out_status[prob] = 9; }   // cl_stitch_plan_collect runs the pair again, on the anti-diagonal kernel
        return;
    }
    if (!B.skip_traceback) traceback_wave<NPW>(B, pd, G, pl, P, prob);
}

template <int NPW, int W, bool WIDE>
__device__ __forceinline__ void lane_dispatch(const ClDeviceBatch& B, const ClProbDesc& pd, uint32_t prob, const ClScoreParams& P, int32_t* lds, uint32_t grp, uint32_t* lane_sync) {
    // two shapes: row predecessors up to 2 ranks back (every pair of the 10 x 1 Mbp MSA), or up to 4; column predecessors up to 3 columns back in both
    if ((pd.pad & 0xFu) <= 2u) lane_body<NPW, 2, W, WIDE>(B, pd, prob, P, lds, grp, lane_sync);
    else lane_body<NPW, 4, W, WIDE>(B, pd, prob, P, lds, grp, lane_sync);
}

// WIDE: a pair's groups are consecutive workgroups of the launch (plist repeats the pair once per group); lane_sync: the plan's progress / done words
template <int W, bool WIDE>
__global__ void __launch_bounds__(64 * W) popoa_lane_kernel(ClDeviceBatch B, const uint32_t* __restrict__ plist, ClScoreParams P, uint32_t* lane_sync) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    cl_tick_start(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
    const uint32_t prob = plist[blockIdx.x];
    uint32_t grp = 0;
    if (WIDE) while (grp < blockIdx.x && plist[blockIdx.x - grp - 1] == prob) ++grp;
    const ClProbDesc pd = B.desc[prob];
    switch (pd.npw) {
    case 1: lane_dispatch<1, W, WIDE>(B, pd, prob, P, lds, grp, lane_sync); break;
    case 2: lane_dispatch<2, W, WIDE>(B, pd, prob, P, lds, grp, lane_sync); break;
    default: lane_dispatch<3, W, WIDE>(B, pd, prob, P, lds, grp, lane_sync); break;
    }
    cl_tick_end(B, gridDim.x <= 4096u || (blockIdx.x & 63u) == 0);
}
