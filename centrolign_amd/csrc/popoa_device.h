// popoa_device.h — device-side data layout of a packed stitch batch (shared by host packer and kernels).
//
// Everything the kernels read is in TOPOLOGICAL-RANK space: the host packer (cl_api.cpp) orders every
// subgraph once (Kahn, as include/centrolign/topological_order.hpp:12-60 of the reference) and rewrites
// labels / predecessor lists / sinks in rank order, so that an anti-diagonal  a + b = d  of the
// (n1+1) x (n2+1) matrix only depends on earlier anti-diagonals.  Index 0 on each axis is the reference's
// extra boundary row/column ("nothing consumed yet", alignment.hpp:788-790), real node of rank r is r+1.
#ifndef CL_POPOA_DEVICE_H
#define CL_POPOA_DEVICE_H

#include <stdint.h>

#define CL_NEG_INF (INT32_MIN / 2)  // cell_t::mininf, alignment.hpp:740

// kernel families
enum { CL_KIND_GENERAL = 0, CL_KIND_LINEAR = 1, CL_KIND_SYS = 2, CL_KIND_STRIP = 3, CL_KIND_LANE = 4 };   // LANE: near-chain pairs in registers (popoa_lane_kernel, popoa_lane.h)   // SYS: the systolic DAG kernel (popoa_sys_kernel); STRIP: the same sweep for
                                                                                       // pairs whose rows do not fit one workgroup's LDS (popoa_strip_kernel)

struct ClProbDesc {
    uint32_t n1, n2;          // node counts (both > 0)
    uint32_t node_base[2];    // first node of this problem in lab[] / poff[] of each side
    uint32_t snk_base[2];     // first sink of this problem in snk[] of each side
    uint32_t snk_cnt[2];
    uint64_t plane_base;      // int32 index of this problem's first DP plane in the workspace
    uint32_t out_base;        // first pair slot of this problem in the alignment output (capacity n1+n2)
    uint8_t  npw;             // 1..3
    uint8_t  kind;            // CL_KIND_*
    uint16_t pad;
    uint32_t aux_base;        // systolic kernel: first entry of the problem's saved-column list in ClDeviceBatch::aux
    uint32_t aux_cnt;         // ... and its length
};

struct ClScoreParams {
    int32_t match;            // +match
    int32_t mismatch;         // -mismatch is applied by the kernels
    int32_t oe[3];            // gap_open[k] + gap_extend[k]
    int32_t ext[3];           // gap_extend[k]
};

// pointers to the packed batch in HBM
struct ClDeviceBatch {
    const ClProbDesc* desc;
    const uint8_t*  lab[2];   // label (low 7 bits) | 0x80 if the node is a source, rank order
    const uint32_t* poff[2];  // CSR offsets (global) of predecessor lists, rank order, node_count+1 entries per side
    const uint32_t* pidx[2];  // predecessor rank+1, in BaseGraph::previous() order
    const uint32_t* snk[2];   // sink rank+1, in SubGraphInfo::sinks order
    const uint32_t* aux;      // systolic kernel: per problem, the columns (0-based matrix columns, ascending) whose cells are kept for good
    int32_t*  planes;         // DP workspace
    uint2*    out_pairs;      // (a, b) with 0 = gap, rank+1 otherwise; each problem fills its slot range from the END
    uint32_t* out_len;        // pairs emitted per problem
    int32_t*  out_score;      // best sink-pair score per problem
    uint32_t* out_status;     // 0 = ok
    unsigned long long* ticks; // null, or [2]: pass << 48 | the low 48 bits of — max over the workgroups of — ~(start tick) and (end tick), 100 MHz ticks of
                              // s_memrealtime — the launch's duration by the kernel's own clock, also inside a step where launches overlap (launches of
                              // more than 4 096 workgroups sample every 64th: they are throughput-bound, their ends are within a wave of one another)
    uint32_t  tick_pass;      // the pass's number (1 .. 65 535), the top 16 bits of both tick words: a later pass's clocks replace an earlier pass's without a reset in between
    int       debug_span_fail; // test hook (CL_SPAN_DEBUG_FAIL=1): every chain pair that spans several workgroups reports failure (status 9), as if a wait had expired
    int       skip_traceback; // measurement hook (CL_DEBUG_SKIP_TRACEBACK=1, scripts/stitch_dag_bench.py): the graph x graph kernels fill only
};

#if defined(__HIPCC__)
constexpr unsigned long long kTickMask = 0xFFFFFFFFFFFFull;
// every kernel of the stitch path calls these as its first and last statement (`sampled`: which workgroups take part)
__device__ __forceinline__ void cl_tick_start(const ClDeviceBatch& B, bool sampled) {
    if (B.ticks && threadIdx.x == 0 && sampled) atomicMax(B.ticks, ((unsigned long long)B.tick_pass << 48) | (~(unsigned long long)__builtin_amdgcn_s_memrealtime() & kTickMask));
}
__device__ __forceinline__ void cl_tick_end(const ClDeviceBatch& B, bool sampled) {
    if (B.ticks && threadIdx.x == 0 && sampled) atomicMax(B.ticks + 1, ((unsigned long long)B.tick_pass << 48) | ((unsigned long long)__builtin_amdgcn_s_memrealtime() & kTickMask));
}
#endif

// popoa_strip_kernel: one workgroup per STRIP of consecutive rows of a large branching pair; strip j reads the last rows of strip j - 1 (its "ghost"
// rows) from a hand-off area in HBM that strip j - 1 fills while it runs
struct ClStripDesc {
    uint32_t prob;        // the problem (ClDeviceBatch::desc)
    uint32_t row_base;    // matrix row of timing index 0 (first ghost row; 0 in the first strip, whose timing index 0 is the boundary row)
    uint32_t n_real;      // rows this strip computes
    uint32_t n_ghost;     // rows in front of them that it reads from the hand-off area (0 in the first strip)
    uint32_t n_out;       // its last rows that it writes to the hand-off area (0 in the last strip)
    uint32_t rec_base;    // first column record of the problem (ClStripDevice::recs)
    uint64_t hand_in;     // first 64-bit word of the incoming hand-off rows [n_ghost][columns + 1][cell words / 2]
    uint64_t hand_out;    // ... of the outgoing ones [n_out][columns + 1][...]
    uint32_t prog;        // this strip's progress word; the strips of a problem are consecutive (prog - strip = the first strip's)
    uint32_t strip, n_strips;
    uint32_t logH;        // ring depth per row
    uint32_t logRW;       // column records in the LDS ring (a power of two above the strip's rows + 48)
};
struct ClStripDevice {
    const ClStripDesc* strips;
    const uint4* recs;                // per problem: one record per column (see popoa_strip_kernel)
    unsigned long long* handoff;
    uint32_t* progress;               // zeroed before every pass
    uint32_t debug_fail;              // test hook (CL_STRIP_DEBUG_FAIL=1): every second pair's last strip reports failure, as if a wait had expired
};

#endif
