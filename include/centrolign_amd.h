/*
 * centrolign_amd.h — C ABI of the MI355X-native anchor-and-stitch hot path.
 *
 * This is the drop-in boundary for centrolign's between-anchor alignment path.  centrolign has no
 * FFI layer of its own (it is one C++ library), so every entry point below names the C++ seam of
 * the reference that it replaces (paths relative to the reference repository root):
 *
 *   cl_po_poa_batch          <-> po_poa<NumPW>(graph1, graph2, sources1, sources2, sinks1, sinks2,
 *                                params, score_out)                include/centrolign/alignment.hpp:78-85
 *                                (implementation po_poa_internal<true,NumPW>, alignment.hpp:753-1163),
 *                                applied to a whole batch of independent subproblems.
 *   cl_stitch_batch_align    <-> the loop of Stitcher::subalign calls inside Stitcher::stitch /
 *                                Stitcher::internal_stitch        include/centrolign/stitcher.hpp:157-203,216-231
 *                                i.e. subalign (src/stitcher.cpp:24-78) + do_alignment<NumPW>
 *                                (stitcher.hpp:237-370) + translate (src/alignment.cpp:26-45) for every
 *                                extracted SubGraphInfo pair.
 *   cl_stitch_plan_*         <-> the same, split into prepare / execute / collect so that a caller can keep
 *                                a batch resident in HBM, overlap batches, and time the device part.
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the ABI (negative int = error, text via
 * cl_last_error); all node ids are SUBGRAPH-LOCAL ids exactly as in SubGraphInfo
 * (include/centrolign/subgraph_extraction.hpp:14-33); neighbour lists keep BaseGraph::previous()/next()
 * order because the reference's traceback tie-breaks depend on it (alignment.hpp:1069-1136).
 * Results are bit-identical to the reference: same int32 scores, same AlignedPair sequence.
 */
#ifndef CENTROLIGN_AMD_H
#define CENTROLIGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CL_ABI_VERSION 12   /* bumped whenever a struct layout or an entry point's signature changes; capi.py refuses a library of another version */

/* AlignedPair::gap (src/alignment.cpp:11) */
#define CL_GAP UINT64_MAX

/* error codes */
enum {
    CL_OK = 0,
    CL_ERR_INVALID_ARGUMENT = -1,   /* malformed batch (offsets not monotone, id out of range, ...) */
    CL_ERR_BAD_GAP_PARAMS = -2,     /* src/stitcher.cpp:34-37: gap_open must increase, gap_extend decrease */
    CL_ERR_NO_DEVICE = -3,          /* no HIP device / HIP runtime failure at context creation */
    CL_ERR_HIP = -4,                /* a HIP call failed; see cl_last_error */
    CL_ERR_OUT_OF_MEMORY = -5,
    CL_ERR_UNSUPPORTED_ROUTE = -6,  /* the requested route is not available through this entry point */
    CL_ERR_CYCLIC_GRAPH = -7,       /* a subgraph is not a DAG (topological_order.hpp:56 asserts) */
    CL_ERR_UNREACHABLE_SINK = -8    /* no (source..sink) connection; the reference has UB here (alignment.hpp:76-77) */
};

/* which algorithm a subproblem was routed to; same meaning as the reference's instrument tags
 * (stitcher.hpp:274-358: "pd1","pd2","po","ad1","ad2","w","u") */
enum {
    CL_ROUTE_PO_POA = 0,
    CL_ROUTE_PURE_DELETION_1 = 1,   /* graph2 empty  -> pure_deletion_alignment(graph1) */
    CL_ROUTE_PURE_DELETION_2 = 2,   /* graph1 empty  -> pure_deletion_alignment(graph2) + swap_graphs */
    CL_ROUTE_DELETION_WFA_1 = 3,
    CL_ROUTE_DELETION_WFA_2 = 4,
    CL_ROUTE_PWFA = 5,
    CL_ROUTE_GREEDY_PARTIAL = 6
};

/* AlignmentParameters<3> (alignment.hpp:56-65).  A NumPW<3 alignment uses the leading NumPW
 * (gap_open, gap_extend) pairs, as truncate_parameters does (alignment.hpp:208-219). */
typedef struct cl_align_params {
    uint32_t match;
    uint32_t mismatch;
    uint32_t gap_open[3];
    uint32_t gap_extend[3];
} cl_align_params;

/* public tunables of Stitcher (stitcher.hpp:48-64) that steer subalign/do_alignment */
typedef struct cl_stitch_params {
    cl_align_params alignment_params;
    uint64_t max_trivial_size;
    uint64_t min_wfa_size;
    uint64_t max_wfa_size;
    double   max_wfa_ratio;
    uint64_t wfa_pruning_dist;
    uint64_t deletion_alignment_ratio;
    uint64_t deletion_alignment_short_max_size;
    uint64_t deletion_alignment_long_min_size;
} cl_stitch_params;

/* Fills the values the centrolign CLI runs with (src/parameters.cpp:74-85), NOT the class defaults. */
void cl_stitch_params_default(cl_stitch_params* p);

/*
 * One side (graph1 or graph2) of every subproblem of a batch, concatenated.
 * Problem k owns nodes [node_off[k], node_off[k+1]) of label/prev_off/back_translation; ids stored in
 * prev_idx/next_idx/src_idx/snk_idx are local to the problem (0 .. node count-1).
 * prev_off/next_off index prev_idx/next_idx globally and have node_off[n_problems]+1 entries.
 */
typedef struct cl_graph_side {
    const uint64_t* node_off;          /* [n_problems + 1] */
    const uint8_t*  label;             /* BaseGraph::label(node) */
    const uint64_t* prev_off;          /* CSR of BaseGraph::previous(node), order preserved */
    const uint32_t* prev_idx;
    const uint64_t* next_off;          /* CSR of BaseGraph::next(node); may be NULL (derived from prev) */
    const uint32_t* next_idx;
    const uint64_t* src_off;           /* [n_problems + 1] SubGraphInfo::sources, order preserved */
    const uint32_t* src_idx;
    const uint64_t* snk_off;           /* [n_problems + 1] SubGraphInfo::sinks, order preserved */
    const uint32_t* snk_idx;
    const uint64_t* back_translation;  /* SubGraphInfo::back_translation; NULL => results keep local ids */
} cl_graph_side;

typedef struct cl_stitch_batch {
    uint64_t       n_problems;
    cl_graph_side  side[2];
    const uint8_t* only_deletion_alns; /* [n_problems] 4th argument of Stitcher::subalign; NULL => all 0 */
} cl_stitch_batch;

/* Library-allocated result; release with cl_stitch_result_free.  pairs has the memory layout of
 * centrolign::AlignedPair[] (two uint64_t: node_id1, node_id2; CL_GAP marks a gap). */
typedef struct cl_stitch_result {
    uint64_t  n_problems;
    uint64_t* aln_off;   /* [n_problems + 1], in pairs */
    uint64_t* pairs;     /* [2 * aln_off[n_problems]] */
    int64_t*  score;     /* [n_problems] score_out of the routed aligner (po_poa: best sink-pair M) */
    uint8_t*  route;     /* [n_problems] CL_ROUTE_* */
    uint8_t*  num_pw;    /* [n_problems] NumPW chosen by subalign (src/stitcher.cpp:47-64) */
} cl_stitch_result;

typedef struct cl_context cl_context;
typedef struct cl_stitch_plan cl_stitch_plan;

int         cl_abi_version(void);
int         cl_device_count(void);
/* Binds a HIP device (gfx950), creates the stream all work of this context runs on. NULL on failure
 * (reason via cl_last_error(NULL)). Contexts are independent; one context is not thread-safe. */
cl_context* cl_context_create(int device_ordinal);
void        cl_context_destroy(cl_context* ctx);
const char* cl_last_error(const cl_context* ctx);
const char* cl_device_name(const cl_context* ctx);

/* ---- One merge over several GPUs: merge groups -----------------------------------------------------------------------------------
 * (no counterpart in the single-threaded reference — the seam stays Core::align, include/centrolign/core.hpp:181-252.)
 * The contexts of a merge group — one per process and device, up to CL_PEER_MAX_MEMBERS — are given the SAME two graphs and call the same
 * cl_merge / cl_core_align / cl_anchor_chain.  Each runs the whole call; the far pass of the affine chaining DP (sparse_affine_chain_dp,
 * include/centrolign/anchorer.hpp:2087-2416: the independent (chain of graph 1, chain of graph 2) tree sets) is divided between them by
 * chain combination, and what a member finds goes straight into the other members' device memory (hipIpc-mapped; peer stores over xGMI),
 * ordered by stream memory operations — no host round trip, no collective.  Every member returns the result of the single-context call,
 * bit for bit.  Protocol: each member calls cl_context_peer_export once, the handles are exchanged by whatever transport the caller has
 * (they are plain bytes), then every member calls cl_context_peer_group with the members' handles in the same order, its own index, and an
 * epoch base that is the same on every member and grows from one group call to the next on every context (e.g. 16 x the merge's number in
 * execution order: a merge shares at most a few DPs).  n_members <= 1 leaves the group.  DPs with fewer than 2 or more than 64 chain
 * combinations, sparse (scale-estimate) DPs and short DPs are not shared.  A member that fails inside a shared DP leaves the others
 * waiting: callers run members under a timeout. */
#define CL_PEER_MAX_MEMBERS 8
typedef struct cl_peer_handle { unsigned char bytes[128]; } cl_peer_handle;
typedef struct cl_peer_stats {
    uint64_t shared_dps;            /* chaining DPs whose far pass this context shared */
    uint64_t shared_far_launches;   /* far launches it ran on its share of the combinations */
    uint64_t merged_blocks;         /* macro-blocks whose other combinations came from the other members */
    uint64_t epoch_mark;            /* highest epoch this context's inbox has been used with: cl_context_peer_group refuses a base below it (the
                                       arrival words are never reset); callers that reuse contexts start from the maximum over the members */
    uint64_t selftest_mark;         /* highest token cl_context_peer_selftest has been given: the next one must be greater */
    uint64_t steals;                /* chunks this context pulled through cl_context_peer_steal's shared counter */
} cl_peer_stats;
int cl_context_peer_export(cl_context* ctx, cl_peer_handle* out);
int cl_context_peer_group(cl_context* ctx, uint32_t n_members, uint32_t my_index, const cl_peer_handle* members, uint32_t epoch_base);
int cl_context_peer_stats(const cl_context* ctx, cl_peer_stats* out);
/* Work stealing inside the current group: the next chunk number of job `job` from ONE atomic counter in member 0's exported memory (system-scope
 * atomics of a one-thread kernel: peer atomics over xGMI, no collective, no host exchange).  Every member calls it with the same job number — larger than
 * any job the group has used — until the chunk it gets is >= its number of chunks; chunk lists are the caller's (centrolign_amd/dist.py: the stitch
 * subproblems of a merge in LPT order).  Without a group the count is local.  The reference has no counterpart (stitcher.hpp:157-203 runs the subproblems
 * one after the other). */
int cl_context_peer_steal(cl_context* ctx, uint32_t job, uint32_t* chunk_out);
/* Every member of the current group at the same time, with the same token (greater than any earlier one): stores, arrival words and waits
 * once round the group.  CL_OK when the mechanism works between these devices; on CL_ERR_HIP (nothing came within timeout_ms) destroy the
 * context — its stream may be stuck behind a wait — and run merges on single contexts. */
int cl_context_peer_selftest(cl_context* ctx, uint32_t token, uint32_t timeout_ms);

/* Device memory of a context (the memory model): the bytes its calls hold at the moment, their high-water mark since the
 * context was made (or since reset_peak), what its block cache keeps for the next call, and what the runtime reports for the whole device
 * (hipMemGetInfo).  The reference has no counterpart (it logs the process's resident set, src/utility.cpp: log_memory_usage). */
typedef struct cl_memory_stats {
    uint64_t live_bytes, peak_bytes, cached_bytes;
    uint64_t device_free_bytes, device_total_bytes;
    uint64_t pinned_host_bytes;     /* page-locked host area of the context */
} cl_memory_stats;
int cl_context_memory(cl_context* ctx, cl_memory_stats* out, int reset_peak);

/* Process-wide counts of the second attempts the device path makes by itself (all contexts, since the library was loaded or since reset):
 * a long run is "clean" when both fallback counts are zero.  The reference has no counterpart (its CPU path has no co-residency to lose). */
typedef struct cl_fallback_stats {
    uint64_t strip_fallbacks;   /* pairs whose strips of rows (popoa_strip_kernel) gave up waiting for one another and were run again anti-diagonal-wise */
    uint64_t walk_stalls;       /* chaining DPs whose walk kernel's workgroups were not co-resident in time and that were run again on the per-block kernels */
    uint64_t chain_dps;         /* chaining DPs run on the device */
    uint64_t stitch_plans;      /* stitch plans executed */
    uint64_t strip_pairs;       /* pairs that took the strips at all */
    uint64_t bond_trims_past_the_end;   /* (ABI 12) cl_identify_bonds calls' end trims that stood where the reference's result is UNDEFINED: Bonder::trim_partition_ends
                                         * (src/bonder.cpp:753-757) takes off intervening_segments[interval.second] — one element PAST that vector when the interval ends at
                                         * the last shared segment — and goes on with whatever the heap holds there (its output changes with MALLOC_PERTURB_ and, on some
                                         * inputs, from run to run: profiles/r06_fuzz_msa.json).  This library counts that element as zeros.  A -c run whose text differs
                                         * from a reference run's with this count non-zero is such a case, not a parity loss */
} cl_fallback_stats;
void cl_fallback_counters(cl_fallback_stats* out, int reset);

/* po_poa<NumPW> for every problem of the batch; num_pw[k] in {1,2,3}. */
int cl_po_poa_batch(cl_context* ctx, const cl_stitch_batch* batch, const uint8_t* num_pw,
                    const cl_align_params* params, cl_stitch_result* out);

/* Stitcher::subalign for every problem of the batch (NumPW choice, routing, translate). */
int cl_stitch_batch_align(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* params,
                          cl_stitch_result* out);

void cl_stitch_result_free(cl_stitch_result* r);
/* arrays of a result for n_problems problems and n_pairs aligned pairs in all, for a caller that fills them itself (a stitch hook); freed by cl_stitch_result_free */
int  cl_stitch_result_alloc(cl_stitch_result* out, uint64_t n_problems, uint64_t n_pairs);

/* Stitch hook: where the subproblems of ONE merge are aligned by several devices.  When set on a context, cl_stitch (hence cl_core_align, cl_merge) hands every
 * extracted batch of at least min_cells DP cells to the hook instead of aligning it on this context alone; the hook returns the complete result, in the batch's order
 * — typically every member of a merge group (cl_context_peer_group) runs the same merge, pulls chunks of the LPT-ordered subproblem list from the group's counter
 * (cl_context_peer_steal), aligns them with cl_stitch_batch_align and exchanges the pieces with the other members (centrolign_amd/msa.py does that over the
 * host group).  Subproblems are independent by construction (stitcher.hpp:157-203 only concatenates), so the result is the single-context one bit for bit.
 * fn == NULL removes the hook.  A non-zero return of the hook fails the call. */
typedef int (*cl_stitch_hook_fn)(void* user, cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* params, cl_stitch_result* out);
int  cl_context_set_stitch_hook(cl_context* ctx, cl_stitch_hook_fn fn, void* user, uint64_t min_cells);

/* --- split form: prepare once, execute many times, collect ------------------------------------------- */
/* Validates, routes, topologically orders and packs the batch, and copies it to HBM (synchronous).
 * force_num_pw may be NULL (subalign semantics) or per-problem NumPW with every non-empty problem forced
 * to PO-POA (cl_po_poa_batch semantics). */
int  cl_stitch_plan_create(cl_context* ctx, const cl_stitch_batch* batch, const cl_stitch_params* params,
                           const uint8_t* force_num_pw, cl_stitch_plan** plan_out);
/* Enqueues the DP + traceback kernels for the whole plan on the context's stream; asynchronous. */
int  cl_stitch_plan_execute(cl_context* ctx, cl_stitch_plan* plan);
/* Same work, launched kernel by kernel with HIP events around every launch so that cl_stitch_plan_launch_info can
 * report per-kernel durations (the unprofiled execute replays a captured hipGraph instead). */
int  cl_stitch_plan_execute_profiled(cl_context* ctx, cl_stitch_plan* plan);
/* one concurrent pass (as cl_stitch_plan_execute) with HIP events round every launch on its stream: cl_launch_info.event_ms (ABI 11) */
int  cl_stitch_plan_execute_evented(cl_context* ctx, cl_stitch_plan* plan);
/* Waits for the stream; returns the device time of the LAST execute in ms (HIP events on the context's
 * stream) through *ms_out if not NULL. */
int  cl_stitch_plan_sync(cl_context* ctx, cl_stitch_plan* plan, float* ms_out);
/* Copies alignments back (synchronous) and assembles the result. */
int  cl_stitch_plan_collect(cl_context* ctx, cl_stitch_plan* plan, cl_stitch_result* out);
void cl_stitch_plan_destroy(cl_context* ctx, cl_stitch_plan* plan);

/* plan statistics, for measurement */
typedef struct cl_plan_stats {
    uint64_t n_problems;
    uint64_t n_po_poa;           /* problems routed to the device PO-POA kernels */
    uint64_t dp_cells;           /* sum (n1+1)*(n2+1) over those */
    uint64_t dp_bytes;           /* sum (n1+1)*(n2+1)*sizeof(cell_t<NumPW>) = 4*(1+2*NumPW) B per cell */
    uint64_t n_linear;           /* PO-POA problems whose two graphs are simple chains */
    uint64_t max_cells;          /* largest single matrix */
    uint64_t workspace_bytes;    /* HBM bytes held by the plan */
    uint64_t n_launches;         /* kernel launches per execute */
    uint64_t n_strip_fallbacks;  /* pairs whose strips (popoa_strip_kernel) gave up waiting for one another and were run again by the anti-diagonal kernel, summed over
                                    the plan's collects (0 unless the device is crowded beyond what the strips' bounded waits tolerate) */
} cl_plan_stats;
int cl_stitch_plan_stats(const cl_stitch_plan* plan, cl_plan_stats* stats_out);

/* one kernel launch of a plan (a group of subproblems that share a kernel variant) */
typedef struct cl_launch_info {
    char     kernel[64];         /* e.g. "popoa_linear<npw=2,rows=4,waves=1>" */
    uint64_t n_problems;
    uint64_t dp_cells;
    uint64_t dp_bytes;           /* algorithmic bytes: cells * sizeof(cell_t<NumPW>) */
    float    last_ms;            /* duration of this launch ALONE on the device in the last cl_stitch_plan_execute_profiled, by the kernel's own clock
                                    (first workgroup's start to last workgroup's end, s_memrealtime); valid after cl_stitch_plan_sync */
    float    in_pass_ms;         /* the same clock in the last cl_stitch_plan_execute, where the plan's launches run side by side (0: none yet) */
    uint32_t lds_bytes;          /* dynamic LDS per workgroup (0: static only) */
    uint32_t max_sweep;          /* the longest dependent chain of the launch: max n1 + n2 over its subproblems */
    uint32_t max_n1, max_n2;     /* the subproblem that has it */
    float    event_ms;           /* HIP events round this launch on its stream in the last cl_stitch_plan_execute_evented (launches side by side as in
                                    cl_stitch_plan_execute): from the stream reaching the launch to its completion — agrees with the rocprofv3 kernel trace of a
                                    step up to the stream's launch gap; 0: none yet.  (ABI 11) */
} cl_launch_info;
int cl_stitch_plan_launch_count(const cl_stitch_plan* plan);
int cl_stitch_plan_launch_info(cl_context* ctx, const cl_stitch_plan* plan, int index, cl_launch_info* info_out);

/* The routes of Stitcher::do_alignment (stitcher.hpp:268-360) that are host algorithms in the reference are host algorithms
 * here: pure deletion, greedy_partial_alignment (alignment.hpp:1212-1611), deletion_wfa_po_poa (:2036-2282), pwfa_po_poa
 * (:2299-2338); cl_stitch_batch_align / the plan API run them at plan creation next to the device problems.  This entry runs
 * ONE subproblem of a batch by its route without a device (CL_ERR_UNSUPPORTED_ROUTE if the route is PO-POA, the device's):
 * pairs_out is malloc'ed, AlignedPair layout, translated through back_translation; release with free(). */
int cl_host_route_align(const cl_stitch_batch* batch, uint64_t problem, const cl_stitch_params* params, int* route_out,
                        uint64_t** pairs_out, uint64_t* n_pairs_out);

/* The topological order the DEVICE ranks the nodes of one subgraph by (no counterpart in the reference, whose DP runs in the order of
 * topological_order.hpp:12-60 — any topological order gives the same DP values, and the traceback's ties follow the previous() and sink
 * lists, which are kept): mode 0 = the packer's choice between the reference's order (1) and the order by level — longest path from a
 * source, ties in the reference's order (2) —, whichever makes fewer rows read more than four ranks back, then the shorter longest read.
 * order_out[rank] = local node id (n entries); the two counts describe the order returned.  Host only, no device needed. */
int cl_stitch_rank_order(const cl_stitch_batch* batch, uint64_t problem, int side, int mode, uint32_t* order_out,
                         uint32_t* far_reads_out /* may be NULL */, uint32_t* longest_read_out /* may be NULL */);

/* --- Stitcher::stitch proper: from a partitioned anchor chain to the stitched base-level alignment ----------------
 * (include/centrolign/stitcher.hpp:34-38,104-206).  The two merge graphs are passed as flat views of BaseGraph
 * (include/centrolign/graph.hpp:96-151) + SentinelTableau (include/centrolign/modify_graph.hpp:33-38). */
typedef struct cl_base_graph {
    uint64_t        n_nodes;
    const uint8_t*  label;       /* [n_nodes] */
    const uint64_t* next_off;    /* [n_nodes+1] CSR of BaseGraph::next, order preserved */
    const uint32_t* next_idx;
    const uint64_t* prev_off;    /* [n_nodes+1] CSR of BaseGraph::previous, order preserved */
    const uint32_t* prev_idx;
    uint64_t        n_paths;     /* BaseGraph::path_size() */
    const uint64_t* path_off;    /* [n_paths+1] */
    const uint32_t* path_nodes;  /* BaseGraph::path(p) concatenated */
    uint64_t        src_id;      /* SentinelTableau::src_id */
    uint64_t        snk_id;      /* SentinelTableau::snk_id */
} cl_base_graph;

/* std::vector<std::vector<anchor_t>>: anchors seg_off[s] .. seg_off[s+1]-1 form segment s; anchor a holds the node
 * pairs walk1/walk2[walk_off[a] .. walk_off[a+1]) (anchor_t::walk1/walk2, include/centrolign/anchorer.hpp:36-57) */
typedef struct cl_anchor_segments {
    uint64_t        n_segments;
    const uint64_t* seg_off;
    const uint64_t* walk_off;
    const uint32_t* walk1;
    const uint32_t* walk2;
} cl_anchor_segments;

typedef struct cl_owned_batch cl_owned_batch;
/* Extractor::extract_graphs_between (include/centrolign/anchorer.hpp:494-585, subgraph_extraction.hpp:52-125) with
 * PathMerge reachability (path_merge.hpp:96-277), flattened in the order Stitcher::stitch consumes the subgraph pairs.
 * Host only (no device needed). */
int  cl_extract_stitch_batch(const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_anchor_segments* segments,
                             cl_owned_batch** batch_out);
const cl_stitch_batch* cl_owned_batch_view(cl_owned_batch* batch);
void cl_owned_batch_free(cl_owned_batch* batch);

typedef struct cl_alignment {
    uint64_t  n_pairs;
    uint64_t* pairs;   /* AlignedPair[n_pairs] */
} cl_alignment;
/* Stitcher::stitch: extraction + every subalign on the device + anchors copied in between. */
int  cl_stitch(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_anchor_segments* segments,
               const cl_stitch_params* params, cl_alignment* out);
/* Stitcher::internal_stitch (include/centrolign/stitcher.hpp:41-43,209-234), the second entry of seam S2: the alignment of a graph with
 * ITSELF along a chain of anchors (anchor a = walk1/walk2[walk_off[a] .. walk_off[a + 1]), node ids of `graph`), the gaps between consecutive
 * anchors aligned as in cl_stitch; nothing in front of the first anchor or behind the last.  Output order as in the reference: anchor 0,
 * then every later anchor's pairs followed by the gap in front of it. */
int cl_internal_stitch(cl_context* ctx, const cl_base_graph* graph, uint64_t n_anchors, const uint64_t* walk_off, const uint32_t* walk1,
                       const uint32_t* walk2, const cl_stitch_params* params, cl_alignment* out);
void cl_alignment_free(cl_alignment* a);

/* --- Stitcher::despecify_indel_breakpoints (src/stitcher.cpp:265-310) -------------------------------------------------
 * Drops the weak anchors that pin the breakpoints of long indels.  Inputs are the anchor_t fields the algorithm reads
 * (score, gap_before, gap_score_before, gap_after, gap_score_after), as parallel arrays in chain order; Stitcher's
 * tunables min_indel_fuzz_length / indel_fuzz_score_proportion (stitcher.hpp:68-71).  On return keep_out[i] tells
 * whether anchor i survives; the gap arrays hold, in their first *n_kept_out entries, the updated values of the kept
 * anchors in order (exactly what the reference leaves in the resized vector). Host only. */
int cl_despecify_indel_breakpoints(uint64_t n_anchors, const double* score, int64_t* gap_before, double* gap_score_before,
                                   int64_t* gap_after, double* gap_score_after, int64_t min_indel_fuzz_length,
                                   double indel_fuzz_score_proportion, uint8_t* keep_out, uint64_t* n_kept_out);

/* --- Anchorer chaining DP (include/centrolign/anchorer.hpp:1812-2547) --------------------------------------------------
 * The seam is sparse_affine_chain_dp itself: (graphs + embedded paths, match sets, gap parameters, local scale) in,
 * the optimal chain out, as the reference's anchor_chain dispatch calls it (anchorer.hpp:1213-1307): sources / sinks are
 * the graph ends (cl_chain_params.global_anchoring, the CLI default) or absent (local chaining); no masked matches. */
/* std::vector<match_set_t> (include/centrolign/match_finder.hpp:21-34), flattened.  Set s owns walks
 * set_off1[s] .. set_off1[s+1]-1 of graph 1 (walk w = nodes1[walk_off1[w] .. walk_off1[w+1])), likewise for graph 2. */
typedef struct cl_match_sets {
    uint64_t        n_sets;
    const uint64_t* set_off1;
    const uint64_t* walk_off1;
    const uint32_t* nodes1;
    const uint64_t* set_off2;
    const uint64_t* walk_off2;
    const uint32_t* nodes2;
    const uint64_t* count1;       /* match_set_t::count1 */
    const uint64_t* count2;
    const uint64_t* full_length;
} cl_match_sets;

/* Anchorer::gap_open / gap_extend (anchorer.hpp:152-175) and the ScoreFunction fields (score_function.hpp:27-45) */
typedef struct cl_chain_params {
    double gap_open[3];
    double gap_extend[3];
    int    anchor_score_function;   /* ScoreFunction::AnchorScore */
    double pair_count_power;
    double length_intercept;
    double length_decay_power;
    int    global_anchoring;        /* Anchorer::global_anchoring (CLI default true, src/parameters.cpp:60): chains start at
                                       the nodes after graph.src_id and end at the nodes before graph.snk_id
                                       (anchorer.hpp:1069-1076); 0 = local chaining (sources/sinks == nullptr) */
} cl_chain_params;
/* the values the CLI runs with (src/parameters.cpp:39-59) */
void cl_chain_params_default(cl_chain_params* p);

typedef struct cl_chain_result {
    uint64_t  n_anchors;
    uint32_t* anchors;     /* [3 * n_anchors]: anchor_t::match_set, idx1, idx2 in chain order */
    uint64_t  n_pairs;     /* number of match pairs that took part */
    float*    dp;          /* [n_pairs] final DP value of every pair in (set, idx1, idx2) order, or NULL */
    uint64_t  n_ties;      /* traceback steps where several predecessors attained the maximum (resolved as the
                              reference's search trees resolve them) */
    float     device_ms;   /* HIP-event time of the DP kernels */
    float     prep_ms;     /* host: coordinates, ordering, packing, upload */
    float     index_ms;    /* value index (device sort + download) */
    float     traceback_ms;/* host: optimum + traceback with tie resolution */
    /* global anchoring, affine DP: anchor_t::gap_before / gap_score_before of the first anchor and gap_after /
       gap_score_after of the last one (anchorer.hpp:2445-2451, 2461-2467); 0 otherwise */
    int64_t   gap_before_first, gap_after_last;
    double    gap_score_before_first, gap_score_after_last;
} cl_chain_result;

/* sparse_affine_chain_dp<..., float, ...> over the leading num_match_sets sets. */
int  cl_chain_sparse_affine(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2,
                            const cl_match_sets* matches, uint64_t num_match_sets, const cl_chain_params* params,
                            double local_scale, int want_dp, cl_chain_result* out);
/* sparse_chain_dp<..., float, ...> (anchorer.hpp:1511-1750): the chaining without gap costs that
 * Anchorer::estimate_score_scale (anchorer.hpp:998-1047) and the leaf calibration (src/core.cpp:122-175) run. */
int  cl_chain_sparse(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
                     uint64_t num_match_sets, const cl_chain_params* params, int want_dp, cl_chain_result* out);
/* exhaustive_chain_dp (anchorer.hpp:1342-1509) + AnchorGraph::heaviest_weight_path (src/anchorer.cpp:68-133): the O(M^2) chaining behind
 * the CLI's "-g 0" and the reference's test oracle, as Anchorer::anchor_chain dispatches it (:1229-1232: no edge scores).  Host only
 * (ctx may be NULL); the anchor graph is never materialised, so memory stays O(M).  cl_chain_params.global_anchoring as above. */
int  cl_chain_exhaustive(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
                         uint64_t num_match_sets, const cl_chain_params* params, cl_chain_result* out);
void cl_chain_result_free(cl_chain_result* r);

/* --- Partitioner::partition_anchors (include/centrolign/partitioner.hpp:72-213) -----------------------------------------
 * Cuts the anchor chain into the well-anchored segments that Stitcher::stitch takes (Core::align, core.hpp:237-249).
 * Host only: extract_graphs_between + min source-sink distances for the gap lengths, then the partition DP
 * (window_average_constrained_partition :353-684 by default; also maximum_weight_partition :215-270 and
 * average_constrained_partition :272-351).  Anchors are passed as the fields the algorithm reads. */
typedef struct cl_partition_params {
    int    constraint_method;        /* Partitioner::ConstraintMethod: 0 Null, 1 Unconstrained, 2 MinAverage, 3 MinWindowAverage (default) */
    double minimum_segment_score;    /* 15000 */
    double minimum_segment_average;  /* 0.1 */
    double window_length;            /* 10000 */
    double generalized_length_mean;  /* -0.5 */
    double boundary_score_factor;    /* 0.95 */
    double score_scale;              /* ScoreFunction::score_scale */
    int    score_boundaries;         /* Core::align passes !is_main_execution */
    int    use_annotated_score;      /* partition_anchors' last argument (false in Core::align) */
    cl_chain_params score_function;  /* the ScoreFunction fields (anchor_weight) */
} cl_partition_params;
void cl_partition_params_default(cl_partition_params* p);
typedef struct cl_anchor_fields {
    uint64_t        n_anchors;
    const uint64_t* walk_off;     /* [n+1] */
    const uint32_t* walk1;
    const uint32_t* walk2;
    const uint64_t* count1;
    const uint64_t* count2;
    const uint64_t* full_length;
    const uint64_t* match_set;
    const double*   score;        /* anchor_t::score, read when use_annotated_score */
} cl_anchor_fields;
/* segments_out: malloc'ed [2 * *n_segments_out] = (first anchor, past-the-last anchor) of every segment, in order;
 * anchors outside every interval are dropped, as in the reference.  Release with free(). */
int cl_partition_anchors(const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_anchor_fields* anchors,
                         const cl_partition_params* params, uint64_t** segments_out, uint64_t* n_segments_out);

/* --- Anchorer::split_branching_matches (include/centrolign/anchorer.hpp:800-956), the first step of anchor_chain when
 * split_matches_at_branchpoints is set (CLI default): match sets whose walks cross the boundary of a superbubble with a
 * large length spread (superbubbles.hpp:63-170, structure_distances.hpp:55-185) near their ends are cut there; the pieces
 * are appended to the vector as new sets.  Host only, no masks.  The result owns its arrays. */
typedef struct cl_split_params {
    uint64_t anchor_split_limit;        /* Anchorer::anchor_split_limit (5) */
    uint64_t min_split_length;          /* Anchorer::min_split_length (128) */
    uint64_t min_path_length_spread;    /* Anchorer::min_path_length_spread (50) */
    uint64_t max_split_match_set_size;  /* Anchorer::max_split_match_set_size (16) */
} cl_split_params;
void cl_split_params_default(cl_split_params* p);
typedef struct cl_owned_match_sets cl_owned_match_sets;
int  cl_split_branching_matches(const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
                                const cl_split_params* params, cl_owned_match_sets** out);
void cl_owned_match_sets_view(const cl_owned_match_sets* sets, cl_match_sets* view_out);
void cl_owned_match_sets_free(cl_owned_match_sets* sets);

/* --- PathMatchFinder::find_matches (include/centrolign/match_finder.hpp:73-85,120-212) ----------------
 * The minimal rare matches between the embedded paths of two merge graphs, as Core calls it for every merge
 * (include/centrolign/core.hpp:289): the index the reference builds with PathESA (path_esa.hpp:81-170: SA-IS suffix array,
 * Kasai LCP) comes from the device (suffix array by prefix doubling on radix sorts, LCP by a descent over the kept rank
 * levels); the LCP-interval tree, Hui's distinct-start-node counts (src/esa.cpp:149-300), the query
 * (esa.hpp:284-494) and the walk-out (esa.hpp:610-665) are one linear host pass each.  The sentinel characters are the labels
 * of graph.src_id / graph.snk_id (SentinelTableau::src_sentinel / snk_sentinel); labels must be < 255.  The result holds
 * the match sets in the reference's order, count1 / count2 / full_length set as match_finder.hpp:199-203 does. */
typedef struct cl_match_params {
    uint64_t        max_count;            /* BaseMatchFinder::max_count: count1 * count2 bound (CLI 3000, src/parameters.cpp:36) */
    int             use_color_set_size;   /* BaseMatchFinder::use_color_set_size: the reference's two counting structures
                                             (esa.hpp:208-279) return the same counts; accepted for interface parity */
    cl_chain_params score;                /* ScoreFunction of the positive-weight filter (match_finder.hpp:162); only the four
                                             ScoreFunction fields are read */
} cl_match_params;
void cl_match_params_default(cl_match_params* p);
typedef struct cl_match_stats {
    uint64_t text_length;        /* joined path text incl. sentinels */
    uint32_t doubling_rounds;    /* device suffix-sort rounds */
    uint64_t n_internal_nodes;   /* LCP intervals */
    uint64_t n_candidates;       /* (parent, internal child) pairs examined */
    float    sa_ms, lcp_ms;      /* device time */
    double   tree_ms, query_ms, walk_ms;   /* host passes */
    double   text_ms, suffix_wall_ms;      /* joining the path text; the device half as the host sees it (transfers included) */
} cl_match_stats;
int cl_find_matches(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_params* params,
                    cl_owned_match_sets** out, cl_match_stats* stats /* may be NULL */);
/* The two halves separately.  cl_match_joined_text: PathESA's joined_seq (path_esa.hpp:92-118; malloc'ed, release with free()).
 * cl_suffix_array_lcp: the device half on any text that ends in a unique smallest character (sa / lcp / isa: n entries
 * each, lcp[0] = 0).  cl_matches_from_suffix_array: the host half for a caller that already holds the suffix array and LCP
 * array of cl_match_joined_text's text (no device needed). */
int cl_match_joined_text(const cl_base_graph* graph1, const cl_base_graph* graph2, uint8_t** text_out, uint64_t* n_out);
int cl_suffix_array_lcp(cl_context* ctx, const uint8_t* text, uint64_t n, uint32_t* sa, uint32_t* lcp, uint32_t* isa, uint32_t* rounds_out);
int cl_matches_from_suffix_array(const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_params* params, const uint32_t* sa,
                                 const uint32_t* lcp, uint64_t n, cl_owned_match_sets** out, cl_match_stats* stats /* may be NULL */);

/* --- Anchorer::anchor_chain (include/centrolign/anchorer.hpp:135-145, 958-1329), no masks, after
 * cl_split_branching_matches (the caller runs it first, as anchor_chain does at :971-973): budgeted match selection (which REORDERS the caller's match sets, :1108-1173),
 * scale estimation (:998-1047), the affine chain, gap and score annotation (:2443-2468), and — with
 * do_fill_in_anchoring — fill_in_anchor_chain (:619-699): every gap of the chain is re-anchored with the matches that lie
 * inside it (divvy_matches :701-798, assign_reanchor_budget / merge_fill_in_chains src/anchorer.cpp:136-222); the DPs of
 * all gaps run as ONE batched device pass. */
typedef struct cl_anchor_params {
    cl_chain_params chain;
    uint64_t max_num_match_pairs;        /* Anchorer::max_num_match_pairs (CLI: 1250000, src/parameters.cpp:39) */
    double   score_scale;                /* ScoreFunction::score_scale (calibrated per input, src/core.cpp:193) */
    int      autocalibrate_gap_penalties;/* Anchorer::autocalibrate_gap_penalties */
    int      do_fill_in_anchoring;       /* Anchorer::do_fill_in_anchoring (CLI default true) */
    int      chaining_algorithm_plus_one;/* 0: the default, SparseAffine.  Otherwise Anchorer::ChainAlgorithm + 1 (the CLI's hidden -g, src/main.cpp:128-129):
                                            2 = Sparse: no scale estimate, sparse_chain_dp for the chain and its fill-in, and — in cl_core_align / cl_merge —
                                            ChainMerge tables (include/centrolign/chain_merge.hpp:100-225: every node on ONE chain) instead of PathMerge, as
                                            Core::execute does (core.hpp:350-357); 3 = SparseAffine; 1 = Exhaustive is cl_chain_exhaustive, not offered here */
} cl_anchor_params;

typedef struct cl_anchor_chain_result {
    uint64_t  n_anchors;
    uint64_t* anchors;           /* [3*n]: anchor_t::match_set (index into the REORDERED sets), idx1, idx2 */
    int64_t*  gap_before;        /* anchor_t::gap_before ... */
    int64_t*  gap_after;
    double*   gap_score_before;
    double*   gap_score_after;
    double*   score;             /* anchor_t::score */
    uint64_t* count1;            /* anchor_t::count1, count2, full_length */
    uint64_t* count2;
    uint64_t* full_length;
    uint64_t* walk_off;          /* [n+1]: anchor a's walk1 / walk2 = walk1/walk2[walk_off[a] .. walk_off[a+1]) (parent node ids) */
    uint32_t* walk1;
    uint32_t* walk2;
    uint64_t  n_sets;
    uint64_t* set_order;         /* [n_sets]: the caller's vector after the call holds original set set_order[k] at position k */
    double    scale;             /* the estimated score scale passed to the affine DP */
    uint64_t  n_ties;
    uint64_t  fill_in_pairs;     /* match pairs chained by the fill-in passes (both the scale estimate's and the final one) */
    float     fill_in_device_ms;
    float     dp_device_ms;      /* device time of the two whole-graph DPs (sparse_chain_dp for the scale, sparse_affine_chain_dp) */
    double    dp_pair_evals;     /* (predecessor record, query) evaluations an all-pairs sweep of those two DPs amounts to:
                                    sum over chain combinations of records x match pairs / 2 */
    uint64_t  dp_match_pairs;    /* match pairs of the affine DP */
    uint32_t  dp_combinations;   /* (chain of graph 1, chain of graph 2) combinations that hold records in the affine DP */
} cl_anchor_chain_result;

int  cl_anchor_chain(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
                     const cl_anchor_params* params, cl_anchor_chain_result* out);
void cl_anchor_chain_result_free(cl_anchor_chain_result* r);

/* --- anchor_chain with masked matches and a given scale: the last two arguments of Anchorer::anchor_chain
 * (include/centrolign/anchorer.hpp:143-145), as the tandem-duplication rounds of cyclisation call it (src/core.cpp:221-227).
 * masked: [3 * n_masked] (set, idx1, idx2) in the indexing of `matches`; a masked pair takes no part in any DP (MatchBank skips it,
 * match_bank.hpp:187-215,252-268; fill-in translates the mask to the divvied sets, anchorer.hpp:662-680) while the forward-edge masks
 * still see every walk (:1753-1774).  override_scale != NULL: the affine DP's scale, no estimate (:975-978).
 * The sets are not moved here (set_order in the result): the mask stays in the caller's indexing, where the reference re-indexes it
 * in place (:1159-1166).  Branch splitting with a mask (:816-820,911-918) is not part of this entry: the leaf graphs it is called on
 * in the reference have no branches. */
int  cl_anchor_chain_masked(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
                            const cl_anchor_params* params, const uint64_t* masked, uint64_t n_masked, const double* override_scale,
                            cl_anchor_chain_result* out);
/* Core::generate_diagonal_mask (src/core.cpp:301-321) and Core::update_mask (:323-372) on flat sets; *masked_out is malloc'ed
 * [3 * *n_masked_out] (release with free()), sorted, without duplicates.  chain_walk1/2: the node pairs of the chain's anchors, concatenated. */
int  cl_generate_diagonal_mask(const cl_match_sets* matches, uint64_t** masked_out, uint64_t* n_masked_out);
int  cl_update_mask(const cl_match_sets* matches, uint64_t n_chain_pairs, const uint32_t* chain_walk1, const uint32_t* chain_walk2,
                    int mask_reciprocal, const uint64_t* masked, uint64_t n_masked, uint64_t** masked_out, uint64_t* n_masked_out);

/* --- Core::align (include/centrolign/core.hpp:181-252): matches of one merge -> the merge's base-level alignment ------
 * anchor_chain (split + chaining + fill-in, device), partition_anchors (host), despecify_indel_breakpoints per
 * segment (host), Stitcher::stitch (device).  score_boundaries = !is_main_execution goes in partition.score_boundaries;
 * partition.score_scale and partition.score_function are taken from `anchor`. */
typedef struct cl_core_align_params {
    int                 split_matches_at_branchpoints;   /* Anchorer::split_matches_at_branchpoints (CLI default true) */
    cl_split_params     split;
    cl_anchor_params    anchor;
    cl_partition_params partition;
    int64_t             min_indel_fuzz_length;           /* Stitcher tunables, stitcher.hpp:68-71 */
    double              indel_fuzz_score_proportion;
    cl_stitch_params    stitch;
} cl_core_align_params;
void cl_core_align_params_default(cl_core_align_params* p);   /* the CLI's values (src/parameters.cpp:36-92) */
typedef struct cl_core_align_result {
    cl_alignment alignment;      /* the stitched Alignment (AlignedPair layout) */
    uint64_t  n_segments;        /* the partitioned, despecified anchor chain that was stitched */
    uint64_t* seg_off;
    uint64_t* walk_off;
    uint32_t* walk1;
    uint32_t* walk2;
    double    scale;             /* estimated score scale of the chaining */
    uint64_t  n_chain_anchors;   /* anchors before partitioning */
    float     chain_ms, partition_ms, stitch_ms;   /* wall time of the three stages */
    float     chain_device_ms;   /* cl_anchor_chain_result.dp_device_ms / dp_pair_evals / dp_match_pairs / dp_combinations */
    double    chain_pair_evals;
    uint64_t  chain_match_pairs;
    uint32_t  chain_combinations;
    /* The reference's memory-restrained dispatch for this merge (memory_restraint_size = 2^30, src/parameters.cpp:40). Informational: the
     * restrained variants (PackedMatchBank / PackedForwardEdges, PackedPathMerge) hold the same tables in fewer bits and select the same chain
     * (packed_match_bank.hpp:150-165,236-249), and the device formulation has no search trees to restrain, so nothing here switches on them. */
    uint32_t  ref_restrain_memory;     /* paths1 * paths2 * max_num_match_pairs * log2(max_num_match_pairs) > 2^30 (core.hpp:194) */
    uint32_t  ref_packed_path_merge;   /* nodes1 * paths1 + nodes2 + paths2 > 2^30 (core.hpp:306-310, as written there) */
    uint32_t  ref_path_merge_widths;   /* bytes of (UIntSize, UIntChain) the reference instantiates: 0x0401, 0x0402 or 0x0802 (core.hpp:318-340) */
} cl_core_align_result;
int  cl_core_align(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
              const cl_core_align_params* params, cl_core_align_result* out);
void cl_core_align_result_free(cl_core_align_result* r);

/* --- fuse (include/centrolign/fuse.hpp:46-152) and the whole loop body of Core::do_execution (core.hpp:268-392) --------------
 * cl_fuse merges `source` into `dest` along their alignment (AlignedPair[n_pairs], graph-1 ids = dest) exactly as the
 * reference does in place: matched nodes with equal labels are identified, the sentinels joined, the other source nodes
 * appended in id order; then the substitution edges, the source edges that are missing, and the source paths after the
 * destination's.  Node ids and the order of every adjacency list are the reference's.  The result's sentinel ids are dest's
 * (core.hpp:388).  Host only. */
typedef struct cl_owned_base_graph cl_owned_base_graph;
int  cl_fuse(const cl_base_graph* dest, const cl_base_graph* source, const uint64_t* pairs, uint64_t n_pairs, cl_owned_base_graph** out);
/* internal_fuse(graph, alignments, …) (include/centrolign/fuse.hpp:144-247), the merge Core::apply_bonds makes along the tandem-duplication
 * alignments of a cyclised run (src/core.cpp:631-636): nodes of ONE graph paired by the alignments (pairs: the alignments concatenated,
 * AlignedPair layout; node ids of `graph`) are merged transitively, one new node per label of a group.  The result may contain cycles.
 * trans_out (may be NULL): [graph->n_nodes] old node -> new node, which also translates an alignment (fuse.hpp:229-243) and the sentinels. */
int  cl_internal_fuse(const cl_base_graph* graph, const uint64_t* pairs, uint64_t n_pairs, cl_owned_base_graph** out, uint64_t* trans_out);
void cl_owned_base_graph_view(const cl_owned_base_graph* graph, cl_base_graph* view_out);
void cl_owned_base_graph_free(cl_owned_base_graph* graph);

/* The data formats either side of a merge (host only; text is malloc'ed and NUL-terminated, release with free()).
 * cl_leaf_graph: make_base_graph + add_sentinels(graph, 5, 6) (src/modify_graph.cpp:30-77, src/execution.cpp:66-73) of one sequence
 * (ACGTN in either case -> 0..4, anything else -> 5, src/utility.cpp:324-345): nodes 0..n-1 in a chain, the source sentinel
 * n, the sink n+1, one path.  cl_explicit_cigar: explicit_cigar(alignment, graph1, graph2)
 * (include/centrolign/alignment.hpp:2804-2843), the pairwise output of the CLI.  cl_write_gfa: write_gfa(graph, tableau, out,
 * decode) (include/centrolign/gfa.hpp:46-157), the MSA output; path_names[p] = BaseGraph::path_name(p). */
int cl_leaf_graph(const char* sequence, uint64_t n, cl_owned_base_graph** out);
int cl_explicit_cigar(const cl_base_graph* graph1, const cl_base_graph* graph2, const uint64_t* pairs, uint64_t n_pairs, char** text_out,
                      uint64_t* len_out /* may be NULL */);
int cl_write_gfa(const cl_base_graph* graph, const char* const* path_names, int decode, char** text_out, uint64_t* len_out /* may be NULL */);
/* The -A output of the CLI on an acyclic result (src/core.cpp:546-550): explicit_cigar(induced_pairwise_alignment(graph, path1, path2), seq1, seq2)
 * (src/alignment.cpp:84-229) — the pairwise alignment of two of the input sequences that the MSA graph implies.  CL_ERR_CYCLIC_GRAPH when path1
 * visits a node twice (the reference throws). */
int cl_induced_pairwise_cigar(const cl_base_graph* graph, uint64_t path1, uint64_t path2, char** text_out, uint64_t* len_out /* may be NULL */);
/* -S / -R (src/core.cpp:370-422, src/execution.cpp:222-277): a finished subproblem is written as PREFIX_<hash>.gfa with cl_write_gfa, a
 * restart loads it back with read_gfa(in) + add_sentinels(graph, 5, 6) — NOT the graph that was written: node ids follow the S lines and the
 * sentinels come last, and the run continues on that graph.  cl_read_gfa builds exactly that graph (add_sentinels = 0: read_gfa alone);
 * *path_names_out is a malloc'ed array of malloc'ed strings.  cl_subproblem_hash_hex: Execution::subproblem_hash of the subproblem's leaf
 * names, printed as to_hex does (16 upper-case digits + NUL). */
int cl_read_gfa(const char* text, uint64_t len, int add_sentinels, cl_owned_base_graph** out, char*** path_names_out /* may be NULL */,
                uint64_t* n_paths_out /* may be NULL */);
int cl_subproblem_hash_hex(const char* const* sequence_names, uint64_t n, char* hex_out /* [17] */);

/* Calibration (Core::calibrate_anchor_scores_and_identify_bonds without cyclisation, src/core.cpp:98-191).
 * cl_estimate_score_scale: Anchorer::estimate_score_scale (include/centrolign/anchorer.hpp:998-1047): the sparse anchor chain,
 * its weight over its length plus the shortest fill-in between its anchors.  cl_leaf_intrinsic_scale: the per-leaf step
 * (src/core.cpp:122-166): the leaf's matches against itself (second copy under sentinels 7 / 8), the main-diagonal
 * subset, estimate_score_scale; ScoreFunction::score_scale is the mean of the leaves' values (:168-184). */
int cl_estimate_score_scale(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_match_sets* matches,
                            const cl_anchor_params* params, double* scale_out);
int cl_leaf_intrinsic_scale(cl_context* ctx, const cl_base_graph* leaf, const cl_match_params* match_params,
                            const cl_anchor_params* anchor_params, double* scale_out);

/* One merge of the progressive MSA from nothing but the two subproblem graphs: reassign_sentinels (5,6 / 7,8),
 * PathMatchFinder::find_matches, Core::align, fuse.  `fused` is the next subproblem's graph. */
typedef struct cl_merge_params {
    cl_match_params      match;
    cl_core_align_params align;
} cl_merge_params;
void cl_merge_params_default(cl_merge_params* p);
typedef struct cl_merge_result {
    cl_core_align_result align;       /* align.alignment = next_problem.alignment; the stitched anchor segments and stage times */
    cl_owned_base_graph* fused;       /* next_problem.graph (+ tableau ids) */
    uint64_t             n_match_sets;
    float                match_ms, align_ms, fuse_ms;
} cl_merge_result;
int  cl_merge(cl_context* ctx, const cl_base_graph* graph1, const cl_base_graph* graph2, const cl_merge_params* params, cl_merge_result* out);
void cl_merge_result_free(cl_merge_result* r);

/* --- Cyclisation (the CLI's -c): the steps around the hot path's calls -----------------------------------------------
 * Bonder (include/centrolign/bonder.hpp:47-108) as Core::calibrate_anchor_scores_and_identify_bonds uses it (src/core.cpp:229-234): on a
 * leaf against itself, bond algorithm LongestNearOptDevConstrained (the class default; the CLI has no switch for the other two). */
typedef struct cl_bond_params {          /* src/parameters.cpp:91-97 -> :174-180 */
    double min_opt_proportion;           /* tandem_dup_score_proportion, 0.2 */
    int    include_gap_scores;           /* include_tandem_dup_gap_scores, true */
    double min_length;                   /* min_cyclizing_length, 100000 */
    double deviation_drift_factor;       /* 150 */
    double separation_drift_factor;      /* 50 */
    double deduplication_slosh_proportion; /* 0.1 */
    double trim_window_proportion;       /* 0.1 */
} cl_bond_params;
void cl_bond_params_default(cl_bond_params* p);
/* a chain of anchors as Bonder reads it: anchor a = walk1/walk2[walk_off[a] .. walk_off[a+1]) (node ids), anchor_t::score, gap_after, gap_score_after */
typedef struct cl_chain_anchors {
    uint64_t n;
    const uint64_t* walk_off;
    const uint32_t* walk1;
    const uint32_t* walk2;
    const double*   score;
    const int64_t*  gap_after;
    const double*   gap_score_after;
} cl_chain_anchors;
/* std::vector<bond_interval_t> (bonder.hpp:22-42) of one leaf: interval i = bonds [interval_off[i], interval_off[i+1]); a bond pairs
 * path positions [offset1, offset1 + length) with [offset2, offset2 + length) */
typedef struct cl_bonds {
    uint64_t  n_intervals;
    uint64_t* interval_off;
    uint64_t* offset1;
    uint64_t* offset2;
    uint64_t* length;
    double*   score;
} cl_bonds;
/* Bonder::identify_bonds (bonder.hpp:116-452; the optimal chain = the leaf's main-diagonal chain of the calibration, the secondary chain =
 * the next-best chain with the diagonal masked) and, with deduplicate != 0, Bonder::deduplicate_self_bonds (src/bonder.cpp:473-551). Host only. */
int  cl_identify_bonds(const cl_base_graph* leaf, const cl_chain_anchors* opt_chain, const cl_chain_anchors* secondary_chain,
                       const cl_bond_params* params, int deduplicate, cl_bonds* out);
void cl_bonds_free(cl_bonds* b);

/* InconsistencyIdentifier (include/centrolign/inconsistency_identifier.hpp:17-57) as the CLI configures it (src/parameters.cpp:98-103 -> :182-187) */
typedef struct cl_polish_params {
    uint64_t max_tight_cycle_size;               /* max_realignment_cycle_size, 10000 */
    uint64_t max_bond_inconsistency_window;      /* inconsistent_indel_window, 100 */
    uint64_t min_inconsistency_disjoint_length;  /* 8 */
    uint64_t min_inconsistency_total_length;     /* 50 */
    uint64_t padding_target_min_length;          /* realignment_min_padding, 1000 */
    uint64_t padding_max_length_limit;           /* realignment_max_padding, 10000 */
} cl_polish_params;
void cl_polish_params_default(cl_polish_params* p);
/* InconsistencyIdentifier::identify_inconsistencies (inconsistency_identifier.hpp:66-187) on a cyclised graph: *bounds_out = malloc'ed
 * [2 * *n_out] (first node, last node) of the mutually disjoint regions to realign, in the reference's order. Host only. */
int  cl_identify_inconsistencies(const cl_base_graph* graph, const cl_polish_params* params, uint64_t** bounds_out, uint64_t* n_out);

/* Core::polish_cyclized_graph (src/core.cpp:650-767): every region cl_identify_inconsistencies reports is realigned from scratch — the stretches
 * of the paths through it as sequences of their own, the guide tree (newick, NULL = in order; sequence_names = the FASTA's names) expanded by
 * their copies (make_copy_expanded_tree, :769-976), match sets induced from the whole graph's matches against itself (InducedMatchFinder),
 * Core::align + fuse per tree node on the device — and put back in place (integrate_polished_subgraphs, :978-1069).  path_names: the names of
 * graph's paths; params->align.anchor.score_scale = the calibrated scale.  *n_regions_out (may be NULL): regions realigned. */
int  cl_polish_cyclized_graph(cl_context* ctx, const cl_base_graph* graph, const char* const* path_names, const char* newick,
                              const char* const* sequence_names, uint64_t n_sequences, const cl_merge_params* params, const cl_polish_params* polish,
                              cl_owned_base_graph** out, uint64_t* n_regions_out);

/* The per-leaf step of the calibration (src/core.cpp:122-175) that also keeps what the tandem-duplication rounds read: the leaf's matches
 * against itself and the main-diagonal chain the scale was estimated on (:168-172).  *memo_out (may be NULL: then this is
 * cl_leaf_intrinsic_scale) is released with cl_leaf_calibration_free. */
typedef struct cl_leaf_calibration cl_leaf_calibration;
int  cl_leaf_calibrate(cl_context* ctx, const cl_base_graph* leaf, const cl_match_params* match_params, const cl_anchor_params* anchor_params,
                       double* scale_out, cl_leaf_calibration** memo_out);
void cl_leaf_calibration_free(cl_leaf_calibration* c);
typedef struct cl_alignment_list {
    uint64_t      n;
    cl_alignment* alignments;
} cl_alignment_list;
void cl_alignment_list_free(cl_alignment_list* l);
/* The tandem-duplication rounds of one leaf (src/core.cpp:199-296) after every leaf has been calibrated (anchor_params->score_scale = the mean
 * of the intrinsic scales, :193): per round Anchorer::anchor_chain with the mask and the leaf's own scale, Bonder::identify_bonds +
 * deduplicate_self_bonds, Core::bonds_to_chain + Stitcher::internal_stitch per bond, Core::update_mask; at most max_rounds
 * (max_tandem_duplication_search_rounds, 3), ending with the first round without bonds.  Alignments in PATH POSITIONS of the leaf (:275-283),
 * in the order the reference appends them. */
int  cl_leaf_bond_alignments(cl_context* ctx, const cl_base_graph* leaf, const cl_leaf_calibration* memo, const cl_anchor_params* anchor_params,
                             const cl_stitch_params* stitch_params, const cl_bond_params* bond_params, uint64_t max_rounds, cl_alignment_list* out);
/* simplify_bubbles (src/modify_graph.cpp:165-382) with purge_uncovered_nodes (:89-163): bubbles whose alleles are plain runs of nodes get their
 * identical alleles merged (every path moves to the first of them), nodes no path visits any more are dropped. Host only; cyclic graphs welcome. */
int  cl_simplify_bubbles(const cl_base_graph* graph, cl_owned_base_graph** out);
/* Core::apply_bonds up to the polishing step (src/core.cpp:613-645): alignment a pairs path positions of path path_of_alignment[a] of `root`;
 * positions -> node ids, internal_fuse along all alignments, simplify_bubbles. */
int  cl_apply_bonds(const cl_base_graph* root, uint64_t n_alignments, const uint64_t* path_of_alignment, const cl_alignment* alignments,
                    cl_owned_base_graph** out);

/* --- the front of the driver: FASTA, guide tree, the whole MSA (the rest of it) -------------------------------
 * cl_parse_fasta: parse_fasta (src/utility.cpp:19-65): a name is the header line up to the first space; sequence lines are joined;
 *   the same complaints (no name line, unequal or growing line lengths, empty input).
 * cl_msa_plan_create: Tree(newick) (src/tree.cpp:39-160; NULL or "" = in_order_newick_string of the names, :17-37) followed by what
 *   Execution's constructor does with it (src/execution.cpp:12-92): prune to the given sequence names, compact, binarize,
 *   small_first_postorder.  Out: the leaves in the order the reference calibrates them (Execution::leaf_subproblems: tree-id order)
 *   as indices into `names`, and the merges in execution order.  A merge's operands are SLOTS: slot i < n_leaves is leaf i of that
 *   order, slot n_leaves + k is the result of merge k; merge_children[2k] is graph 1 of merge k (the node's first child).
 * cl_msa: main() from the parsed inputs on (src/main.cpp:239-301): plan, leaf graphs, calibration (score_scale = mean intrinsic
 *   scale in leaf order), cl_merge per node, then explicit_cigar + a line end for two sequences (src/main.cpp:295) / write_gfa otherwise: the text is
 *   byte for byte what the CLI writes to its standard output.  text_out is malloc'ed. */
typedef struct cl_fasta {
    uint64_t            n_sequences;
    const char* const*  names;       /* NUL-terminated */
    const char* const*  sequences;
    const uint64_t*     lengths;
    void*               owner;
} cl_fasta;
int  cl_parse_fasta(cl_context* ctx /* may be NULL */, const char* text, uint64_t len, cl_fasta* out);
void cl_fasta_free(cl_fasta* f);

typedef struct cl_msa_plan {
    uint64_t  n_leaves;
    uint64_t* leaf_sequence;     /* [n_leaves] index into the names given */
    uint64_t  n_merges;
    uint64_t* merge_children;    /* [2 * n_merges] slots of graph 1 and graph 2 */
} cl_msa_plan;
int  cl_msa_plan_create(cl_context* ctx /* may be NULL */, const char* newick, const char* const* names, uint64_t n_names, cl_msa_plan* out);
void cl_msa_plan_free(cl_msa_plan* p);

typedef struct cl_msa_params {
    cl_merge_params merge;            /* merge.align.anchor.score_scale is overwritten by the calibration unless it is skipped */
    int             skip_calibration; /* --skip-calibration of the CLI: merge.align.anchor.score_scale stays what the caller set (cl_msa_params_default: 0.303092, the CLI's own start value) */
    int             n_workers;        /* contexts (threads) that run leaf calibrations and independent merges side by side; <= 1: one */
    const int*      devices;          /* where the worker contexts sit: worker w on device ordinal devices[w % n_devices] (worker 0 is ctx itself and stays on
                                         ctx's device); NULL / n_devices 0: all on ctx's device.  One process, several GPUs: graphs are host arrays at this
                                         boundary, so nothing travels between devices but the calls */
    int             n_devices;
    const char*     subproblems_prefix;      /* -S: every finished subproblem is written as PREFIX_<hash>.gfa, one line each in PREFIX_info.txt
                                                (Core::emit_subproblem, src/core.cpp:397-422); NULL: off */
    int             restart;                 /* -R: subproblems whose files exist are loaded instead of computed (Execution::restart,
                                                src/execution.cpp:222-277); needs subproblems_prefix */
    const char*     induced_pairwise_prefix; /* -A: PREFIX_<name1>_<name2>.txt with the induced pairwise CIGAR of every pair of sequences
                                                (Core::output_pairwise_alignments, src/core.cpp:523-575); NULL: off */
    int             cyclize;                 /* -c: tandem duplications found in every sequence (cl_leaf_bond_alignments) are merged into cycles of
                                                the final graph (cl_apply_bonds) and the graph is polished (cl_polish_cyclized_graph): src/core.cpp:63-94.
                                                With subproblems_prefix the bond alignments go to PREFIX_bonds.txt (Core::emit_restart_bonds, :476-489)
                                                and a restart reads them back instead of searching again (Core::restart_bonds, :491-521) */
    uint64_t        max_tandem_duplication_search_rounds;   /* 3 */
    cl_bond_params   bonds;
    cl_polish_params polish;
} cl_msa_params;
void cl_msa_params_default(cl_msa_params* p);
typedef struct cl_msa_stats {
    uint64_t n_merges, root_nodes;
    double   score_scale, calibration_s, match_s, align_s, fuse_s, total_s;
    uint64_t n_restarted;        /* subproblems loaded from files (-R) */
    uint64_t n_bonds;            /* tandem duplications merged (-c) */
    uint64_t n_polished_regions; /* regions realigned by the polishing step (-c) */
    double   bonds_s, cyclize_s; /* time in the tandem-duplication rounds; in apply_bonds + polishing */
} cl_msa_stats;
int  cl_msa(cl_context* ctx, const char* fasta_text, uint64_t fasta_len, const char* newick /* NULL: in-order tree */,
            const cl_msa_params* params, char** text_out, uint64_t* len_out, cl_msa_stats* stats /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* CENTROLIGN_AMD_H */
