// cl_align_api.cpp — Core::align (include/centrolign/core.hpp:181-252) on top of the seams of this library:
//   anchorer.anchor_chain            -> cl_split_branching_matches + cl_anchor_chain   (device: both chaining DPs, fill-in)
//   partitioner.partition_anchors    -> cl_partition_anchors                           (host)
//   stitcher.despecify_indel_breakpoints per segment -> cl_despecify_indel_breakpoints (host)
//   stitcher.stitch                  -> cl_stitch                                      (device: every between-anchor DP)
// Pure composition: no arithmetic of its own.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <thread>

#include <cmath>
#include "cl_internal.hpp"
#include "stitch_host.hpp"

thread_local ClSharedTables cl_tls_tables;

extern "C" {

void cl_core_align_params_default(cl_core_align_params* p) {
    memset(p, 0, sizeof(*p));
    p->split_matches_at_branchpoints = 1;
    cl_split_params_default(&p->split);
    cl_chain_params_default(&p->anchor.chain);
    p->anchor.max_num_match_pairs = 1250000;
    p->anchor.score_scale = 1.0;
    p->anchor.autocalibrate_gap_penalties = 1;
    p->anchor.do_fill_in_anchoring = 1;
    cl_partition_params_default(&p->partition);
    p->min_indel_fuzz_length = 50;
    p->indel_fuzz_score_proportion = 0.001;
    cl_stitch_params_default(&p->stitch);
}

void cl_core_align_result_free(cl_core_align_result* r) {
    if (!r) return;
    cl_alignment_free(&r->alignment);
    free(r->seg_off); free(r->walk_off); free(r->walk1); free(r->walk2);
    memset(r, 0, sizeof(*r));
}

int cl_core_align(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* matches, const cl_core_align_params* ap,
             cl_core_align_result* out) {
    return cl_core_align_prepared(ctx, g1, g2, matches, ap, out, nullptr);
}

}  // extern "C"

ClPathMergeTables::ClPathMergeTables() : x1(new clhost::PathMergeTable()), x2(new clhost::PathMergeTable()) {}
ClPathMergeTables::~ClPathMergeTables() { wait(); delete x1; delete x2; }
void ClPathMergeTables::start(const cl_base_graph* g1, const cl_base_graph* g2, bool chain_merge) {
    // chain_merge: ChainMerge tables (every node on one chain) instead of PathMerge, as Core::execute builds them when the chaining algorithm is not
    // SparseAffine (core.hpp:350-357)
    auto one = [chain_merge](clhost::PathMergeTable* x, const cl_base_graph* g) { return chain_merge ? x->build_chain_merge(*g) : x->build(*g); };
    // (small graphs — the thousands of realignments of a polishing step — are done on the spot: a thread and a pool hand-off cost more than the tables)
    if ((g1->n_nodes + 1) * (g1->n_paths + 1) + (g2->n_nodes + 1) * (g2->n_paths + 1) < (1u << 18)) { ok1 = one(x1, g1); ok2 = one(x2, g2); return; }
    builder = std::thread([this, g1, g2, one] { cl_pool_run(2, [&](unsigned t) { if (t) ok2 = one(x2, g2); else ok1 = one(x1, g1); }); });
}
void ClPathMergeTables::wait() { if (builder.joinable()) builder.join(); }

// cl_core_align; `ready`: the two PathMerge tables a caller started building earlier (cl_merge: beside the match finding)
int cl_core_align_prepared(cl_context* ctx, const cl_base_graph* g1, const cl_base_graph* g2, const cl_match_sets* matches, const cl_core_align_params* ap,
                           cl_core_align_result* out, ClPathMergeTables* ready) {
    cl_bind_device(ctx);
    if (!ctx || !g1 || !g2 || !matches || !ap || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    // the two PathMerge tables, built side by side and shared by every stage below (cl_internal.hpp: cl_shared_table)
    ClPathMergeTables own;
    ClPathMergeTables& tabs = ready ? *ready : own;
    const bool timing = getenv("CL_CHAIN_TIMING") != nullptr;
    const auto t_tab = std::chrono::steady_clock::now();
    if (!ready) own.start(g1, g2, ap->anchor.chaining_algorithm_plus_one == 2);
    tabs.wait();
    if (!tabs.ok1 || !tabs.ok2) { cl_set_error(ctx, "graph is not acyclic"); return CL_ERR_CYCLIC_GRAPH; }
    const clhost::PathMergeTable &x1 = *tabs.x1, &x2 = *tabs.x2;
    struct Registered {
        ClSharedTables saved;
        Registered(const cl_base_graph* a, const clhost::PathMergeTable* xa, const cl_base_graph* b, const clhost::PathMergeTable* xb) : saved(cl_tls_tables) {
            cl_tls_tables.g[0] = a; cl_tls_tables.x[0] = xa; cl_tls_tables.g[1] = b; cl_tls_tables.x[1] = xb;
        }
        ~Registered() { cl_tls_tables = saved; }
    } registered(g1, &x1, g2, &x2);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return (float)std::chrono::duration<double, std::milli>(now() - t).count(); };
    int rc;
    auto t0 = now();
    if (timing) fprintf(stderr, "[cl_core_align] path-merge tables          %8.1f ms\n", ms_since(t_tab));
    // anchor chain (core.hpp:194-197)
    cl_owned_match_sets* split = nullptr;
    cl_match_sets view = *matches;
    if (ap->split_matches_at_branchpoints) {
        if ((rc = cl_split_branching_matches_unless_identity(g1, g2, matches, &ap->split, &split))) { cl_set_error(ctx, "split_branching_matches failed"); return rc; }
        if (split) cl_owned_match_sets_view(split, &view);
    }
    if (timing) fprintf(stderr, "[cl_core_align] branch split               %8.1f ms\n", ms_since(t0));
    cl_anchor_chain_result ch;
    rc = cl_anchor_chain(ctx, g1, g2, &view, &ap->anchor, &ch);
    cl_owned_match_sets_free(split);
    if (rc) return rc;
    out->scale = ch.scale;
    out->n_chain_anchors = ch.n_anchors;
    out->chain_device_ms = ch.dp_device_ms;
    out->chain_pair_evals = ch.dp_pair_evals;
    out->chain_match_pairs = ch.dp_match_pairs;
    out->chain_combinations = ch.dp_combinations;
    {   // what the reference would dispatch to (core.hpp:194, 303-343); see the header
        const double restraint = (double)(1u << 30);
        const double mp = (double)ap->anchor.max_num_match_pairs;
        out->ref_restrain_memory = (double)g1->n_paths * (double)g2->n_paths * mp * log2(mp) > restraint ? 1u : 0u;
        out->ref_packed_path_merge = (double)g1->n_nodes * (double)g1->n_paths + (double)g2->n_nodes + (double)g2->n_paths > restraint ? 1u : 0u;
        const uint64_t max_nodes = std::max(g1->n_nodes, g2->n_nodes), max_paths = std::max(g1->n_paths, g2->n_paths);
        const bool small_nodes = max_nodes < 0xFFFFFFFFull;
        if (small_nodes && max_paths < 0xFF) out->ref_path_merge_widths = 0x0401;
        else if (small_nodes && (out->ref_packed_path_merge ? max_paths < 0xFFFF : true)) out->ref_path_merge_widths = 0x0402;
        else out->ref_path_merge_widths = 0x0802;
    }
    out->chain_ms = ms_since(t0);
    t0 = now();
    // partition (core.hpp:237-241)
    cl_partition_params pp = ap->partition;
    pp.score_scale = ap->anchor.score_scale;
    pp.score_function = ap->anchor.chain;
    std::vector<uint64_t> match_set(ch.n_anchors);
    for (uint64_t i = 0; i < ch.n_anchors; ++i) match_set[i] = ch.anchors[3 * i];
    cl_anchor_fields af{ch.n_anchors, ch.walk_off, ch.walk1, ch.walk2, ch.count1, ch.count2, ch.full_length, match_set.data(), ch.score};
    uint64_t* segs = nullptr;
    uint64_t n_segs = 0;
    if ((rc = cl_partition_anchors(g1, g2, &af, &pp, &segs, &n_segs))) { cl_anchor_chain_result_free(&ch); cl_set_error(ctx, "partition_anchors failed"); return rc; }
    // despecify_indel_breakpoints per segment (core.hpp:245-247), then flatten the kept anchors for the stitcher
    std::vector<uint64_t> seg_off{0}, walk_off{0};
    std::vector<uint32_t> w1, w2;
    for (uint64_t sgi = 0; sgi < n_segs && !rc; ++sgi) {
        const uint64_t a = segs[2 * sgi], b = segs[2 * sgi + 1], n = b - a;
        std::vector<int64_t> gb(ch.gap_before + a, ch.gap_before + b), ga(ch.gap_after + a, ch.gap_after + b);
        std::vector<double> gsb(ch.gap_score_before + a, ch.gap_score_before + b), gsa(ch.gap_score_after + a, ch.gap_score_after + b);
        std::vector<uint8_t> keep(n ? n : 1, 1);
        uint64_t kept = 0;
        rc = cl_despecify_indel_breakpoints(n, ch.score + a, gb.data(), gsb.data(), ga.data(), gsa.data(), ap->min_indel_fuzz_length,
                                            ap->indel_fuzz_score_proportion, keep.data(), &kept);
        if (rc) break;
        for (uint64_t i = 0; i < n; ++i) {
            if (!keep[i]) continue;
            w1.insert(w1.end(), ch.walk1 + ch.walk_off[a + i], ch.walk1 + ch.walk_off[a + i + 1]);
            w2.insert(w2.end(), ch.walk2 + ch.walk_off[a + i], ch.walk2 + ch.walk_off[a + i + 1]);
            walk_off.push_back(w1.size());
        }
        seg_off.push_back(walk_off.size() - 1);
    }
    free(segs);
    cl_anchor_chain_result_free(&ch);
    if (rc) { cl_set_error(ctx, "despecify_indel_breakpoints failed"); return rc; }
    out->partition_ms = ms_since(t0);
    t0 = now();
    // stitch (core.hpp:249-251)
    cl_anchor_segments sg{seg_off.size() - 1, seg_off.data(), walk_off.data(), w1.data(), w2.data()};
    if ((rc = cl_stitch(ctx, g1, g2, &sg, &ap->stitch, &out->alignment))) return rc;
    out->stitch_ms = ms_since(t0);
    out->n_segments = seg_off.size() - 1;
    out->seg_off = (uint64_t*)malloc(seg_off.size() * sizeof(uint64_t));
    out->walk_off = (uint64_t*)malloc(walk_off.size() * sizeof(uint64_t));
    out->walk1 = (uint32_t*)malloc((w1.size() ? w1.size() : 1) * sizeof(uint32_t));
    out->walk2 = (uint32_t*)malloc((w2.size() ? w2.size() : 1) * sizeof(uint32_t));
    if (!out->seg_off || !out->walk_off || !out->walk1 || !out->walk2) { cl_core_align_result_free(out); return CL_ERR_OUT_OF_MEMORY; }
    memcpy(out->seg_off, seg_off.data(), seg_off.size() * sizeof(uint64_t));
    memcpy(out->walk_off, walk_off.data(), walk_off.size() * sizeof(uint64_t));
    if (!w1.empty()) { memcpy(out->walk1, w1.data(), w1.size() * sizeof(uint32_t)); memcpy(out->walk2, w2.data(), w2.size() * sizeof(uint32_t)); }
    return CL_OK;
}
