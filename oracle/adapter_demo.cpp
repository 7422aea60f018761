/*
 * adapter_demo.cpp — TEST INFRASTRUCTURE ONLY.  Drop-in demonstration on real reference objects: runs the
 * reference pipeline (match finding, chaining, partitioning, extraction — all the UNMODIFIED reference, linked
 * from oracle/_ref) on a FASTA pair / small MSA and, for every merge, stitches twice:
 *   (1) the reference's own Stitcher::subalign loop on the CPU        (include/centrolign/stitcher.hpp:157-203)
 *   (2) include/centrolign_amd/stitch_adapter.hpp -> C ABI -> MI355X
 * and compares the two stitched Alignments element by element.  The patch INTEGRATION.md proposes for
 * Stitcher::stitch is exactly path (2).  Built only where /root/reference exists; the binary travels to the
 * GPU box inside oracle/_ref/.
 *
 * With a fourth argument "core" the comparison is made one level up, at Core::align (include/centrolign/core.hpp:181-252):
 *   (1) the reference's anchor_chain -> partition_anchors -> despecify_indel_breakpoints -> stitch on the CPU
 *   (2) include/centrolign_amd/core_adapter.hpp -> cl_core_align -> MI355X, from the same match sets
 *
 * With "seams" the wrappers of include/centrolign_amd/seam_wrappers.hpp — the reference's own signatures at S1 / S3 / internal_stitch — are
 * called next to the reference's functions on the same objects: anchor_chain per merge (chain, annotation, the order it leaves the match
 * sets in), po_poa<1|2|3> on extracted subgraph pairs (alignment + score), and on every leaf the masked anchor_chain of the cyclisation
 * rounds (src/core.cpp:221-227) + Stitcher::internal_stitch of its chain.
 *
 * usage: adapter_demo <fasta> [newick-file|-] [max_num_match_pairs] [core|seams]
 */
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "centrolign/core.hpp"
#include "centrolign/gfa.hpp"
#include "centrolign/parameters.hpp"
#include "centrolign/stitcher.hpp"
#include "centrolign/utility.hpp"

#include "../include/centrolign_amd/stitch_adapter.hpp"
#include "../include/centrolign_amd/core_adapter.hpp"
#include "../include/centrolign_amd/seam_wrappers.hpp"
#include "../include/centrolign_amd/core_facade.hpp"

using namespace centrolign;

namespace {

struct OpenStitcher : public Stitcher {
    using Stitcher::subalign;
    using Extractor::extract_graphs_between;
};

struct DemoCore : public Core {
    DemoCore(std::vector<std::pair<std::string, std::string>>&& seqs, Tree&& tree) : Core(std::move(seqs), std::move(tree)) {}
    centrolign_amd::Device* dev = nullptr;
    size_t merges = 0, mismatched = 0, problems = 0;
    double t_cpu = 0, t_gpu = 0;

    template <class XMerge>
    Alignment align_both(std::vector<match_set_t>& matches, const Subproblem& sp1, const Subproblem& sp2, XMerge& x1, XMerge& x2) {
        using clk = std::chrono::steady_clock;
        bool restrain = (sp1.graph.path_size() * sp2.graph.path_size() * anchorer.max_num_match_pairs * log2(anchorer.max_num_match_pairs) > memory_restraint_size);
        auto anchors = anchorer.anchor_chain(matches, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, restrain);
        auto segments = partitioner.partition_anchors(anchors, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, false);
        for (auto& seg : segments) stitcher.despecify_indel_breakpoints(seg);
        OpenStitcher st;
        static_cast<Stitcher&>(st) = stitcher;
        std::vector<std::vector<std::pair<SubGraphInfo, SubGraphInfo>>> within;
        std::vector<std::pair<SubGraphInfo, SubGraphInfo>> between;
        std::tie(within, between) = st.extract_graphs_between(segments, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2);

        // (1) reference loop, stitcher.hpp:157-203
        auto a0 = clk::now();
        Alignment ref;
        for (size_t i = 0; i < between.size(); ++i) {
            if (i != 0) {
                const auto& seg = segments[i - 1];
                for (size_t j = 0; j < seg.size(); ++j) {
                    if (j != 0) st.subalign(within[i - 1][j - 1].first, within[i - 1][j - 1].second, ref, false);
                    for (size_t k = 0; k < seg[j].walk1.size(); ++k) ref.emplace_back(seg[j].walk1[k], seg[j].walk2[k]);
                }
            }
            st.subalign(between[i].first, between[i].second, ref, true);
        }
        t_cpu += std::chrono::duration<double>(clk::now() - a0).count();

        // (2) the same loop with the subalign calls batched onto the GPU
        auto a1 = clk::now();
        centrolign_amd::StitchBatchBuilder batch;
        for (size_t i = 0; i < between.size(); ++i) {
            if (i != 0)
                for (size_t j = 1; j < segments[i - 1].size(); ++j) batch.add(within[i - 1][j - 1].first, within[i - 1][j - 1].second, false);
            batch.add(between[i].first, between[i].second, true);
        }
        auto alns = dev->subalign_all<AlignedPair>(batch, centrolign_amd::stitch_params_of(stitcher));
        Alignment got;
        size_t k = 0;
        for (size_t i = 0; i < between.size(); ++i) {
            if (i != 0) {
                const auto& seg = segments[i - 1];
                for (size_t j = 0; j < seg.size(); ++j) {
                    if (j != 0) { got.insert(got.end(), alns[k].begin(), alns[k].end()); ++k; }
                    for (size_t w = 0; w < seg[j].walk1.size(); ++w) got.emplace_back(seg[j].walk1[w], seg[j].walk2[w]);
                }
            }
            got.insert(got.end(), alns[k].begin(), alns[k].end());
            ++k;
        }
        t_gpu += std::chrono::duration<double>(clk::now() - a1).count();
        problems += batch.size();
        bool same = got.size() == ref.size();
        for (size_t i = 0; same && i < got.size(); ++i) same = got[i] == ref[i];
        if (!same) ++mismatched;
        printf("merge %zu: %zu subproblems, stitched length %zu, GPU path %s the reference\n", merges, batch.size(), ref.size(), same ? "==" : "!=");
        ++merges;
        return got;  // continue the MSA on the GPU result: later merges then depend on it
    }

    bool seams = false;

    static bool same_chain(const std::vector<anchor_t>& a, const std::vector<anchor_t>& b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i].walk1 != b[i].walk1 || a[i].walk2 != b[i].walk2 || a[i].count1 != b[i].count1 || a[i].count2 != b[i].count2 ||
                a[i].full_length != b[i].full_length || a[i].score != b[i].score || a[i].gap_before != b[i].gap_before ||
                a[i].gap_after != b[i].gap_after || a[i].gap_score_before != b[i].gap_score_before || a[i].gap_score_after != b[i].gap_score_after ||
                a[i].match_set != b[i].match_set || a[i].idx1 != b[i].idx1 || a[i].idx2 != b[i].idx2)
                return false;
        return true;
    }
    static bool same_sets(const std::vector<match_set_t>& a, const std::vector<match_set_t>& b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i].walks1 != b[i].walks1 || a[i].walks2 != b[i].walks2 || a[i].count1 != b[i].count1 || a[i].count2 != b[i].count2) return false;
        return true;
    }
    template <int NumPW>
    bool po_poa_both(const SubGraphInfo& i1, const SubGraphInfo& i2) {
        AlignmentParameters<NumPW> p;
        p.match = stitcher.alignment_params.match;
        p.mismatch = stitcher.alignment_params.mismatch;
        for (int k = 0; k < NumPW; ++k) { p.gap_open[k] = stitcher.alignment_params.gap_open[k]; p.gap_extend[k] = stitcher.alignment_params.gap_extend[k]; }
        int64_t s_ref = 0, s_got = 0;
        Alignment ref = po_poa(i1.subgraph, i2.subgraph, i1.sources, i2.sources, i1.sinks, i2.sinks, p, &s_ref);
        Alignment got = centrolign_amd::po_poa<NumPW, Alignment>(*dev, i1.subgraph, i2.subgraph, i1.sources, i2.sources, i1.sinks, i2.sinks, p, &s_got);
        return ref == got && s_ref == s_got;
    }

    // seam S3 (+ S1 on the gaps of its chain) on the objects of one merge
    template <class XMerge>
    void seams_of_merge(const std::vector<match_set_t>& matches, const Subproblem& sp1, const Subproblem& sp2, XMerge& x1, XMerge& x2) {
        std::vector<match_set_t> m_ref = matches, m_got = matches;
        auto ref = anchorer.anchor_chain(m_ref, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, false);
        auto got = centrolign_amd::anchor_chain<anchor_t>(*dev, anchorer, score_function, m_got, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, false);
        const bool ok3 = same_chain(ref, got) && same_sets(m_ref, m_got);
        if (!ok3) ++mismatched;
        OpenStitcher st;
        static_cast<Stitcher&>(st) = stitcher;
        auto gaps = st.extract_graphs_between(ref, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2);
        size_t tried = 0, bad = 0;
        for (size_t i = 0; i < gaps.size() && tried < 60; ++i) {
            if (gaps[i].first.subgraph.node_size() == 0 || gaps[i].second.subgraph.node_size() == 0) continue;
            const int npw = 1 + (int)(tried % 3);
            const bool ok = npw == 1 ? po_poa_both<1>(gaps[i].first, gaps[i].second) : npw == 2 ? po_poa_both<2>(gaps[i].first, gaps[i].second) : po_poa_both<3>(gaps[i].first, gaps[i].second);
            ++tried;
            if (!ok) ++bad;
        }
        if (bad) ++mismatched;
        printf("merge %zu: anchor_chain wrapper %s the reference (%zu anchors, %zu sets after the call); po_poa<1|2|3> wrapper %s on %zu gap pairs\n", merges,
               ok3 ? "==" : "!=", ref.size(), m_ref.size(), bad ? "!=" : "==", tried);
        // the same call with masked matches and an overriding scale on graphs WITH branch points: where the splitting cuts match sets, the
        // mask has to follow the pieces (anchorer.hpp:816-820, 911-918)
        {
            std::unordered_set<std::tuple<size_t, size_t, size_t>> k_ref, k_got;
            for (size_t s = 0; s < matches.size(); s += 7) k_ref.emplace(s, 0, 0);
            k_got = k_ref;
            double sc_ref = 0.6, sc_got = 0.6;
            std::vector<match_set_t> m2r = matches, m2g = matches;
            Anchorer eager = anchorer;            // (thresholds under which the small bubbles of these test graphs already count as branch points)
            eager.min_path_length_spread = 1;
            eager.min_split_length = 2;
            auto r2 = eager.anchor_chain(m2r, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, false, &k_ref, &sc_ref);
            auto g2 = centrolign_amd::anchor_chain<anchor_t>(*dev, eager, score_function, m2g, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, x1, x2, false, &k_got, &sc_got);
            const bool ok = same_chain(r2, g2) && same_sets(m2r, m2g) && k_ref == k_got;
            if (!ok) ++mismatched;
            printf("merge %zu: masked + split anchor_chain wrapper %s the reference (%zu anchors, %zu -> %zu sets, mask %zu)\n", merges, ok ? "==" : "!=", r2.size(),
                   matches.size(), m2r.size(), k_ref.size());
        }
    }

    // the cyclisation round of one leaf (src/core.cpp:204-269): masked anchor_chain with the given scale, internal_stitch of its chain
    void seams_of_leaf(const Subproblem& leaf) {
        Subproblem sp = leaf;
        reassign_sentinels(sp.graph, sp.tableau, 5, 6);
        SentinelTableau dummy = sp.tableau;
        dummy.src_sentinel = 7;
        dummy.snk_sentinel = 8;
        auto matches = path_match_finder.find_matches(sp.graph, sp.graph, sp.tableau, dummy);
        PathMerge<> pm(sp.graph, sp.tableau);
        auto mask_ref = generate_diagonal_mask(matches);
        auto mask_got = mask_ref;
        double scale_ref = score_function.score_scale, scale_got = scale_ref;
        std::vector<match_set_t> m_ref = matches, m_got = matches;
        auto ref = anchorer.anchor_chain(m_ref, sp.graph, sp.graph, sp.tableau, sp.tableau, pm, pm, false, &mask_ref, &scale_ref);
        auto got = centrolign_amd::anchor_chain<anchor_t>(*dev, anchorer, score_function, m_got, sp.graph, sp.graph, sp.tableau, sp.tableau, pm, pm, false,
                                                          &mask_got, &scale_got);
        const bool ok3 = same_chain(ref, got) && same_sets(m_ref, m_got) && mask_ref == mask_got;
        if (!ok3) ++mismatched;
        std::vector<anchor_t> head(ref.begin(), ref.begin() + std::min<size_t>(ref.size(), 50));
        Alignment s_ref = stitcher.internal_stitch(head, sp.graph, pm);
        Alignment s_got = centrolign_amd::internal_stitch<Alignment>(*dev, stitcher, head, sp.graph, sp.tableau, pm);
        const bool ok2 = s_ref == s_got;
        if (!ok2) ++mismatched;
        printf("leaf %s: masked anchor_chain wrapper %s the reference (%zu anchors, mask %zu); internal_stitch wrapper %s (%zu anchors, %zu pairs)\n",
               sp.name.c_str(), ok3 ? "==" : "!=", ref.size(), mask_ref.size(), ok2 ? "==" : "!=", head.size(), s_ref.size());
    }

    bool whole_align = false;
    double t_match_cpu = 0, t_match_gpu = 0;

    // Core::align both ways from the same matches (anchor_chain reorders and extends its argument, so each side gets a copy)
    template <class XMerge>
    Alignment align_core_both(std::vector<match_set_t>& matches, const Subproblem& sp1, const Subproblem& sp2, XMerge& x1, XMerge& x2) {
        using clk = std::chrono::steady_clock;
        auto a1 = clk::now();
        const cl_core_align_params prm = centrolign_amd::core_align_params_of(*this, true);
        Alignment got = centrolign_amd::core_align<AlignedPair>(*dev, sp1.graph, sp1.tableau, sp2.graph, sp2.tableau, matches, prm);
        t_gpu += std::chrono::duration<double>(clk::now() - a1).count();
        auto a0 = clk::now();
        std::vector<match_set_t> copy = matches;
        Alignment ref = align(copy, sp1, sp2, x1, x2, true);   // the reference's Core::align (the merge structures are consumed)
        t_cpu += std::chrono::duration<double>(clk::now() - a0).count();
        bool same = got.size() == ref.size();
        for (size_t i = 0; same && i < got.size(); ++i) same = got[i] == ref[i];
        if (!same) ++mismatched;
        printf("merge %zu: %zu match sets, alignment length %zu, GPU Core::align %s the reference\n", merges, matches.size(), ref.size(), same ? "==" : "!=");
        ++merges;
        return got;
    }

    void leaves_first() {
        if (!skip_calibration) calibrate_anchor_scores_and_identify_bonds();
        skip_calibration = true;
        for (auto* leaf : main_execution.leaf_subproblems()) seams_of_leaf(*leaf);
    }

    void run() {
        if (!skip_calibration) calibrate_anchor_scores_and_identify_bonds();
        while (!main_execution.finished()) {
            auto ptrs = main_execution.next();
            auto& next_problem = *std::get<0>(ptrs);
            auto& sp1 = *std::get<1>(ptrs);
            auto& sp2 = *std::get<2>(ptrs);
            reassign_sentinels(sp1.graph, sp1.tableau, 5, 6);
            reassign_sentinels(sp2.graph, sp2.tableau, 7, 8);
            using clk = std::chrono::steady_clock;
            auto m0 = clk::now();
            auto matches = path_match_finder.find_matches(sp1.graph, sp2.graph, sp1.tableau, sp2.tableau);
            t_match_cpu += std::chrono::duration<double>(clk::now() - m0).count();
            if (whole_align) {   // PathMatchFinder::find_matches through the library, from the same reference objects
                auto m1 = clk::now();
                auto ours = centrolign_amd::find_matches<match_set_t>(*dev, sp1.graph, sp2.graph, sp1.tableau, sp2.tableau,
                                                                      centrolign_amd::match_params_of(path_match_finder, score_function));
                t_match_gpu += std::chrono::duration<double>(clk::now() - m1).count();
                bool same = ours.size() == matches.size();
                for (size_t i = 0; same && i < ours.size(); ++i)
                    same = ours[i].walks1 == matches[i].walks1 && ours[i].walks2 == matches[i].walks2 && ours[i].count1 == matches[i].count1 &&
                           ours[i].count2 == matches[i].count2 && ours[i].full_length == matches[i].full_length;
                if (!same) ++mismatched;
                printf("merge %zu: %zu match sets, GPU find_matches %s the reference\n", merges, matches.size(), same ? "==" : "!=");
                matches = std::move(ours);
            }
            PathMerge<uint32_t, uint8_t> pm1(sp1.graph, sp1.tableau);
            PathMerge<uint32_t, uint8_t> pm2(sp2.graph, sp2.tableau);
            if (seams) seams_of_merge(matches, sp1, sp2, pm1, pm2);
            next_problem.alignment = whole_align ? align_core_both(matches, sp1, sp2, pm1, pm2) : align_both(matches, sp1, sp2, pm1, pm2);
            BaseGraph fused = sp1.graph;
            fuse(fused, sp2.graph, sp1.tableau, sp2.tableau, next_problem.alignment);
            if (whole_align) {   // fuse through the library, against the graph the reference has just fused
                const bool same = centrolign_amd::fuse_equals(sp1.graph, sp2.graph, sp1.tableau, sp2.tableau, next_problem.alignment, fused);
                if (!same) ++mismatched;
                printf("merge %zu: fused graph %zu nodes, cl_fuse %s the reference\n", merges - 1, (size_t)fused.node_size(), same ? "==" : "!=");
            }
            next_problem.graph = std::move(fused);
            next_problem.tableau = sp1.tableau;
            next_problem.complete = true;
        }
    }
};

}  // namespace

// "facade": the reference's own Core (files in, execute(), root subproblem out) beside centrolign_amd::Core of
// include/centrolign_amd/core_facade.hpp, the same calls in the same order — and the same GFA / CIGAR text
static int facade_mode(int argc, char** argv) {
    const std::string fasta = argv[1], tree_file = (argc > 2 && std::string(argv[2]) != "-") ? argv[2] : "";
    Parameters params;
    params.set<std::string>("fasta_name", fasta);
    if (argc > 3) params.set<int64_t>("max_num_match_pairs", atoll(argv[3]));
    params.validate();
    logging::level = logging::Silent;
    std::string want;
    {
        std::ifstream fin(fasta);
        auto parsed = parse_fasta(fin);
        std::vector<std::string> names;
        for (const auto& p : parsed) names.push_back(p.first);
        std::string newick = in_order_newick_string(names);
        if (!tree_file.empty()) { std::ifstream tin(tree_file); std::stringstream ss; ss << tin.rdbuf(); newick = ss.str(); }
        Tree tree(newick);
        Core core(std::move(parsed), std::move(tree));
        if (names.size() == 2) params.set<bool>("preserve_subproblems", true);   // as main() does: the CIGAR needs the leaves (src/main.cpp:272-276)
        params.apply(core);
        core.execute();
        std::stringstream out;
        const auto& root = core.root_subproblem();
        if (names.size() == 2) out << explicit_cigar(root.alignment, core.leaf_subproblem(names[0]).graph, core.leaf_subproblem(names[1]).graph) << '\n';
        else write_gfa(root.graph, root.tableau, out);
        want = out.str();
    }
    centrolign_amd::Core core(fasta, tree_file);
    if (argc > 3) core.anchorer.max_num_match_pairs = (uint64_t)atoll(argv[3]);
    core.execute();
    std::string got = core.output();
    while (!got.empty() && got.back() == '\n') got.pop_back();
    while (!want.empty() && want.back() == '\n') want.pop_back();
    const auto& root = core.root_subproblem();
    printf("facade: root subproblem %llu nodes, %llu paths, %zu aligned pairs, complete %d; output %zu bytes %s the reference's Core (%zu bytes)\n",
           (unsigned long long)root.graph.n_nodes, (unsigned long long)root.graph.n_paths, root.alignment.size(), (int)root.complete, got.size(),
           got == want ? "==" : "!=", want.size());
    printf(got == want ? "DROP-IN OK\n" : "DROP-IN FAILED\n");
    return got == want ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <fasta> [newick|-] [max_num_match_pairs] [core|seams|facade]\n", argv[0]); return 2; }
    try {
        if (argc > 4 && std::string(argv[4]) == "facade") return facade_mode(argc, argv);
        Parameters params;
        params.set<std::string>("fasta_name", argv[1]);
        if (argc > 3) params.set<int64_t>("max_num_match_pairs", atoll(argv[3]));
        params.validate();
        logging::level = logging::Silent;
        std::ifstream fin(argv[1]);
        auto parsed = parse_fasta(fin);
        std::vector<std::string> names;
        for (const auto& p : parsed) names.push_back(p.first);
        std::string newick;
        if (argc > 2 && std::string(argv[2]) != "-") {
            std::ifstream tin(argv[2]);
            std::stringstream ss;
            ss << tin.rdbuf();
            newick = ss.str();
        } else {
            newick = in_order_newick_string(names);
        }
        Tree tree(newick);
        DemoCore core(std::move(parsed), std::move(tree));
        params.apply(core);
        core.preserve_subproblems = true;
        centrolign_amd::Device dev(0);
        core.dev = &dev;
        core.whole_align = argc > 4 && std::string(argv[4]) == "core";
        core.seams = argc > 4 && std::string(argv[4]) == "seams";
        if (core.seams) core.leaves_first();
        core.run();
        if (core.whole_align)
        {
            printf("%zu merges, find_matches: reference CPU %.3f s, adapter+GPU %.3f s (incl. flattening and rebuilding the vectors)\n", core.merges,
                   core.t_match_cpu, core.t_match_gpu);
            printf("%zu merges, Core::align: reference CPU %.3f s, adapter+GPU %.3f s (incl. flattening)\n", core.merges, core.t_cpu, core.t_gpu);
        }
        else
            printf("%zu merges, %zu subproblems, subalign loop: reference CPU %.3f s, adapter+GPU %.3f s (incl. flatten, H2D, D2H)\n",
                   core.merges, core.problems, core.t_cpu, core.t_gpu);
        printf(core.mismatched ? "DROP-IN FAILED\n" : "DROP-IN OK\n");
        return core.mismatched ? 1 : 0;
    } catch (std::exception& ex) {
        fprintf(stderr, "adapter_demo: %s\n", ex.what());
        return 3;
    }
}
