// core_facade.hpp — centrolign's Core / Execution surface (include/centrolign/core.hpp:30-103, include/centrolign/execution.hpp:18-120) over
// the MI355X library: a caller that drives the reference through `Core core(fasta, tree); core.<tunable> = ...; core.execute();
// core.root_subproblem()` does the same here, with the same member names, and every merge runs through cl_merge
// (match finding, Core::align, fuse on the device path).  Header-only C++11, nothing but include/centrolign_amd.h underneath.
//
// What is mirrored, member for member:
//   Core:      the two constructors (files / parsed inputs — the guide tree as Newick text, which is what Tree's constructor takes,
//              src/tree.cpp:39-160), execute(), restart(), root_subproblem(), leaf_subproblem(name), and the public tunables
//              skip_calibration, cyclize_tandem_duplications, threads, max_tandem_duplication_search_rounds, subproblems_prefix,
//              induced_pairwise_prefix (core.hpp:70-103).  The configurable submodules (anchorer, partitioner, stitcher, ...) are the
//              parameter blocks of cl_merge_params under the reference's submodule names: core.anchorer.max_num_match_pairs,
//              core.path_match_finder.max_count, core.stitcher.alignment_params, ...
//   Execution: finished(), next(), current(), leaf_subproblem(name), final_subproblem(), leaf_subproblems(), subproblem_hash(),
//              leaf_descendents() over the plan cl_msa_plan_create returns (Tree pruning / binarisation / small-first post-order of
//              src/execution.cpp:12-92 happen there).
//   Subproblem: graph, tableau, alignment, name, complete (execution.hpp:18-28) — the graph as the C ABI's flat arrays
//              (cl_owned_base_graph + view), the alignment as AlignedPair-compatible (node, node) pairs with gap = UINT64_MAX.
// Not mirrored: the -c flow step by step (execute() hands the whole job to cl_msa when cyclize_tandem_duplications is set and keeps
// its text), logging, bonds_prefix / subalignments_filepath.
#ifndef CENTROLIGN_AMD_CORE_FACADE_HPP
#define CENTROLIGN_AMD_CORE_FACADE_HPP

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../centrolign_amd.h"

namespace centrolign_amd {

struct SentinelTableau { uint64_t src_id = UINT64_MAX, snk_id = UINT64_MAX; };   // include/centrolign/modify_graph.hpp:33-38

// include/centrolign/execution.hpp:18-28
struct Subproblem {
    cl_owned_base_graph* owned = nullptr;          // the library's graph; `graph` is the view of it (flat arrays of include/centrolign_amd.h)
    cl_base_graph graph;
    SentinelTableau tableau;
    std::vector<std::pair<uint64_t, uint64_t>> alignment;   // Alignment = vector<AlignedPair>, gap = UINT64_MAX (alignment.hpp:34-51)
    std::string name;
    bool complete = false;

    Subproblem() { std::memset(&graph, 0, sizeof(graph)); }
    Subproblem(const Subproblem&) = delete;
    Subproblem& operator=(const Subproblem&) = delete;
    Subproblem(Subproblem&& o) noexcept : owned(o.owned), graph(o.graph), tableau(o.tableau), alignment(std::move(o.alignment)), name(std::move(o.name)), complete(o.complete) { o.owned = nullptr; }
    ~Subproblem() { if (owned) cl_owned_base_graph_free(owned); }
    void take(cl_owned_base_graph* g) {
        if (owned) cl_owned_base_graph_free(owned);
        owned = g;
        cl_owned_base_graph_view(g, &graph);
        tableau.src_id = graph.src_id;
        tableau.snk_id = graph.snk_id;
    }
};

inline void check(int rc, cl_context* ctx, const char* what) {
    if (rc != CL_OK) throw std::runtime_error(std::string(what) + ": " + cl_last_error(ctx) + " (" + std::to_string(rc) + ")");
}

// include/centrolign/execution.hpp:33-120
class Execution {
public:
    Execution() = default;
    // names_and_sequences as parse_fasta returns them; newick = the guide tree's text ("" : in-order tree over the names, src/tree.cpp:17-37)
    Execution(std::vector<std::pair<std::string, std::string>>&& names_and_sequences, const std::string& newick) {
        input = std::move(names_and_sequences);
        std::vector<const char*> names;
        for (const auto& p : input) names.push_back(p.first.c_str());
        cl_msa_plan plan;
        check(cl_msa_plan_create(nullptr, newick.empty() ? nullptr : newick.c_str(), names.data(), names.size(), &plan), nullptr, "Execution: guide tree");
        n_leaves = plan.n_leaves;
        leaf_sequence.assign(plan.leaf_sequence, plan.leaf_sequence + plan.n_leaves);
        merge_children.assign(plan.merge_children, plan.merge_children + 2 * plan.n_merges);
        cl_msa_plan_free(&plan);
        problems.resize(n_leaves + merge_children.size() / 2);
        for (uint64_t i = 0; i < n_leaves; ++i) {
            const auto& in = input[leaf_sequence[i]];
            Subproblem& sp = problems[i];
            cl_owned_base_graph* g = nullptr;
            check(cl_leaf_graph(in.second.data(), in.second.size(), &g), nullptr, "Execution: leaf graph");   // make_base_graph + add_sentinels (src/execution.cpp:66-73)
            sp.take(g);
            sp.name = in.first;
            sp.complete = true;
            leaf_of_name[in.first] = i;
        }
    }
    bool finished() const { return next_merge == merge_children.size() / 2; }
    // the next subproblem and its two children (parent first)
    std::tuple<Subproblem*, Subproblem*, Subproblem*> next() {
        if (finished()) return std::make_tuple((Subproblem*)nullptr, (Subproblem*)nullptr, (Subproblem*)nullptr);
        const uint64_t k = next_merge++;
        return std::make_tuple(&problems[n_leaves + k], &problems[merge_children[2 * k]], &problems[merge_children[2 * k + 1]]);
    }
    std::tuple<const Subproblem*, const Subproblem*, const Subproblem*> current() const {
        if (next_merge == 0) return std::make_tuple((const Subproblem*)nullptr, (const Subproblem*)nullptr, (const Subproblem*)nullptr);
        const uint64_t k = next_merge - 1;
        return std::make_tuple(&problems[n_leaves + k], &problems[merge_children[2 * k]], &problems[merge_children[2 * k + 1]]);
    }
    const Subproblem& leaf_subproblem(const std::string& name) const { return problems.at(leaf_of_name.at(name)); }
    Subproblem& final_subproblem() { return problems.back(); }
    const Subproblem& final_subproblem() const { return problems.back(); }
    std::vector<Subproblem*> leaf_subproblems() {   // in the order the reference calibrates them (tree-id order)
        std::vector<Subproblem*> out;
        for (uint64_t i = 0; i < n_leaves; ++i) out.push_back(&problems[i]);
        return out;
    }
    std::vector<std::string> leaf_descendents(const Subproblem& sp) const {
        std::vector<std::string> out;
        collect(&sp - problems.data(), out);
        return out;
    }
    // the hash -S names a subproblem's file by (Core::subproblem_file_name, src/core.cpp:378-380)
    uint64_t subproblem_hash(const Subproblem& sp) const {
        const auto names = leaf_descendents(sp);
        std::vector<const char*> p;
        for (const auto& n : names) p.push_back(n.c_str());
        char hex[17];
        if (cl_subproblem_hash_hex(p.data(), p.size(), hex) != CL_OK) throw std::runtime_error("subproblem_hash");
        return std::strtoull(hex, nullptr, 16);
    }
    uint64_t num_leaves() const { return n_leaves; }
    uint64_t num_merges() const { return merge_children.size() / 2; }

private:
    void collect(uint64_t slot, std::vector<std::string>& out) const {
        if (slot < n_leaves) { out.push_back(problems[slot].name); return; }
        collect(merge_children[2 * (slot - n_leaves)], out);
        collect(merge_children[2 * (slot - n_leaves) + 1], out);
    }
    std::vector<std::pair<std::string, std::string>> input;
    uint64_t n_leaves = 0, next_merge = 0;
    std::vector<uint64_t> leaf_sequence, merge_children;
    std::vector<Subproblem> problems;
    std::unordered_map<std::string, uint64_t> leaf_of_name;
};

// include/centrolign/core.hpp:30-103
class Core {
public:
    // parse files to construct core (either may be - for stdin; an empty tree file name = the in-order tree)
    Core(const std::string& fasta_file, const std::string& tree_file) {
        const std::string fa = slurp(fasta_file), nwk = tree_file.empty() ? std::string() : slurp(tree_file);
        cl_fasta f;
        check(cl_parse_fasta(nullptr, fa.data(), fa.size(), &f), nullptr, "Core: FASTA");
        std::vector<std::pair<std::string, std::string>> in;
        for (uint64_t i = 0; i < f.n_sequences; ++i) in.emplace_back(f.names[i], std::string(f.sequences[i], f.lengths[i]));
        cl_fasta_free(&f);
        init(std::move(in), nwk);
    }
    // construct core from already-parsed inputs (consumes the inputs)
    Core(std::vector<std::pair<std::string, std::string>>&& names_and_sequences, const std::string& newick) { init(std::move(names_and_sequences), newick); }
    ~Core() { if (ctx) cl_context_destroy(ctx); }
    Core(const Core&) = delete;
    Core& operator=(const Core&) = delete;

    cl_merge_params merge_params;                  // (declared first: the submodule names below refer into it)
    /* the configurable submodules, as the library's parameter blocks under the reference's member names */
    cl_match_params&      path_match_finder = merge_params.match;            // max_count, use_color_set_size, ...
    cl_anchor_params&     anchorer = merge_params.align.anchor;              // max_num_match_pairs, global_anchoring, gap parameters, score function
    cl_partition_params&  partitioner = merge_params.align.partition;
    cl_stitch_params&     stitcher = merge_params.align.stitch;              // alignment_params, min_wfa_size, ...

    bool skip_calibration = false;                 // don't calibrate the scale of the scoring function before executing
    bool cyclize_tandem_duplications = false;      // merge tandem duplications into cycles in the final graph (-c)
    uint64_t threads = 1;                          // worker contexts that run independent merges and calibrations side by side (cl_msa path)
    size_t max_tandem_duplication_search_rounds = 3;
    std::string subproblems_prefix;                // -S
    std::string induced_pairwise_prefix;           // -A
    int device = 0;                                // which GPU

    // trigger the MSA (src/core.cpp:45-120): calibration, then Execution's merges in order, each one cl_merge
    void execute() {
        if (!ctx) { ctx = cl_context_create(device); if (!ctx) throw std::runtime_error(std::string("Core: ") + cl_last_error(nullptr)); }
        if (cyclize_tandem_duplications || threads > 1 || !subproblems_prefix.empty() || !induced_pairwise_prefix.empty() || restarting) { execute_whole(); return; }
        if (!skip_calibration) {   // Core::calibrate_anchor_scores (src/core.cpp:122-184): the mean of the leaves' intrinsic scales, in leaf order
            double sum = 0;
            const auto leaves = execution.leaf_subproblems();
            for (Subproblem* leaf : leaves) {
                double s = 0;
                check(cl_leaf_intrinsic_scale(ctx, &leaf->graph, &merge_params.match, &merge_params.align.anchor, &s), ctx, "Core: calibration");
                sum += s;
            }
            merge_params.align.anchor.score_scale = sum / (double)leaves.size();
        }
        while (!execution.finished()) {
            Subproblem *next = nullptr, *sp1 = nullptr, *sp2 = nullptr;
            std::tie(next, sp1, sp2) = execution.next();
            cl_merge_result r;
            check(cl_merge(ctx, &sp1->graph, &sp2->graph, &merge_params, &r), ctx, "Core: merge");
            next->take(r.fused);
            r.fused = nullptr;
            next->alignment.resize(r.align.alignment.n_pairs);
            for (uint64_t i = 0; i < r.align.alignment.n_pairs; ++i) next->alignment[i] = std::make_pair(r.align.alignment.pairs[2 * i], r.align.alignment.pairs[2 * i + 1]);
            next->complete = true;
            cl_merge_result_free(&r);
        }
    }
    // load alignments from the prefix and start where they left off (-R)
    void restart() { restarting = true; execute(); }

    const Subproblem& root_subproblem() const { return execution.final_subproblem(); }
    const Subproblem& leaf_subproblem(const std::string& name) const { return execution.leaf_subproblem(name); }

    // what main() prints (src/main.cpp:270-301): the explicit CIGAR of two sequences, the GFA of more
    std::string output() const {
        if (!whole_text.empty()) return whole_text;
        const Subproblem& root = execution.final_subproblem();
        char* text = nullptr;
        uint64_t len = 0;
        if (execution.num_leaves() == 2) {
            const Subproblem& a = execution.leaf_subproblem(sequence_names.front());   // main(): the leaves in the order of the FASTA (src/main.cpp:292-296)
            const Subproblem& b = execution.leaf_subproblem(sequence_names.back());
            std::vector<uint64_t> flat;
            for (const auto& p : root.alignment) { flat.push_back(p.first); flat.push_back(p.second); }
            check(cl_explicit_cigar(&a.graph, &b.graph, flat.data(), root.alignment.size(), &text, &len), ctx, "Core: CIGAR");
        } else {
            const auto names = execution.leaf_descendents(root);
            std::vector<const char*> p;
            for (const auto& n : names) p.push_back(n.c_str());
            check(cl_write_gfa(&root.graph, p.data(), 1, &text, &len), ctx, "Core: GFA");
        }
        std::string out(text, len);
        free(text);
        return out;
    }

    Execution execution;

private:
    void init(std::vector<std::pair<std::string, std::string>>&& in, const std::string& newick) {
        cl_merge_params_default(&merge_params);
        std::ostringstream fa;
        for (const auto& p : in) { fa << '>' << p.first << '\n' << p.second << '\n'; sequence_names.push_back(p.first); }
        fasta_text = fa.str();
        newick_text = newick;
        execution = Execution(std::move(in), newick);
    }
    // -c, -S / -R, -A, worker threads: the library's own driver (cl_msa) runs the whole job; the root subproblem is read back from its text
    void execute_whole() {
        cl_msa_params p;
        cl_msa_params_default(&p);
        p.merge = merge_params;
        p.skip_calibration = skip_calibration;
        p.n_workers = (int)threads;
        p.subproblems_prefix = subproblems_prefix.empty() ? nullptr : subproblems_prefix.c_str();
        p.restart = restarting;
        p.induced_pairwise_prefix = induced_pairwise_prefix.empty() ? nullptr : induced_pairwise_prefix.c_str();
        p.cyclize = cyclize_tandem_duplications;
        p.max_tandem_duplication_search_rounds = max_tandem_duplication_search_rounds;
        char* text = nullptr;
        uint64_t len = 0;
        check(cl_msa(ctx, fasta_text.data(), fasta_text.size(), newick_text.empty() ? nullptr : newick_text.c_str(), &p, &text, &len, nullptr), ctx, "Core: cl_msa");
        whole_text.assign(text, len);
        free(text);
        if (execution.num_leaves() > 2) {
            cl_owned_base_graph* g = nullptr;
            check(cl_read_gfa(whole_text.data(), whole_text.size(), 1, &g, nullptr, nullptr), ctx, "Core: reading the root back");
            execution.final_subproblem().take(g);
            execution.final_subproblem().complete = true;
        }
    }
    static std::string slurp(const std::string& path) {
        std::stringstream ss;
        if (path == "-") ss << std::cin.rdbuf();
        else {
            std::ifstream in(path);
            if (!in) throw std::runtime_error("could not open " + path);
            ss << in.rdbuf();
        }
        return ss.str();
    }
    cl_context* ctx = nullptr;
    std::string fasta_text, newick_text, whole_text;
    std::vector<std::string> sequence_names;
    bool restarting = false;
};

}  // namespace centrolign_amd

#endif
