// cl_plan_api.cpp — the front of the reference's driver, behind the C ABI (include/centrolign_amd.h):
//   cl_parse_fasta        parse_fasta                          src/utility.cpp:19-65
//   cl_msa_plan_create    Tree(newick) + Execution's setup     src/tree.cpp:39-160, src/execution.cpp:12-92
//                         (prune to the FASTA's names, compact, binarize, small_first_postorder) -> leaf order + merge order
//   cl_msa                main() -> Core::execute -> output    src/main.cpp:239-301, src/core.cpp:63-94
// Host code.  What must be reproduced is the ORDER the reference works in: leaves are calibrated in tree-id order (the mean of the
// intrinsic scales is a floating-point sum, src/core.cpp:169-173), merges run in small_first_postorder order, and the first child of a
// tree node is graph 1 of its merge (src/execution.cpp:98-124).  Tree ids follow the Newick text (a node is numbered when its '('
// is met, a leaf when the ',' or ')' behind it is), survive prune / compact as a stable renumbering, and binarize appends its new
// nodes; the tree is kept here as flat arrays (parent, ordered child lists) with those ids.
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <fstream>
#include <iterator>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "cl_internal.hpp"

namespace {

constexpr uint32_t kNoNode = 0xFFFFFFFFu;

struct GuideTree {
    std::vector<uint32_t> parent;
    std::vector<std::vector<uint32_t>> kids;
    std::vector<std::string> label;
    uint32_t root = kNoNode;
    std::string error;

    uint32_t add(uint32_t par) {
        const uint32_t id = (uint32_t)parent.size();
        parent.push_back(par);
        kids.emplace_back();
        label.emplace_back();
        if (par != kNoNode) kids[par].push_back(id);
        return id;
    }

    // label text between two structural characters: up to an unquoted ':', trimmed, one pair of quotes removed (src/tree.cpp:198-245)
    bool set_label(uint32_t v, const std::string& s, size_t b, size_t e) {
        bool q = false;
        size_t colon = e;
        for (size_t i = b; i < e; ++i) {
            if (s[i] == '"') q = !q;
            else if (!q && s[i] == ':') { colon = i; break; }
        }
        if (colon != e) {
            size_t d = colon + 1;
            while (d < e && isspace((unsigned char)s[d])) ++d;
            if (d == e) { error = "Newick string has ':' without a distance following it"; return false; }
        }
        size_t lo = b, hi = colon;
        while (lo < hi && isspace((unsigned char)s[lo])) ++lo;
        while (hi > lo + 1 && isspace((unsigned char)s[hi - 1])) --hi;
        for (size_t i = lo + 1; i + 1 < hi; ++i)
            if (s[i] == '"') { error = "Newick string label has internal quotation mark: " + s.substr(lo, hi - lo); return false; }
        if (lo < hi && s[lo] == '"') {
            if (lo + 1 == hi) { error = "Newick string label consists of only one quotation mark"; return false; }
            if (s[hi - 1] != '"') { error = "Newick string label has unmatched quotation mark: " + s.substr(lo, hi - lo); return false; }
            ++lo; --hi;
        }
        label[v] = s.substr(lo, hi - lo);
        return true;
    }

    bool parse(const std::string& s) {
        // the checks of src/tree.cpp:41-60
        size_t semi = std::string::npos;
        {
            bool q = false;
            for (size_t i = 0; i < s.size(); ++i) {
                if (s[i] == '"') q = !q;
                else if (!q && s[i] == ';') { semi = i; break; }
            }
        }
        if (semi == std::string::npos) { error = "Newick string is missing a terminating ';'"; return false; }
        for (size_t i = semi + 1; i < s.size(); ++i)
            if (!isspace((unsigned char)s[i])) { error = "Newick string includes characters after the terminating ';'"; return false; }
        if (std::count(s.begin(), s.end(), '"') % 2 == 1) { error = "Newick string has an odd number of quotation marks"; return false; }
        if (s.find('\'') != std::string::npos) { error = "Newick string parser does not support single quotes (')"; return false; }
        // structural characters outside quotes, in text order
        std::vector<size_t> marks;
        {
            bool q = false;
            for (size_t i = 0; i <= semi; ++i) {
                if (s[i] == '"') q = !q;
                else if (!q && (s[i] == '(' || s[i] == ')' || s[i] == ',' || s[i] == ';')) marks.push_back(i);
            }
        }
        bool any_paren = false;
        for (size_t m : marks) any_paren |= s[m] == '(' || s[m] == ')';
        if (!any_paren) {   // no parentheses (src/tree.cpp:62-70): one node whose label is everything up to the ';', commas included
            root = add(kNoNode);
            return set_label(root, s, 0, semi) && index_labels();
        }
        std::vector<uint32_t> open;     // nodes whose child list is being read
        uint32_t closed = kNoNode;      // the node whose ')' was the previous mark: the text up to the next mark is ITS label
        size_t from = 0;
        for (size_t m : marks) {
            const char c = s[m];
            if (c == ';') {
                if (closed != kNoNode && !set_label(closed, s, from, m)) return false;
                break;
            }
            if (c == '(') {
                if (open.empty()) {
                    if (!parent.empty()) { error = "Newick string encodes a disconnected tree"; return false; }
                    root = add(kNoNode);
                    open.push_back(root);
                } else {
                    open.push_back(add(open.back()));
                }
                closed = kNoNode;
            } else {   // ',' or ')': ends a leaf's text, or the label of the node that has just been closed
                if (open.empty()) { error = "Newick string is not a tree"; return false; }
                const uint32_t v = closed != kNoNode ? closed : add(open.back());
                if (!set_label(v, s, from, m)) return false;
                closed = kNoNode;
                if (c == ')') { closed = open.back(); open.pop_back(); }
            }
            from = m + 1;
        }
        return index_labels();
    }

    std::unordered_map<std::string, uint32_t> by_label;
    bool index_labels() {
        for (uint32_t v = 0; v < label.size(); ++v) {
            if (label[v].find('#') != std::string::npos) { error = "Tree labels may not include '#': " + label[v]; return false; }
            if (label[v].empty()) continue;
            if (!by_label.emplace(label[v], v).second) { error = "Duplicate label " + label[v] + " in guide tree"; return false; }
        }
        return true;
    }

    // drop the nodes not kept; ids close up in order, child lists keep their order (Tree::filter, src/tree.cpp:469-525)
    void keep_only(const std::vector<char>& keep) {
        std::vector<uint32_t> new_id(parent.size(), kNoNode);
        uint32_t n = 0;
        for (uint32_t v = 0; v < parent.size(); ++v) if (keep[v]) new_id[v] = n++;
        if (n == parent.size()) return;
        std::vector<std::vector<uint32_t>> k2(n);
        std::vector<std::string> l2(n);
        for (uint32_t v = 0; v < parent.size(); ++v) {
            if (!keep[v]) continue;
            for (uint32_t c : kids[v]) if (keep[c]) k2[new_id[v]].push_back(new_id[c]);
            l2[new_id[v]] = std::move(label[v]);
        }
        kids.swap(k2);
        label.swap(l2);
        parent.assign(n, kNoNode);
        for (uint32_t v = 0; v < n; ++v) for (uint32_t c : kids[v]) parent[c] = v;
        root = kNoNode;
        for (uint32_t v = 0; v < n; ++v) if (parent[v] == kNoNode) { root = v; break; }
        by_label.clear();
        for (uint32_t v = 0; v < n; ++v) if (!label[v].empty()) by_label[label[v]] = v;
    }

    // Tree::prune (src/tree.cpp:385-425): the kept leaves, their ancestors, minus the unary stem above their last common ancestor
    void prune_to(const std::vector<uint32_t>& leaves) {
        std::vector<char> keep(parent.size(), 0);
        for (uint32_t v : leaves)
            for (uint32_t h = v; h != kNoNode && !keep[h]; h = parent[h]) keep[h] = 1;
        uint32_t h = root;
        while (h != kNoNode && keep[h]) {
            uint32_t only = kNoNode, cnt = 0;
            for (uint32_t c : kids[h]) if (keep[c]) { only = c; ++cnt; }
            if (cnt != 1) break;
            keep[h] = 0;
            h = only;
        }
        if (!leaves.empty()) keep[leaves.front()] = 1;
        keep_only(keep);
    }

    // Tree::compact (src/tree.cpp:427-467): a node with one child hands that child to its own parent — APPENDED to the parent's list
    void compact() {
        std::vector<char> keep(parent.size(), 1);
        for (uint32_t v = 0; v < parent.size(); ++v) {
            if (kids[v].size() != 1) continue;
            keep[v] = 0;
            const uint32_t c = kids[v].front();
            if (v == root) { root = c; parent[c] = kNoNode; }
            else { kids[parent[v]].push_back(c); parent[c] = parent[v]; }
        }
        keep_only(keep);
    }

    // Tree::binarize (src/tree.cpp:281-329): a node with k > 2 children keeps the first; k - 2 new nodes (appended ids) hang off
    // one another to the right, each with the next child on its left, the last one with the final two
    void binarize() {
        const uint32_t n0 = (uint32_t)parent.size();
        for (uint32_t v = 0; v < n0; ++v) {
            if (kids[v].size() <= 2) continue;
            const std::vector<uint32_t> ch = kids[v];
            kids[v].assign(1, ch.front());
            uint32_t prev = v;
            for (size_t i = 2; i < ch.size(); ++i) {
                const uint32_t nn = add(prev);
                kids[nn].push_back(ch[i - 1]);
                parent[ch[i - 1]] = nn;
                prev = nn;
            }
            kids[prev].push_back(ch.back());
            parent[ch.back()] = prev;
        }
    }

    // Tree::postorder (src/tree.cpp:527-556) visits the LAST child's subtree first (children are pushed in order and popped from the back)
    std::vector<uint32_t> postorder() const {
        std::vector<uint32_t> order;
        std::vector<std::pair<uint32_t, bool>> st;
        if (root != kNoNode) st.emplace_back(root, false);
        while (!st.empty()) {
            if (st.back().second) { order.push_back(st.back().first); st.pop_back(); continue; }
            st.back().second = true;
            const uint32_t v = st.back().first;
            for (uint32_t c : kids[v]) st.emplace_back(c, false);
        }
        return order;
    }

    // Tree::small_first_postorder (src/tree.cpp:558-586): ids stably sorted by (leaves below, position in postorder)
    std::vector<uint32_t> small_first_postorder() const {
        const size_t n = parent.size();
        std::vector<std::pair<uint64_t, uint64_t>> pri(n, {0, 0});
        uint64_t p = 0;
        for (uint32_t v : postorder()) {
            if (kids[v].empty()) pri[v].first = 1;
            else for (uint32_t c : kids[v]) pri[v].first += pri[c].first;
            pri[v].second = p++;
        }
        std::vector<uint32_t> order(n);
        for (uint32_t v = 0; v < n; ++v) order[v] = v;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pri[a] < pri[b]; });
        return order;
    }
};

// in_order_newick_string (src/tree.cpp:17-37): the left-deep tree over the sequences in file order
bool in_order_newick(const std::vector<std::string>& names, std::string& out, std::string& error) {
    for (const auto& n : names)
        if (n.find('"') != std::string::npos) { error = "Sequence names cannot have internal quotation marks: " + n; return false; }
    out.assign(names.size() > 1 ? names.size() - 1 : 0, '(');
    if (!names.empty()) {
        out += '"' + names.front() + '"';
        for (size_t i = 1; i < names.size(); ++i) out += ",\"" + names[i] + "\")";
    }
    out += ';';
    return true;
}

}  // namespace

int cl_processed_guide_tree(const char* newick, const char* const* names, uint64_t n_names, ClGuideTreeView& out, std::string& error) {
    std::vector<std::string> nm(names, names + n_names);
    std::string text;
    GuideTree t;
    if (newick && *newick) text = newick;
    else if (!in_order_newick(nm, text, t.error)) { error = t.error; return CL_ERR_INVALID_ARGUMENT; }
    if (!t.parse(text)) { error = t.error; return CL_ERR_INVALID_ARGUMENT; }
    std::vector<uint32_t> leaf_ids;
    for (const auto& n : nm) {
        auto it = t.by_label.find(n);
        if (it == t.by_label.end() || !t.kids[it->second].empty()) { error = "Guide tree does not include sequence " + n + " as a leaf"; return CL_ERR_INVALID_ARGUMENT; }
        leaf_ids.push_back(it->second);
    }
    t.prune_to(leaf_ids);
    t.compact();
    t.binarize();
    out.kids = t.kids;
    out.label = t.label;
    out.root = t.root;
    out.postorder = t.postorder();
    return CL_OK;
}

struct cl_fasta_owned {
    std::vector<std::string> names, seqs;
    std::vector<const char*> name_ptr, seq_ptr;
    std::vector<uint64_t> seq_len;
};

extern "C" {

void cl_fasta_free(cl_fasta* f) {
    if (!f) return;
    delete (cl_fasta_owned*)f->owner;
    memset(f, 0, sizeof(*f));
}

int cl_parse_fasta(cl_context* ctx, const char* text, uint64_t len, cl_fasta* out) {
    if (!text || !out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    std::unique_ptr<cl_fasta_owned> own(new cl_fasta_owned());
    // line by line like std::getline in a `while (in)` loop (src/utility.cpp:27-62): after the last line one more, empty, line is seen
    uint64_t at = 0, line_num = 0;
    size_t prev = SIZE_MAX, prev_prev = SIZE_MAX;
    bool more = true;
    while (more) {
        uint64_t end = at;
        while (end < len && text[end] != '\n') ++end;
        const std::string line(text + at, text + end);
        if (end >= len) more = false;
        at = end + 1;
        ++line_num;
        if (!line.empty() && line.front() == '>') {
            const size_t sp = line.find(' ');
            own->names.push_back(line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1));
            own->seqs.emplace_back();
            if (own->names.back().empty()) { cl_set_error(ctx, "FASTA input is missing sequence name at line %llu", (unsigned long long)line_num); return CL_ERR_INVALID_ARGUMENT; }
            prev = prev_prev = SIZE_MAX;
        } else {
            if (own->names.empty()) {
                if (line.empty() && !more) break;
                cl_set_error(ctx, "FASTA input does not have sequence name line");
                return CL_ERR_INVALID_ARGUMENT;
            }
            if (prev_prev != SIZE_MAX && prev != prev_prev && !line.empty()) {
                cl_set_error(ctx, "Encountered sequence lines of unequal lengths that were not followed by a sequence name at line %llu of FASTA input", (unsigned long long)line_num);
                return CL_ERR_INVALID_ARGUMENT;
            }
            if (prev != SIZE_MAX && line.size() > prev) {
                cl_set_error(ctx, "Encountered adjacent sequence lines of increasing lengths at line %llu of FASTA input", (unsigned long long)line_num);
                return CL_ERR_INVALID_ARGUMENT;
            }
            own->seqs.back() += line;
            prev_prev = prev;
            prev = line.size();
        }
    }
    if (own->names.empty()) { cl_set_error(ctx, "FASTA file is empty"); return CL_ERR_INVALID_ARGUMENT; }
    for (size_t i = 0; i < own->names.size(); ++i) {
        own->name_ptr.push_back(own->names[i].c_str());
        own->seq_ptr.push_back(own->seqs[i].c_str());
        own->seq_len.push_back(own->seqs[i].size());
    }
    out->n_sequences = own->names.size();
    out->names = own->name_ptr.data();
    out->sequences = own->seq_ptr.data();
    out->lengths = own->seq_len.data();
    out->owner = own.release();
    return CL_OK;
}

void cl_msa_plan_free(cl_msa_plan* p) {
    if (!p) return;
    free(p->leaf_sequence);
    free(p->merge_children);
    memset(p, 0, sizeof(*p));
}

int cl_msa_plan_create(cl_context* ctx, const char* newick, const char* const* names, uint64_t n_names, cl_msa_plan* out) {
    if (!names || !out || n_names == 0) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    memset(out, 0, sizeof(*out));
    std::vector<std::string> nm(names, names + n_names);
    std::unordered_map<std::string, uint64_t> seq_of;
    for (uint64_t i = 0; i < n_names; ++i)
        if (!seq_of.emplace(nm[i], i).second) { cl_set_error(ctx, "FASTA contains duplicate name %s", nm[i].c_str()); return CL_ERR_INVALID_ARGUMENT; }
    std::string text;
    GuideTree t;
    if (newick && *newick) text = newick;
    else if (!in_order_newick(nm, text, t.error)) { cl_set_error(ctx, "%s", t.error.c_str()); return CL_ERR_INVALID_ARGUMENT; }
    if (!t.parse(text)) { cl_set_error(ctx, "%s", t.error.c_str()); return CL_ERR_INVALID_ARGUMENT; }
    // the match between the FASTA and the tree (src/execution.cpp:31-47)
    std::vector<uint32_t> leaf_ids;
    for (const auto& n : nm) {
        auto it = t.by_label.find(n);
        if (it == t.by_label.end()) { cl_set_error(ctx, "Guide tree does not include sequence %s", n.c_str()); return CL_ERR_INVALID_ARGUMENT; }
        if (!t.kids[it->second].empty()) { cl_set_error(ctx, "Sequence %s is not a leaf in the guide tree", n.c_str()); return CL_ERR_INVALID_ARGUMENT; }
        leaf_ids.push_back(it->second);
    }
    t.prune_to(leaf_ids);
    t.compact();
    t.binarize();
    // slots: leaves in tree-id order (Execution::leaf_subproblems, src/execution.cpp:141-153), then one per merge in execution order
    const size_t n = t.parent.size();
    std::vector<uint64_t> slot(n, UINT64_MAX);
    std::vector<uint64_t> leaf_seq;
    for (uint32_t v = 0; v < n; ++v)
        if (t.kids[v].empty()) {
            auto it = seq_of.find(t.label[v]);
            if (it == seq_of.end()) { cl_set_error(ctx, "guide tree leaf %s has no sequence", t.label[v].c_str()); return CL_ERR_INVALID_ARGUMENT; }
            slot[v] = leaf_seq.size();
            leaf_seq.push_back(it->second);
        }
    std::vector<uint64_t> merges;
    uint64_t next = leaf_seq.size();
    for (uint32_t v : t.small_first_postorder()) {
        if (t.kids[v].empty()) continue;
        if (t.kids[v].size() != 2 || slot[t.kids[v][0]] == UINT64_MAX || slot[t.kids[v][1]] == UINT64_MAX) {
            cl_set_error(ctx, "Attempting execution with a tree that is not binary");
            return CL_ERR_INVALID_ARGUMENT;
        }
        merges.push_back(slot[t.kids[v].front()]);
        merges.push_back(slot[t.kids[v].back()]);
        slot[v] = next++;
    }
    out->n_leaves = leaf_seq.size();
    out->n_merges = merges.size() / 2;
    out->leaf_sequence = (uint64_t*)malloc((leaf_seq.size() ? leaf_seq.size() : 1) * sizeof(uint64_t));
    out->merge_children = (uint64_t*)malloc((merges.size() ? merges.size() : 1) * sizeof(uint64_t));
    if (!out->leaf_sequence || !out->merge_children) { cl_msa_plan_free(out); return CL_ERR_OUT_OF_MEMORY; }
    memcpy(out->leaf_sequence, leaf_seq.data(), leaf_seq.size() * sizeof(uint64_t));
    if (!merges.empty()) memcpy(out->merge_children, merges.data(), merges.size() * sizeof(uint64_t));
    return CL_OK;
}

void cl_msa_params_default(cl_msa_params* p) {
    memset(p, 0, sizeof(*p));
    cl_merge_params_default(&p->merge);
    p->merge.align.anchor.score_scale = 0.303092;   // ScoreFunction::score_scale as the CLI starts with (include/centrolign/score_function.hpp:39): what a run with skip_calibration keeps
    p->skip_calibration = 0;
    p->cyclize = 0;
    p->max_tandem_duplication_search_rounds = 3;   // src/parameters.cpp:90
    cl_bond_params_default(&p->bonds);
    cl_polish_params_default(&p->polish);
}

// main() of the reference from the parsed inputs on (src/main.cpp:239-301): plan, leaf graphs, calibration (src/core.cpp:98-191 without
// cyclisation: score_scale = mean of the leaves' intrinsic scales in leaf order), one cl_merge per tree node in execution order, then
// explicit_cigar for two sequences, write_gfa otherwise.  Everything numerical happens behind the seams this function calls.
int cl_msa(cl_context* ctx, const char* fasta_text, uint64_t fasta_len, const char* newick, const cl_msa_params* params, char** text_out,
           uint64_t* len_out, cl_msa_stats* stats) {
    cl_bind_device(ctx);
    if (!ctx || !fasta_text || !params || !text_out || !len_out) { cl_set_error(ctx, "null argument"); return CL_ERR_INVALID_ARGUMENT; }
    *text_out = nullptr;
    *len_out = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    const auto t_all = now();
    cl_fasta fa;
    int rc = cl_parse_fasta(ctx, fasta_text, fasta_len, &fa);
    if (rc) return rc;
    if (fa.n_sequences < 2) { cl_fasta_free(&fa); cl_set_error(ctx, "FASTA input contains %llu sequence(s), cannot form an alignment", (unsigned long long)fa.n_sequences); return CL_ERR_INVALID_ARGUMENT; }
    cl_msa_plan plan;
    rc = cl_msa_plan_create(ctx, newick, fa.names, fa.n_sequences, &plan);
    if (rc) { cl_fasta_free(&fa); return rc; }
    const uint64_t n_slots = plan.n_leaves + plan.n_merges;
    std::vector<cl_owned_base_graph*> graph(n_slots, nullptr);
    std::vector<std::vector<uint64_t>> paths(n_slots);   // sequence indices of a slot's paths, in path order
    cl_alignment root_aln{};
    uint64_t root_children[2] = {0, 0};
    cl_msa_stats st{};
    std::vector<std::pair<uint64_t, cl_alignment>> bond_alns;   // -c: (sequence, bond alignment in path positions), in the reference's order
    auto free_bonds = [&]() { for (auto& b : bond_alns) free(b.second.pairs); bond_alns.clear(); };
    auto fail = [&](int code) {
        free_bonds();
        for (auto* g : graph) cl_owned_base_graph_free(g);
        cl_alignment_free(&root_aln);
        cl_msa_plan_free(&plan);
        cl_fasta_free(&fa);
        return code;
    };
    {   // the leaf graphs, side by side (25 ms each at 1 Mbp)
        std::vector<int> leaf_rc(plan.n_leaves, CL_OK);
        cl_parallel_for(plan.n_leaves, [&](uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                const uint64_t s = plan.leaf_sequence[i];
                leaf_rc[i] = cl_leaf_graph(fa.sequences[s], fa.lengths[s], &graph[i]);
            }
        }, 1);
        for (uint64_t i = 0; i < plan.n_leaves; ++i) {
            const uint64_t s = plan.leaf_sequence[i];
            if ((rc = leaf_rc[i])) { cl_set_error(ctx, "sequence %s cannot be made into a graph", fa.names[s]); return fail(rc); }
            paths[i].assign(1, s);
        }
    }
    cl_merge_params mp = params->merge;
    // worker contexts (n_workers > 1): the leaf calibrations, and every merge whose two children are there, run side by side, one thread per
    // context — the merges of a guide tree that do not depend on one another (src/execution.cpp:98-124 hands them out one at a time)
    const int n_workers = std::max(1, std::min(params->n_workers, 16));
    std::vector<cl_context*> workers(1, ctx);
    for (int w = 1; w < n_workers; ++w) {
        cl_context* c = cl_context_create(params->devices && params->n_devices > 0 ? params->devices[w % params->n_devices] : ctx->device);
        if (!c) break;
        workers.push_back(c);
    }
    struct WorkersGuard { std::vector<cl_context*>& w; ~WorkersGuard() { for (size_t i = 1; i < w.size(); ++i) cl_context_destroy(w[i]); } } workers_guard{workers};
    auto run_workers = [&](const std::function<void(cl_context*)>& body) {
        std::vector<std::thread> th;
        for (size_t w = 1; w < workers.size(); ++w) th.emplace_back(body, workers[w]);
        body(workers[0]);
        for (auto& t : th) t.join();
    };
    // -c (src/core.cpp:63-75): the tandem-duplication bonds come from the calibration pass, or — on a restart — from PREFIX_bonds.txt
    const bool cyclize = params->cyclize != 0;
    const std::string bonds_file = (params->subproblems_prefix ? std::string(params->subproblems_prefix) : std::string()) + "_bonds.txt";
    bool bonds_restarted = false;
    if (cyclize && params->restart && params->subproblems_prefix && *params->subproblems_prefix) {   // Core::restart_bonds (:491-521)
        std::ifstream in(bonds_file);
        if (!in) { cl_set_error(ctx, "Couldn't open tandem duplication bonds to restart from file %s.", bonds_file.c_str()); return fail(CL_ERR_INVALID_ARGUMENT); }
        std::string line;
        std::vector<uint64_t> cur;
        auto close_current = [&]() {
            if (bond_alns.empty()) return;
            cl_alignment& a = bond_alns.back().second;
            a.n_pairs = cur.size() / 2;
            a.pairs = (uint64_t*)malloc((cur.size() ? cur.size() : 1) * sizeof(uint64_t));
            if (!cur.empty()) memcpy(a.pairs, cur.data(), cur.size() * sizeof(uint64_t));
            cur.clear();
        };
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            if (line.front() == '#') {
                close_current();
                const std::string name = line.substr(1);
                uint64_t sq = 0;
                while (sq < fa.n_sequences && name != fa.names[sq]) ++sq;
                if (sq == fa.n_sequences) { free_bonds(); cl_set_error(ctx, "%s names a sequence that is not in the input: %s", bonds_file.c_str(), name.c_str()); return fail(CL_ERR_INVALID_ARGUMENT); }
                bond_alns.emplace_back(sq, cl_alignment{});
            } else if (!bond_alns.empty()) {
                long long x = 0, y = 0;
                if (sscanf(line.c_str(), "%lld\t%lld", &x, &y) == 2) { cur.push_back((uint64_t)(int64_t)x); cur.push_back((uint64_t)(int64_t)y); }
            }
        }
        close_current();
        bonds_restarted = true;
    }
    if (!params->skip_calibration || (cyclize && !bonds_restarted)) {
        const auto t = now();
        const bool search_bonds = cyclize && !bonds_restarted;
        std::vector<double> scales(plan.n_leaves, 0.0);
        std::vector<cl_leaf_calibration*> memo(plan.n_leaves, nullptr);
        auto free_memo = [&]() { for (auto*& m : memo) { cl_leaf_calibration_free(m); m = nullptr; } };
        {
            std::atomic<uint64_t> next{0};
            std::atomic<int> first_rc{CL_OK};
            run_workers([&](cl_context* c) {
                for (uint64_t i = next.fetch_add(1); i < plan.n_leaves && first_rc.load() == CL_OK; i = next.fetch_add(1)) {
                    cl_base_graph v;
                    cl_owned_base_graph_view(graph[i], &v);
                    const int r = search_bonds ? cl_leaf_calibrate(c, &v, &mp.match, &mp.align.anchor, &scales[i], &memo[i])
                                               : cl_leaf_intrinsic_scale(c, &v, &mp.match, &mp.align.anchor, &scales[i]);
                    if (r) { int expected = CL_OK; if (first_rc.compare_exchange_strong(expected, r) && c != ctx) cl_set_error(ctx, "%s", cl_last_error(c)); }
                }
            });
            if ((rc = first_rc.load())) { free_memo(); return fail(rc); }
        }
        if (!params->skip_calibration) {
            double mean = 0.0;
            for (double sc : scales) mean += sc;   // summed in leaf order (src/core.cpp:169-173)
            mean /= (double)plan.n_leaves;
            mp.align.anchor.score_scale = mean;
        }
        st.calibration_s = secs(t);
        if (search_bonds) {   // the tandem-duplication rounds (:196-297), leaf by leaf in the reference, leaves side by side here
            const auto tb = now();
            std::vector<cl_alignment_list> found(plan.n_leaves);
            for (auto& f : found) memset(&f, 0, sizeof(f));
            std::atomic<uint64_t> next{0};
            std::atomic<int> first_rc{CL_OK};
            run_workers([&](cl_context* c) {
                for (uint64_t i = next.fetch_add(1); i < plan.n_leaves && first_rc.load() == CL_OK; i = next.fetch_add(1)) {
                    cl_base_graph v;
                    cl_owned_base_graph_view(graph[i], &v);
                    const int r = cl_leaf_bond_alignments(c, &v, memo[i], &mp.align.anchor, &mp.align.stitch, &params->bonds, params->max_tandem_duplication_search_rounds, &found[i]);
                    if (r) { int expected = CL_OK; if (first_rc.compare_exchange_strong(expected, r) && c != ctx) cl_set_error(ctx, "%s", cl_last_error(c)); }
                }
            });
            free_memo();
            for (uint64_t i = 0; i < plan.n_leaves; ++i) {
                for (uint64_t b = 0; b < found[i].n; ++b) { bond_alns.emplace_back(plan.leaf_sequence[i], found[i].alignments[b]); found[i].alignments[b].pairs = nullptr; }
                cl_alignment_list_free(&found[i]);
            }
            if ((rc = first_rc.load())) { free_bonds(); return fail(rc); }
            st.bonds_s = secs(tb);
        }
    }
    if (cyclize && !bonds_restarted && params->subproblems_prefix && *params->subproblems_prefix) {   // Core::emit_restart_bonds (:476-489)
        std::ofstream bo(bonds_file);
        if (!bo) { free_bonds(); cl_set_error(ctx, "Couldn't write subproblem bonds to file '%s'.", bonds_file.c_str()); return fail(CL_ERR_INVALID_ARGUMENT); }
        for (const auto& b : bond_alns) {
            bo << '#' << fa.names[b.first] << '\n';
            for (uint64_t i = 0; i < b.second.n_pairs; ++i) bo << (long long)(int64_t)b.second.pairs[2 * i] << '\t' << (long long)(int64_t)b.second.pairs[2 * i + 1] << '\n';
        }
    }
    st.n_bonds = bond_alns.size();
    st.score_scale = mp.align.anchor.score_scale;
    // -S / -R (src/core.cpp:370-422, src/execution.cpp:190-203, 222-277)
    const std::string prefix = params->subproblems_prefix ? params->subproblems_prefix : "";
    std::vector<std::vector<uint64_t>> leaf_set(n_slots);   // the sequences under every slot
    for (uint64_t i = 0; i < plan.n_leaves; ++i) leaf_set[i].assign(1, plan.leaf_sequence[i]);
    for (uint64_t k = 0; k < plan.n_merges; ++k) {
        auto& dst = leaf_set[plan.n_leaves + k];
        dst = leaf_set[plan.merge_children[2 * k]];
        dst.insert(dst.end(), leaf_set[plan.merge_children[2 * k + 1]].begin(), leaf_set[plan.merge_children[2 * k + 1]].end());
    }
    auto file_of = [&](uint64_t slot) {
        std::vector<const char*> nm;
        for (uint64_t s : leaf_set[slot]) nm.push_back(fa.names[s]);
        char hex[17];
        (void)cl_subproblem_hash_hex(nm.data(), nm.size(), hex);
        return prefix + "_" + hex + ".gfa";
    };
    std::vector<char> done(n_slots, 0);   // loaded, or below something loaded
    if (params->restart && !prefix.empty()) {
        for (uint64_t k = plan.n_merges; k-- > 0;) {   // the root first: what is loaded makes everything below it unnecessary
            const uint64_t slot = plan.n_leaves + k;
            if (!done[slot]) {
                std::ifstream in(file_of(slot), std::ios::binary);
                if (in) {
                    const std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
                    char** nm = nullptr;
                    uint64_t np = 0;
                    if ((rc = cl_read_gfa(text.data(), text.size(), 1, &graph[slot], &nm, &np))) { cl_set_error(ctx, "cannot read %s", file_of(slot).c_str()); return fail(rc); }
                    for (uint64_t p = 0; p < np; ++p) {
                        uint64_t s = 0;
                        while (s < fa.n_sequences && strcmp(fa.names[s], nm[p]) != 0) ++s;
                        if (s < fa.n_sequences) paths[slot].push_back(s);
                        free(nm[p]);
                    }
                    free(nm);
                    if (paths[slot].size() != np) { cl_set_error(ctx, "%s names a path that is not an input sequence", file_of(slot).c_str()); return fail(CL_ERR_INVALID_ARGUMENT); }
                    done[slot] = 1;
                    ++st.n_restarted;
                }
            }
            if (done[slot]) {   // (loaded here or under a loaded ancestor): its children are not needed
                for (int c = 0; c < 2; ++c) {
                    const uint64_t ch = plan.merge_children[2 * k + c];
                    if (ch >= plan.n_leaves) done[ch] = done[ch] ? done[ch] : 2;
                }
            }
        }
    }
    std::mutex state_mutex;
    auto do_merge = [&](uint64_t k, cl_context* c) -> int {
        const uint64_t a = plan.merge_children[2 * k], b = plan.merge_children[2 * k + 1];
        cl_base_graph g1, g2;
        cl_owned_base_graph_view(graph[a], &g1);
        cl_owned_base_graph_view(graph[b], &g2);
        cl_merge_result r;
        int rc = cl_merge(c, &g1, &g2, &mp, &r);
        if (rc) { if (c != ctx) cl_set_error(ctx, "%s", cl_last_error(c)); return rc; }
        std::lock_guard<std::mutex> lock(state_mutex);
        st.match_s += r.match_ms * 1e-3;
        st.align_s += r.align_ms * 1e-3;
        st.fuse_s += r.fuse_ms * 1e-3;
        const uint64_t slot = plan.n_leaves + k;
        graph[slot] = r.fused;
        r.fused = nullptr;
        paths[slot] = paths[a];
        paths[slot].insert(paths[slot].end(), paths[b].begin(), paths[b].end());
        if (k + 1 == plan.n_merges) {
            root_aln = r.align.alignment;
            r.align.alignment.pairs = nullptr;
            r.align.alignment.n_pairs = 0;
            root_children[0] = a; root_children[1] = b;
        } else if (fa.n_sequences > 2) {   // a child is not needed again (two sequences: the CIGAR needs the leaves)
            cl_owned_base_graph_free(graph[a]); graph[a] = nullptr;
            cl_owned_base_graph_free(graph[b]); graph[b] = nullptr;
        }
        cl_merge_result_free(&r);
        ++st.n_merges;
        if (!prefix.empty()) {   // Core::emit_subproblem
            const std::string info_name = prefix + "_info.txt", gfa_name = file_of(slot);
            const bool header = !std::ifstream(info_name);
            std::ofstream info(info_name, std::ios::app), gfa(gfa_name, std::ios::binary);
            if (!info || !gfa) { cl_set_error(ctx, "Failed to write to subproblem file %s", gfa_name.c_str()); return (int)CL_ERR_INVALID_ARGUMENT; }
            if (header) info << "filename\tsequences\n";
            std::vector<std::string> sorted;
            for (uint64_t s : leaf_set[slot]) sorted.push_back(fa.names[s]);
            std::sort(sorted.begin(), sorted.end());
            info << gfa_name << '\t';
            for (size_t i = 0; i < sorted.size(); ++i) info << (i ? "," : "") << sorted[i];
            info << '\n';
            std::vector<const char*> nm;
            for (uint64_t s : paths[slot]) nm.push_back(fa.names[s]);
            cl_base_graph g;
            cl_owned_base_graph_view(graph[slot], &g);
            char* text = nullptr;
            uint64_t len = 0;
            if ((rc = cl_write_gfa(&g, nm.data(), 1, &text, &len))) return rc;
            gfa.write(text, (std::streamsize)len);
            free(text);
        }
            return CL_OK;
    };
    {
        // ready queue: a merge is ready when both children are there (leaves, loaded subproblems, finished merges)
        std::mutex qm;
        std::condition_variable qcv;
        std::vector<char> have(n_slots, 0), queued(plan.n_merges, 0);
        std::deque<uint64_t> ready;
        uint64_t remaining = 0;
        int first_rc = CL_OK;
        for (uint64_t i = 0; i < plan.n_leaves; ++i) have[i] = 1;
        for (uint64_t k = 0; k < plan.n_merges; ++k) { if (done[plan.n_leaves + k] == 1) have[plan.n_leaves + k] = 1; if (!done[plan.n_leaves + k]) ++remaining; }
        auto enqueue_ready = [&]() {   // under qm
            for (uint64_t k = 0; k < plan.n_merges; ++k)
                if (!queued[k] && !done[plan.n_leaves + k] && have[plan.merge_children[2 * k]] && have[plan.merge_children[2 * k + 1]]) { queued[k] = 1; ready.push_back(k); }
        };
        enqueue_ready();
        run_workers([&](cl_context* c) {
            while (true) {
                uint64_t k;
                {
                    std::unique_lock<std::mutex> lock(qm);
                    qcv.wait(lock, [&] { return !ready.empty() || remaining == 0 || first_rc != CL_OK; });
                    if (ready.empty()) return;
                    k = ready.front();
                    ready.pop_front();
                }
                const int r = do_merge(k, c);
                {
                    std::lock_guard<std::mutex> lock(qm);
                    if (r && first_rc == CL_OK) first_rc = r;
                    have[plan.n_leaves + k] = 1;
                    --remaining;
                    if (first_rc == CL_OK) enqueue_ready(); else { ready.clear(); remaining = 0; }
                }
                qcv.notify_all();
            }
        });
        if (first_rc) return fail(first_rc);
    }
    const uint64_t root = n_slots - 1;
    if (cyclize && !bond_alns.empty()) {   // Core::apply_bonds (src/core.cpp:594-648)
        const auto tc = now();
        cl_base_graph g;
        cl_owned_base_graph_view(graph[root], &g);
        std::vector<uint64_t> path_of;
        std::vector<cl_alignment> alns;
        for (const auto& b : bond_alns) {
            uint64_t p = 0;
            while (p < paths[root].size() && paths[root][p] != b.first) ++p;
            path_of.push_back(p);
            alns.push_back(b.second);
        }
        cl_owned_base_graph* cyc = nullptr;
        rc = cl_apply_bonds(&g, alns.size(), path_of.data(), alns.data(), &cyc);
        free_bonds();
        if (rc) { cl_set_error(ctx, "merging the tandem duplications failed"); return fail(rc); }
        cl_owned_base_graph_free(graph[root]);
        graph[root] = cyc;
        cl_alignment_free(&root_aln);   // (root_subproblem.alignment.clear(), :645)
        cl_owned_base_graph_view(graph[root], &g);
        std::vector<const char*> pn, sn;
        for (uint64_t sq : paths[root]) pn.push_back(fa.names[sq]);
        for (uint64_t sq = 0; sq < fa.n_sequences; ++sq) sn.push_back(fa.names[sq]);
        cl_owned_base_graph* polished = nullptr;
        uint64_t n_regions = 0;
        rc = cl_polish_cyclized_graph_workers(workers.data(), (unsigned)workers.size(), &g, pn.data(), newick, sn.data(), sn.size(), &mp, &params->polish, &polished, &n_regions);
        if (rc) return fail(rc);
        cl_owned_base_graph_free(graph[root]);
        graph[root] = polished;
        st.n_polished_regions = n_regions;
        st.cyclize_s = secs(tc);
    }
    free_bonds();
    if (fa.n_sequences == 2) {
        // explicit_cigar(root.alignment, leaf of the FIRST sequence, leaf of the LAST one) (src/main.cpp:292-296)
        uint64_t first = 0, last = 0;
        for (uint64_t i = 0; i < plan.n_leaves; ++i) { if (plan.leaf_sequence[i] == 0) first = i; if (plan.leaf_sequence[i] == 1) last = i; }
        cl_base_graph g1, g2;
        cl_owned_base_graph_view(graph[first], &g1);
        cl_owned_base_graph_view(graph[last], &g2);
        (void)root_children;
        uint64_t n_text = 0;
        rc = cl_explicit_cigar(&g1, &g2, root_aln.pairs, root_aln.n_pairs, text_out, &n_text);
        if (rc == CL_OK) {   // main() prints the CIGAR and a line end (src/main.cpp:295): the text handed back is what the CLI writes to its standard output
            char* with_nl = (char*)realloc(*text_out, n_text + 2);
            if (!with_nl) { free(*text_out); *text_out = nullptr; rc = CL_ERR_OUT_OF_MEMORY; }
            else { with_nl[n_text] = '\n'; with_nl[n_text + 1] = '\0'; *text_out = with_nl; if (len_out) *len_out = n_text + 1; }
        }
    } else {
        std::vector<const char*> names;
        for (uint64_t s : paths[root]) names.push_back(fa.names[s]);
        cl_base_graph g;
        cl_owned_base_graph_view(graph[root], &g);
        st.root_nodes = g.n_nodes;
        rc = cl_write_gfa(&g, names.data(), 1, text_out, len_out);
    }
    if (rc) { cl_set_error(ctx, "writing the output failed"); return fail(rc); }
    if (params->induced_pairwise_prefix && *params->induced_pairwise_prefix) {   // Core::output_pairwise_alignments(false)
        cl_base_graph g;
        cl_owned_base_graph_view(graph[root], &g);
        for (uint64_t p1 = 0; p1 < g.n_paths; ++p1)
            for (uint64_t p2 = p1 + 1; p2 < g.n_paths; ++p2) {
                std::string n1 = fa.names[paths[root][p1]], n2 = fa.names[paths[root][p2]];
                std::replace(n1.begin(), n1.end(), '/', '_');
                std::replace(n2.begin(), n2.end(), '/', '_');
                const std::string name = std::string(params->induced_pairwise_prefix) + "_" + n1 + "_" + n2 + ".txt";
                char* text = nullptr;
                if ((rc = cl_induced_pairwise_cigar(&g, p1, p2, &text, nullptr))) { cl_set_error(ctx, "induced pairwise alignment of %s and %s failed", n1.c_str(), n2.c_str()); return fail(rc); }
                std::ofstream out(name);
                if (!out) { free(text); cl_set_error(ctx, "could not write to induced pairwise alignment file %s", name.c_str()); return fail(CL_ERR_INVALID_ARGUMENT); }
                out << text << '\n';
                free(text);
            }
    }
    st.total_s = secs(t_all);
    if (stats) *stats = st;
    fail(CL_OK);
    return CL_OK;
}

}  // extern "C"
