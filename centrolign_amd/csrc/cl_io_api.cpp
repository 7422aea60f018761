// cl_io_api.cpp — the data formats either side of a merge: the leaf graph of a sequence (make_base_graph + add_sentinels,
// src/modify_graph.cpp:30-77), the explicit CIGAR of a pairwise alignment (include/centrolign/alignment.hpp:2804-2843) and the
// GFA of a subproblem graph (include/centrolign/gfa.hpp:46-157).  Host code; byte-identical text.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cl_internal.hpp"

namespace {

// encode_base (src/utility.cpp:324-345): A C G T N in either case -> 0..4, anything else -> 5
uint8_t encode_base(char c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'N': case 'n': return 4;
    default: return 5;
    }
}

char* to_c_string(const std::string& s, uint64_t* len_out) {
    char* p = (char*)malloc(s.size() + 1);
    if (!p) return nullptr;
    memcpy(p, s.data(), s.size());
    p[s.size()] = '\0';
    if (len_out) *len_out = s.size();
    return p;
}

}  // namespace

extern "C" {

int cl_leaf_graph(const char* sequence, uint64_t n, cl_owned_base_graph** out) {
    if (!sequence || n == 0 || n >= 0xFFFFFFF0ull || !out) return CL_ERR_INVALID_ARGUMENT;   // make_base_graph asserts !sequence.empty()
    std::unique_ptr<cl_owned_base_graph> g(new cl_owned_base_graph());
    const uint32_t src = (uint32_t)n, snk = (uint32_t)n + 1;
    g->label.resize(n + 2);
    for (uint64_t i = 0; i < n; ++i) g->label[i] = encode_base(sequence[i]);
    g->label[src] = 5;   // add_sentinels(graph, 5, 6) (src/execution.cpp:70)
    g->label[snk] = 6;
    g->next_off.resize(n + 3);
    g->prev_off.resize(n + 3);
    g->next_idx.resize(n + 1);
    g->prev_idx.resize(n + 1);
    for (uint64_t i = 0; i < n; ++i) {
        g->next_off[i] = i;
        g->next_idx[i] = i + 1 < n ? (uint32_t)(i + 1) : snk;
        g->prev_off[i] = i;
        g->prev_idx[i] = i ? (uint32_t)(i - 1) : src;
    }
    g->next_off[src] = n; g->next_idx[n] = 0;          // source -> first base
    g->next_off[snk] = n + 1; g->next_off[n + 2] = n + 1;
    g->prev_off[src] = n; g->prev_off[snk] = n;        // the source has no predecessor
    g->prev_idx[n] = (uint32_t)(n - 1);                // last base -> sink
    g->prev_off[n + 2] = n + 1;
    g->path_off = {0, n};
    g->path_nodes.resize(n);
    for (uint64_t i = 0; i < n; ++i) g->path_nodes[i] = (uint32_t)i;
    g->src_id = src;
    g->snk_id = snk;
    *out = g.release();
    return CL_OK;
}

int cl_explicit_cigar(const cl_base_graph* g1, const cl_base_graph* g2, const uint64_t* pairs, uint64_t n_pairs, char** text_out, uint64_t* len_out) {
    if (!g1 || !g2 || (n_pairs && !pairs) || !text_out) return CL_ERR_INVALID_ARGUMENT;
    const uint64_t gap = ~(uint64_t)0;
    std::string out;
    int curr_len = 0;   // an int in the reference too
    char curr_op = '\0';
    for (uint64_t i = 0; i < n_pairs; ++i) {
        const uint64_t a = pairs[2 * i], b = pairs[2 * i + 1];
        if ((a != gap && a >= g1->n_nodes) || (b != gap && b >= g2->n_nodes)) return CL_ERR_INVALID_ARGUMENT;
        const char op = a == gap ? 'I' : b == gap ? 'D' : g1->label[a] == g2->label[b] ? '=' : 'X';
        if (op == curr_op) { ++curr_len; continue; }
        if (curr_len != 0) { out += std::to_string(curr_len); out += curr_op; }
        curr_len = 1;
        curr_op = op;
    }
    if (curr_len != 0) { out += std::to_string(curr_len); out += curr_op; }
    *text_out = to_c_string(out, len_out);
    return *text_out ? CL_OK : CL_ERR_OUT_OF_MEMORY;
}

int cl_write_gfa(const cl_base_graph* g, const char* const* path_names, int decode, char** text_out, uint64_t* len_out) {
    if (!g || !text_out || (g->n_paths && !path_names)) return CL_ERR_INVALID_ARGUMENT;
    const uint64_t n = g->n_nodes;
    auto sentinel = [&](uint64_t v) { return v == g->src_id || v == g->snk_id; };
    auto next_size = [&](uint64_t v) { return g->next_off[v + 1] - g->next_off[v]; };
    auto prev_size = [&](uint64_t v) { return g->prev_off[v + 1] - g->prev_off[v]; };
    auto first_next = [&](uint64_t v) { return (uint64_t)g->next_idx[g->next_off[v]]; };
    auto first_prev = [&](uint64_t v) { return (uint64_t)g->prev_idx[g->prev_off[v]]; };
    std::vector<uint8_t> path_begin(n, 0), path_end(n, 0), compacted_end(n, 0);
    std::vector<uint64_t> compacted_id(n, ~(uint64_t)0);
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        if (g->path_off[p + 1] == g->path_off[p]) return CL_ERR_INVALID_ARGUMENT;   // the reference reads front() / back()
        path_begin[g->path_nodes[g->path_off[p]]] = 1;
        path_end[g->path_nodes[g->path_off[p + 1] - 1]] = 1;
    }
    static const char dec[] = {'A', 'C', 'G', 'T', 'N', '\0'};   // src/utility.cpp:355
    std::string out = "H\tVN:Z:1.0\n";
    // segments: maximal non-branching runs that no path begins or ends inside (gfa.hpp:66-121)
    uint64_t next_compacted_id = 1;
    std::vector<uint64_t> run;
    for (uint64_t v = 0; v < n; ++v) {
        if (compacted_id[v] != ~(uint64_t)0 || sentinel(v)) continue;
        run.assign(1, v);
        while (!path_begin[run.back()] && prev_size(run.back()) == 1 && !path_end[first_prev(run.back())] &&
               next_size(first_prev(run.back())) == 1 && !sentinel(first_prev(run.back())))
            run.push_back(first_prev(run.back()));
        std::reverse(run.begin(), run.end());
        while (!path_end[run.back()] && next_size(run.back()) == 1 && !path_begin[first_next(run.back())] &&
               prev_size(first_next(run.back())) == 1 && !sentinel(first_next(run.back())))
            run.push_back(first_next(run.back()));
        out += "S\t";
        out += std::to_string(next_compacted_id);
        out += '\t';
        for (uint64_t u : run) {
            compacted_id[u] = next_compacted_id;
            out += decode ? dec[g->label[u] < 5 ? g->label[u] : 5] : (char)g->label[u];
        }
        out += '\n';
        ++next_compacted_id;
        compacted_end[run.back()] = 1;
    }
    // links (:123-136)
    for (uint64_t v = 0; v < n; ++v) {
        if (!compacted_end[v] || sentinel(v)) continue;
        for (uint64_t e = g->next_off[v]; e < g->next_off[v + 1]; ++e) {
            const uint64_t w = g->next_idx[e];
            if (sentinel(w)) continue;
            out += "L\t";
            out += std::to_string(compacted_id[v]);
            out += "\t+\t";
            out += std::to_string(compacted_id[w]);
            out += "\t+\t*\n";
        }
    }
    // paths (:138-156)
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        out += "P\t";
        out += path_names[p];
        out += '\t';
        bool write_next = true, first = true;
        for (uint64_t i = g->path_off[p]; i < g->path_off[p + 1]; ++i) {
            const uint64_t v = g->path_nodes[i];
            if (sentinel(v)) continue;
            if (write_next) {
                if (!first) out += ',';
                out += std::to_string(compacted_id[v]);
                out += '+';
                first = false;
            }
            write_next = compacted_end[v];
        }
        out += "\t*\n";
    }
    *text_out = to_c_string(out, len_out);
    return *text_out ? CL_OK : CL_ERR_OUT_OF_MEMORY;
}

// read_gfa(in, encode = true) + add_sentinels(graph, 5, 6) (src/gfa.cpp:9-96, src/modify_graph.cpp:47-77): how Execution::restart loads a
// subproblem written by -S (src/execution.cpp:240-251).  Node ids follow the S lines (a compacted node becomes a chain), adjacency lists the
// order of insertion: chain edges as the S lines are read, L lines in file order, then per node the sink edge before the source edge.
int cl_read_gfa(const char* text, uint64_t len, int add_sentinels, cl_owned_base_graph** out, char*** path_names_out, uint64_t* n_paths_out) {
    if (!text || !out) return CL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (path_names_out) *path_names_out = nullptr;
    if (n_paths_out) *n_paths_out = 0;
    std::vector<uint8_t> label;
    std::vector<std::vector<uint32_t>> next, prev;
    std::vector<std::pair<int64_t, int64_t>> span;   // GFA id -> (first node, last node)
    std::vector<std::string> names;
    std::vector<std::vector<uint32_t>> paths;
    auto add_node = [&](uint8_t l) { label.push_back(l); next.emplace_back(); prev.emplace_back(); return (uint32_t)(label.size() - 1); };
    auto add_edge = [&](uint32_t a, uint32_t b) { next[a].push_back(b); prev[b].push_back(a); };
    auto tokens_of = [](const std::string& line, char delim) {
        std::vector<std::string> t;
        size_t at = 0;
        while (true) {
            const size_t e = line.find(delim, at);
            if (e == std::string::npos) { t.push_back(line.substr(at)); break; }
            t.push_back(line.substr(at, e - at));
            at = e + 1;
        }
        return t;
    };
    auto parse_id = [](const std::string& s, int64_t& v) {
        if (s.empty()) return false;
        char* end = nullptr;
        v = strtoll(s.c_str(), &end, 10);
        return end && *end == '\0' && v >= 0;
    };
    uint64_t at = 0;
    while (at < len) {
        const char* nl = (const char*)memchr(text + at, '\n', len - at);
        const uint64_t e = nl ? (uint64_t)(nl - text) : len;
        const std::string line(text + at, text + e);
        at = e + 1;
        if (line.empty()) continue;
        const std::vector<std::string> tk = tokens_of(line, '\t');
        if (tk[0] == "S") {
            int64_t id;
            if (tk.size() != 3 || tk[2].empty() || !parse_id(tk[1], id)) return CL_ERR_INVALID_ARGUMENT;
            while ((int64_t)span.size() <= id) span.emplace_back(-1, -1);
            uint32_t node = add_node(encode_base(tk[2][0]));
            span[id] = {node, node};
            for (size_t i = 1; i < tk[2].size(); ++i) {
                const uint32_t nx = add_node(encode_base(tk[2][i]));
                span[id].second = nx;
                add_edge(node, nx);
                node = nx;
            }
        } else if (tk[0] == "L") {
            if (tk.size() != 6 || (tk[5] != "*" && tk[5] != "0M") || tk[2] != tk[4]) return CL_ERR_INVALID_ARGUMENT;
            int64_t a, b;
            if (!parse_id(tk[1], a) || !parse_id(tk[3], b)) return CL_ERR_INVALID_ARGUMENT;
            if (tk[2] == "-") std::swap(a, b);
            if (a >= (int64_t)span.size() || b >= (int64_t)span.size() || span[a].first < 0 || span[b].first < 0) return CL_ERR_INVALID_ARGUMENT;
            add_edge((uint32_t)span[a].second, (uint32_t)span[b].first);
        } else if (tk[0] == "P") {
            if (tk.size() != 4 || tk[3] != "*") return CL_ERR_INVALID_ARGUMENT;
            names.push_back(tk[1]);
            paths.emplace_back();
            for (const std::string& step : tokens_of(tk[2], ',')) {
                int64_t id;
                if (step.empty() || step.back() != '+' || !parse_id(step.substr(0, step.size() - 1), id) || id >= (int64_t)span.size() || span[id].first < 0)
                    return CL_ERR_INVALID_ARGUMENT;
                uint32_t node = (uint32_t)span[id].first;
                paths.back().push_back(node);
                while (node != (uint32_t)span[id].second) { node = next[node].front(); paths.back().push_back(node); }
            }
        }
    }
    uint64_t src = 0, snk = 0;
    if (add_sentinels) {
        std::vector<char> begins(label.size(), 0), ends(label.size(), 0);
        for (const auto& pth : paths) if (!pth.empty()) { begins[pth.front()] = 1; ends[pth.back()] = 1; }
        const uint32_t n0 = (uint32_t)label.size();
        src = add_node(5);
        snk = add_node(6);
        if (n0 == 0) add_edge((uint32_t)src, (uint32_t)snk);
        else
            for (uint32_t v = 0; v < n0; ++v) {
                const bool no_next = next[v].empty(), no_prev = prev[v].empty();   // (as they were before this node's own sentinel edges)
                if (no_next || ends[v]) add_edge(v, (uint32_t)snk);
                if (no_prev || begins[v]) add_edge((uint32_t)src, v);
            }
    }
    std::unique_ptr<cl_owned_base_graph> g(new cl_owned_base_graph());
    g->label = label;
    g->next_off.assign(1, 0);
    g->prev_off.assign(1, 0);
    for (size_t v = 0; v < label.size(); ++v) {
        g->next_idx.insert(g->next_idx.end(), next[v].begin(), next[v].end());
        g->next_off.push_back(g->next_idx.size());
        g->prev_idx.insert(g->prev_idx.end(), prev[v].begin(), prev[v].end());
        g->prev_off.push_back(g->prev_idx.size());
    }
    g->path_off.assign(1, 0);
    for (const auto& pth : paths) { g->path_nodes.insert(g->path_nodes.end(), pth.begin(), pth.end()); g->path_off.push_back(g->path_nodes.size()); }
    g->src_id = src;
    g->snk_id = snk;
    if (path_names_out) {
        char** arr = (char**)calloc(names.size() ? names.size() : 1, sizeof(char*));
        if (!arr) return CL_ERR_OUT_OF_MEMORY;
        for (size_t i = 0; i < names.size(); ++i) arr[i] = to_c_string(names[i], nullptr);
        *path_names_out = arr;
    }
    if (n_paths_out) *n_paths_out = names.size();
    *out = g.release();
    return CL_OK;
}

// Execution::subproblem_hash (src/execution.cpp:190-203) as to_hex prints it (include/centrolign/utility.hpp:310-328): the name of a
// subproblem's file under -S / -R is PREFIX + "_" + this + ".gfa" (src/core.cpp:378-380).  hex_out: 17 bytes.
int cl_subproblem_hash_hex(const char* const* sequence_names, uint64_t n, char* hex_out) {
    if ((n && !sequence_names) || !hex_out) return CL_ERR_INVALID_ARGUMENT;
    std::vector<std::string> names(sequence_names, sequence_names + n);
    std::sort(names.begin(), names.end());
    auto combine = [](uint64_t& seed, uint64_t v) { seed ^= v + 0x9e3779b9ull + (seed << 6) + (seed >> 2); };   // hash_combine with libstdc++'s identity std::hash
    uint64_t h = 660422875706093811ull;
    for (const std::string& nm : names) {
        combine(h, 2110260111091729000ull);
        for (char c : nm) combine(h, (uint64_t)(size_t)c);
    }
    static const char digits[] = "0123456789ABCDEF";
    for (int i = 0; i < 16; ++i) hex_out[i] = digits[(h >> (60 - 4 * i)) & 0xF];
    hex_out[16] = '\0';
    return CL_OK;
}

// induced_pairwise_alignment(graph, path1, path2) (src/alignment.cpp:130-229) printed with explicit_cigar(alignment, seq1, seq2) (:84-123):
// the -A output of the CLI for an acyclic graph (src/core.cpp:546-550) — the alignment of two input sequences that the MSA graph implies.
// Positions on the two paths, not node ids; equal-length gap runs of at most 4 on both sides are read as mismatches, other mixed runs as
// one deletion followed by one insertion.
int cl_induced_pairwise_cigar(const cl_base_graph* g, uint64_t path1, uint64_t path2, char** text_out, uint64_t* len_out) {
    if (!g || !text_out || path1 >= g->n_paths || path2 >= g->n_paths) return CL_ERR_INVALID_ARGUMENT;
    const uint64_t gap = ~(uint64_t)0;
    const uint32_t* p1 = g->path_nodes + g->path_off[path1];
    const uint32_t* p2 = g->path_nodes + g->path_off[path2];
    const uint64_t n1 = g->path_off[path1 + 1] - g->path_off[path1], n2 = g->path_off[path2 + 1] - g->path_off[path2];
    std::vector<uint64_t> index_in_path1(g->n_nodes, gap);
    for (uint64_t i = 0; i < n1; ++i) {
        if (p1[i] >= g->n_nodes || index_in_path1[p1[i]] != gap) return CL_ERR_CYCLIC_GRAPH;   // "follows cycles in the graph" (:138-140)
        index_in_path1[p1[i]] = i;
    }
    std::vector<std::pair<uint64_t, uint64_t>> aln;
    uint64_t j = 0;
    for (uint64_t i = 0; i < n2; ++i) {
        if (p2[i] >= g->n_nodes) return CL_ERR_INVALID_ARGUMENT;
        const uint64_t at = index_in_path1[p2[i]];
        if (at == gap) aln.emplace_back(gap, i);
        else {
            while (j < at) aln.emplace_back(j++, gap);
            aln.emplace_back(j++, i);
        }
    }
    while (j < n1) aln.emplace_back(j++, gap);
    // consolidate the gap runs (:170-224)
    const size_t max_mismatch_size = 4;
    size_t removed = 0;
    for (size_t i = 0; i < aln.size();) {
        if (aln[i].first != gap && aln[i].second != gap) { aln[i - removed] = aln[i]; ++i; continue; }
        size_t e = i, gaps1 = 0, gaps2 = 0;
        while (e < aln.size() && (aln[e].first == gap || aln[e].second == gap)) { gaps1 += aln[e].first == gap; gaps2 += aln[e].second == gap; ++e; }
        uint64_t last1 = gap, last2 = gap;   // -1: the run starts the alignment
        if (i != 0) { last1 = aln[i - removed - 1].first; last2 = aln[i - removed - 1].second; }
        if (gaps1 == gaps2 && gaps1 <= max_mismatch_size) {
            for (uint64_t k = 0; k < gaps1; ++k) aln[i - removed + k] = {last1 + k + 1, last2 + k + 1};
            removed += gaps1;
        } else {
            for (uint64_t k = 0; k < gaps2; ++k) aln[i - removed + k] = {last1 + k + 1, gap};
            for (uint64_t k = 0; k < gaps1; ++k) aln[i - removed + k + gaps2] = {gap, last2 + k + 1};
        }
        i = e;
    }
    aln.resize(aln.size() - removed);
    std::string out;
    int curr_len = 0;
    char curr_op = '\0';
    for (const auto& ap : aln) {
        const char op = ap.first == gap ? 'I' : ap.second == gap ? 'D' : g->label[p1[ap.first]] == g->label[p2[ap.second]] ? '=' : 'X';
        if (op == curr_op) { ++curr_len; continue; }
        if (curr_len != 0) { out += std::to_string(curr_len); out += curr_op; }
        curr_len = 1;
        curr_op = op;
    }
    if (curr_len != 0) { out += std::to_string(curr_len); out += curr_op; }
    *text_out = to_c_string(out, len_out);
    return *text_out ? CL_OK : CL_ERR_OUT_OF_MEMORY;
}

}  // extern "C"
