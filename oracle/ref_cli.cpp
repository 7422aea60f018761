/* Test infrastructure only: a command-line front end for the UNMODIFIED compiled reference (oracle/_ref/libcentrolign_ref.so),
 * standing where /root/reference/src/main.cpp:54-315 stands (that file is not compiled because it needs the cmake-generated
 * version.cpp).  It exists so that long reference runs — BASELINE configs[2], 10 x 1 Mbp — can be left running as a separate
 * process under a memory limit, writing every finished subproblem as a GFA file (-S of the CLI: Core::subproblems_prefix,
 * src/core.cpp:370-422); tests/golden/make_c3_digests.py turns those files into the digests the -m gpu suite reproduces.
 *
 *   ref_cli FASTA NEWICK|- PREFIX|- OUT|- [max_num_match_pairs] [verbosity] [restart 0|1] [overrides]
 *   overrides: typed settings of the reference's Parameters, "b:cyclize_tandem_duplications=1;i:min_cyclizing_length=5000;d:name=0.2"
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "centrolign/alignment.hpp"
#include "centrolign/core.hpp"
#include "centrolign/gfa.hpp"
#include "centrolign/logging.hpp"
#include "centrolign/parameters.hpp"
#include "centrolign/utility.hpp"

using namespace centrolign;

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: ref_cli FASTA NEWICK|- PREFIX|- OUT|- [max_num_match_pairs] [verbosity] [restart]\n");
        return 2;
    }
    std::string fasta = argv[1], newick_path = argv[2], prefix = argv[3], out = argv[4];
    long long budget = argc > 5 ? atoll(argv[5]) : 0;
    int verbosity = argc > 6 ? atoi(argv[6]) : 2;
    bool restart = argc > 7 && atoi(argv[7]) != 0;
    try {
        Parameters params;
        params.set<std::string>("fasta_name", fasta);
        if (prefix != "-") params.set<std::string>("subproblems_prefix", prefix);
        if (budget > 0) params.set<int64_t>("max_num_match_pairs", (int64_t)budget);
        if (restart) params.set<bool>("restart", true);
        if (argc > 8) {
            std::stringstream ss(argv[8]);
            std::string item;
            while (std::getline(ss, item, ';')) {
                const size_t eq = item.find('=');
                if (item.size() < 4 || item[1] != ':' || eq == std::string::npos) continue;
                const std::string name = item.substr(2, eq - 2), val = item.substr(eq + 1);
                if (item[0] == 'b') params.set<bool>(name, val != "0");
                else if (item[0] == 'i') params.set<int64_t>(name, (int64_t)atoll(val.c_str()));
                else if (item[0] == 'd') params.set<double>(name, atof(val.c_str()));
                else params.set<std::string>(name, val);
            }
        }
        params.validate();
        logging::level = (logging::LoggingLevel)verbosity;
        std::ifstream fin(fasta);
        if (!fin) { fprintf(stderr, "cannot read %s\n", fasta.c_str()); return 1; }
        auto parsed = parse_fasta(fin);
        std::vector<std::string> names;
        for (const auto& p : parsed) names.push_back(p.first);
        std::string newick;
        if (newick_path != "-") {
            std::ifstream tin(newick_path);
            std::stringstream ss;
            ss << tin.rdbuf();
            newick = ss.str();
        } else {
            newick = in_order_newick_string(names);
        }
        Tree tree(newick);
        Core core(std::move(parsed), std::move(tree));
        if (names.size() == 2) params.set<bool>("preserve_subproblems", true);
        params.apply(core);
        if (restart) core.restart();
        core.execute();
        std::ofstream fo;
        std::ostream* os = &std::cout;
        if (out != "-") { fo.open(out); os = &fo; }
        if (names.size() == 2) {
            const auto& root = core.root_subproblem();
            *os << explicit_cigar(root.alignment, core.leaf_subproblem(names.front()).graph, core.leaf_subproblem(names.back()).graph) << '\n';
        } else {
            const auto& root = core.root_subproblem();
            write_gfa(root.graph, root.tableau, *os);
        }
        return 0;
    } catch (std::exception& ex) {
        fprintf(stderr, "ref_cli: %s\n", ex.what());
        return 3;
    }
}
