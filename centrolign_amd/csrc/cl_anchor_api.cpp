// cl_anchor_api.cpp — host-side anchor-chain post-processing of the stitch path.
//   cl_despecify_indel_breakpoints <-> Stitcher::despecify_indel_breakpoints (src/stitcher.cpp:265-310) over
//                                      identify_despecification_partition (src/stitcher.cpp:115-263) and
//                                      PartitionClient::traceback (include/centrolign/partition_client.hpp:31-54)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <tuple>
#include <vector>

#include "cl_internal.hpp"
#include "search_trees.hpp"

namespace {

using Val = std::tuple<int64_t, double, size_t>;

// src/stitcher.cpp:115-263
std::vector<std::pair<size_t, size_t>> despecification_partition(uint64_t n, const double* score, const int64_t* gap_before,
                                                                 int64_t min_len, double prop) {
    static const double inf = std::numeric_limits<double>::max();
    static const double mininf = std::numeric_limits<double>::lowest();
    std::vector<std::pair<int64_t, int64_t>> limit(n, {0, 0});
    int64_t prev_indel = -1, before_prev = -1;
    for (size_t i = 0; i < n; ++i) {
        if (i != 0 && std::llabs(gap_before[i]) >= min_len) { before_prev = prev_indel; prev_indel = (int64_t)i; }
        if (before_prev != -1 && prev_indel != -1) {
            limit[i].first = before_prev + 1;
            limit[i].second = std::min<int64_t>((int64_t)i, prev_indel + 1);
        } else if (prev_indel != -1) {
            limit[i].first = std::min<int64_t>(1, (int64_t)i);
            limit[i].second = std::min<int64_t>(prev_indel + 1, (int64_t)i);
        }
    }
    std::vector<double> prefix(n + 1, 0.0);
    for (size_t i = 0; i < n; ++i) prefix[i + 1] = prefix[i] + score[i];
    std::vector<double> index_key(n, mininf);
    for (size_t i = 1; i < n; ++i) index_key[i] = prefix[i] + prop * score[i - 1];
    std::vector<std::tuple<int64_t, double, Val>> data;
    data.reserve(n + 1);
    for (size_t i = 0; i < n; ++i) data.emplace_back((int64_t)i, index_key[i], Val(0, 0.0, 0));
    clhost::OrthoTree<int64_t, double, Val> tree(data);
    std::vector<std::pair<Val, Val>> dp(n + 1, {Val(-1, 0.0, 0), Val(-1, 0.0, 0)});
    std::vector<size_t> back(n + 1, (size_t)-1);
    std::get<0>(dp.front().first) = 0;
    size_t opt = 0;
    for (size_t i = 1; i + 1 < dp.size(); ++i) {
        dp[i].first = std::max(dp[i - 1].first, dp[i - 1].second);
        const double query_key = prefix[i] - prop * score[i];
        const size_t it = tree.range_max(limit[i].first, limit[i].second, query_key, inf);
        if (it != tree.end()) {
            const Val& v = tree.val[it];
            // (the second field is computed from get<0> in the reference, src/stitcher.cpp:221-223; kept as is)
            dp[i].second = Val(std::get<0>(v) + 1, (double)std::get<0>(v) - prefix[i] + prefix[(size_t)tree.key1[it]], i);
            back[i] = (size_t)tree.key1[it];
            if (dp[i].second > dp[opt].second) opt = i;
        }
        const size_t at = tree.find((int64_t)i, index_key[i]);
        tree.update(at, Val(std::get<0>(dp[i].first), std::get<1>(dp[i].first), i));
    }
    // partition_client.hpp:31-54
    std::vector<std::pair<size_t, size_t>> partition;
    bool in_interval = true;
    size_t tb = opt;
    while (tb > 0) {
        if (in_interval) {
            const size_t prev = back[tb];
            partition.emplace_back(prev, tb);
            tb = prev;
            in_interval = false;
        } else {
            in_interval = (dp[tb].first == dp[tb - 1].second);
            --tb;
        }
    }
    std::reverse(partition.begin(), partition.end());
    return partition;
}

}  // namespace

extern "C" {

int cl_despecify_indel_breakpoints(uint64_t n_anchors, const double* score, int64_t* gap_before, double* gap_score_before,
                                   int64_t* gap_after, double* gap_score_after, int64_t min_indel_fuzz_length,
                                   double indel_fuzz_score_proportion, uint8_t* keep_out, uint64_t* n_kept_out) {
    if ((n_anchors && (!score || !gap_before || !gap_score_before || !gap_after || !gap_score_after || !keep_out)) || !n_kept_out) {
        cl_set_error(nullptr, "null argument");
        return CL_ERR_INVALID_ARGUMENT;
    }
    const auto removal = despecification_partition(n_anchors, score, gap_before, min_indel_fuzz_length, indel_fuzz_score_proportion);
    // src/stitcher.cpp:282-309, on parallel arrays: kept anchors are compacted to the front of the gap arrays
    size_t removed = 0, d = 0;
    int64_t gap = 0;
    double gap_score = 0.0;
    std::vector<int64_t> gb(gap_before, gap_before + n_anchors), ga(gap_after, gap_after + n_anchors);
    std::vector<double> gsb(gap_score_before, gap_score_before + n_anchors), gsa(gap_score_after, gap_score_after + n_anchors);
    std::vector<size_t> origin(n_anchors);
    for (size_t i = 0; i < n_anchors; ++i) origin[i] = i;
    for (size_t i = 0; i < n_anchors; ++i) keep_out[i] = 1;
    for (size_t i = 0; i < n_anchors; ++i) {
        if (d < removal.size() && i >= removal[d].first && i < removal[d].second) {
            gap += gb[i];
            gap_score += gsb[i];
            keep_out[origin[i]] = 0;
            ++removed;
        } else if (removed != 0) {
            gb[i - removed] = gb[i]; ga[i - removed] = ga[i]; gsb[i - removed] = gsb[i]; gsa[i - removed] = gsa[i];
            origin[i - removed] = origin[i];
        }
        if (d < removal.size() && i == removal[d].second) {
            ga[i - removed - 1] = gap;
            gsa[i - removed - 1] = gap_score;
            gb[i - removed] = gap;
            gsb[i - removed] = gap_score;
            gap = 0;
            gap_score = 0.0;
            ++d;
        }
    }
    const size_t kept = n_anchors - removed;
    for (size_t i = 0; i < kept; ++i) { gap_before[i] = gb[i]; gap_after[i] = ga[i]; gap_score_before[i] = gsb[i]; gap_score_after[i] = gsa[i]; }
    *n_kept_out = kept;
    return CL_OK;
}

}  // extern "C"
