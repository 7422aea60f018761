// stitch_adapter.hpp — header-only C++11 glue between centrolign's own types and the C ABI of
// include/centrolign_amd.h.  It is what a centrolign maintainer includes to route the subalign loop of
// Stitcher::stitch (include/centrolign/stitcher.hpp:157-203) through the MI355X library; see INTEGRATION.md.
//
// It is written against the reference's *concepts*, not its headers, so it compiles on its own:
//   SubGraphInfoT : .subgraph (node_size(), label(id), next(id), previous(id)), .back_translation, .sources,
//                   .sinks                               (include/centrolign/subgraph_extraction.hpp:14-33)
//   AlignedPairT  : constructible from (uint64_t, uint64_t), standard layout {uint64_t node_id1, node_id2}
//                                                         (include/centrolign/alignment.hpp:34-46)
//   StitcherT     : public tunables alignment_params, max_trivial_size, ... (include/centrolign/stitcher.hpp:48-64)
#ifndef CENTROLIGN_AMD_STITCH_ADAPTER_HPP
#define CENTROLIGN_AMD_STITCH_ADAPTER_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../centrolign_amd.h"

namespace centrolign_amd {

// Flattens SubGraphInfo pairs into the cl_stitch_batch layout (neighbour order preserved).
class StitchBatchBuilder {
public:
    template <class SubGraphInfoT>
    void add(const SubGraphInfoT& s1, const SubGraphInfoT& s2, bool only_deletion_alns) {
        add_side(0, s1);
        add_side(1, s2);
        only_del_.push_back(only_deletion_alns ? 1 : 0);
    }
    size_t size() const { return only_del_.size(); }
    // the returned struct points into this builder; keep the builder alive while it is in use
    cl_stitch_batch view() const {
        cl_stitch_batch b;
        b.n_problems = only_del_.size();
        for (int s = 0; s < 2; ++s) {
            const Side& d = side_[s];
            b.side[s].node_off = d.node_off.data();
            b.side[s].label = d.label.data();
            b.side[s].prev_off = d.prev_off.data();
            b.side[s].prev_idx = d.prev_idx.data();
            b.side[s].next_off = d.next_off.data();
            b.side[s].next_idx = d.next_idx.data();
            b.side[s].src_off = d.src_off.data();
            b.side[s].src_idx = d.src_idx.data();
            b.side[s].snk_off = d.snk_off.data();
            b.side[s].snk_idx = d.snk_idx.data();
            b.side[s].back_translation = d.back.data();
        }
        b.only_deletion_alns = only_del_.data();
        return b;
    }

private:
    struct Side {
        std::vector<uint64_t> node_off{0}, prev_off{0}, next_off{0}, src_off{0}, snk_off{0}, back;
        std::vector<uint8_t> label;
        std::vector<uint32_t> prev_idx, next_idx, src_idx, snk_idx;
    };
    Side side_[2];
    std::vector<uint8_t> only_del_;

    template <class SubGraphInfoT>
    void add_side(int s, const SubGraphInfoT& info) {
        Side& d = side_[s];
        const auto& g = info.subgraph;
        for (uint64_t v = 0; v < g.node_size(); ++v) {
            d.label.push_back((uint8_t)g.label(v));
            for (auto p : g.previous(v)) d.prev_idx.push_back((uint32_t)p);
            d.prev_off.push_back(d.prev_idx.size());
            for (auto q : g.next(v)) d.next_idx.push_back((uint32_t)q);
            d.next_off.push_back(d.next_idx.size());
            d.back.push_back(info.back_translation[v]);
        }
        d.node_off.push_back(d.label.size());
        for (auto x : info.sources) d.src_idx.push_back((uint32_t)x);
        d.src_off.push_back(d.src_idx.size());
        for (auto x : info.sinks) d.snk_idx.push_back((uint32_t)x);
        d.snk_off.push_back(d.snk_idx.size());
    }
};

template <class StitcherT>
cl_stitch_params stitch_params_of(const StitcherT& st) {
    cl_stitch_params p;
    p.alignment_params.match = st.alignment_params.match;
    p.alignment_params.mismatch = st.alignment_params.mismatch;
    for (int i = 0; i < 3; ++i) {
        p.alignment_params.gap_open[i] = st.alignment_params.gap_open[i];
        p.alignment_params.gap_extend[i] = st.alignment_params.gap_extend[i];
    }
    p.max_trivial_size = st.max_trivial_size;
    p.min_wfa_size = st.min_wfa_size;
    p.max_wfa_size = st.max_wfa_size;
    p.max_wfa_ratio = st.max_wfa_ratio;
    p.wfa_pruning_dist = st.wfa_pruning_dist;
    p.deletion_alignment_ratio = st.deletion_alignment_ratio;
    p.deletion_alignment_short_max_size = st.deletion_alignment_short_max_size;
    p.deletion_alignment_long_min_size = st.deletion_alignment_long_min_size;
    return p;
}

// RAII over cl_context; throws std::runtime_error like the reference does on bad input (src/stitcher.cpp:36)
class Device {
public:
    explicit Device(int ordinal = 0) : ctx_(cl_context_create(ordinal)) {
        if (!ctx_) throw std::runtime_error(std::string("centrolign_amd: ") + cl_last_error(nullptr));
    }
    ~Device() { cl_context_destroy(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    cl_context* get() const { return ctx_; }

    // Stitcher::subalign for every problem of the batch; appends nothing itself: result k is the
    // inter-anchor alignment the reference would have pushed onto `stitched` for subproblem k.
    template <class AlignedPairT>
    std::vector<std::vector<AlignedPairT>> subalign_all(const StitchBatchBuilder& batch, const cl_stitch_params& params) const {
        cl_stitch_batch b = batch.view();
        cl_stitch_result r;
        int rc = cl_stitch_batch_align(ctx_, &b, &params, &r);
        if (rc != CL_OK) throw std::runtime_error(std::string("centrolign_amd: ") + cl_last_error(ctx_));
        std::vector<std::vector<AlignedPairT>> out(r.n_problems);
        for (uint64_t k = 0; k < r.n_problems; ++k) {
            out[k].reserve(r.aln_off[k + 1] - r.aln_off[k]);
            for (uint64_t i = r.aln_off[k]; i < r.aln_off[k + 1]; ++i) out[k].emplace_back(r.pairs[2 * i], r.pairs[2 * i + 1]);
        }
        cl_stitch_result_free(&r);
        return out;
    }

private:
    cl_context* ctx_;
};

}  // namespace centrolign_amd

#endif
