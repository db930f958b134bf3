// graphlily/module/assign_vector_sparse_module.h -- AssignVectorSparseModule on MI355X (reference
// module/assign_vector_sparse_module.h:19-335): BFS mode scatters a value at the mask's indices,
// SSSP mode relaxes and emits the new frontier.
#ifndef GRAPHLILY_ASSIGN_VECTOR_SPARSE_MODULE_H_
#define GRAPHLILY_ASSIGN_VECTOR_SPARSE_MODULE_H_

#include <algorithm>
#include <type_traits>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/module/base_module.h"

namespace graphlily {
namespace module {

template <typename vector_data_t, typename sparse_vector_data_t>
class AssignVectorSparseModule : public BaseModule {
    static_assert(sizeof(vector_data_t) == 4, "one of the reference's 32-bit value types (float, unsigned, graphlily::ufixed_32_8)");
    static_assert(sizeof(sparse_vector_data_t) == sizeof(gl_idx_val), "sparse element must be {uint32 index; 32-bit value}");
    typedef graphlily::value_kind<vector_data_t> VK;
    using aligned_mask_t = std::vector<sparse_vector_data_t, aligned_allocator<sparse_vector_data_t>>;
    using aligned_dense_vec_t = std::vector<vector_data_t, aligned_allocator<vector_data_t>>;

    bool generate_new_frontier_;
    aligned_mask_t mask_, new_frontier_;
    aligned_dense_vec_t inout_;

    static void die_(const char *msg) {
        std::cout << msg << std::endl;
        exit(EXIT_FAILURE);
    }
    uint32_t capacity_() const {
        size_t cap = mask_buf.size() / sizeof(gl_idx_val);
        if (generate_new_frontier_) cap = std::min(cap, new_frontier_buf.size() / sizeof(gl_idx_val));
        return cap ? (uint32_t)(cap - 1) : 0u;
    }

public:
    DeviceBuffer mask_buf;
    DeviceBuffer inout_buf;
    DeviceBuffer new_frontier_buf;

    explicit AssignVectorSparseModule(bool generate_new_frontier)
        : BaseModule("overlay"), generate_new_frontier_(generate_new_frontier) {}

    void send_mask_host_to_device(aligned_mask_t &mask) {
        barrier_();
        mask_buf = DeviceBuffer(sizeof(gl_idx_val) * mask.size());
        mask_buf.upload(mask.data(), sizeof(gl_idx_val) * mask.size());
        if (generate_new_frontier_) {  // the new frontier can never be longer than the mask
            new_frontier_buf = DeviceBuffer(sizeof(gl_idx_val) * mask.size());
            const gl_idx_val head{0u, 0.0f};
            if (!mask.empty()) new_frontier_buf.upload(&head, sizeof(head));
        }
    }
    void send_inout_host_to_device(aligned_dense_vec_t &inout) {
        barrier_();
        inout_buf = DeviceBuffer(sizeof(float) * inout.size());
        inout_buf.upload(inout.data(), sizeof(float) * inout.size());
    }
    void bind_mask_buf(DeviceBuffer src_buf) { mask_buf = src_buf; }
    void bind_inout_buf(DeviceBuffer src_buf) { inout_buf = src_buf; }
    void bind_new_frontier_buf(DeviceBuffer src_buf) {
        if (!generate_new_frontier_) die_("[ERROR]: this->generate_new_frontier_ should be true");
        new_frontier_buf = src_buf;
    }

    // BFS mode
    void run(vector_data_t val) {
        barrier_();
        if (generate_new_frontier_) die_("[ERROR]: this->generate_new_frontier_ should be false");
        GRAPHLILY_CHECK(gl_assign_sparse_typed(mask_buf.rptr(), inout_buf.ptr(), VK::bits(val), capacity_()));
        finish_();
    }
    // SSSP mode
    void run() {
        barrier_();
        if (!generate_new_frontier_) die_("[ERROR]: this->generate_new_frontier_ should be true");
        GRAPHLILY_CHECK(gl_assign_sparse_new_frontier_typed(mask_buf.rptr(), inout_buf.ptr(), new_frontier_buf.ptr(), capacity_(), VK::kind));
        finish_();
    }

    aligned_mask_t send_mask_device_to_host() {
        mask_.resize(mask_buf.size() / sizeof(gl_idx_val));
        mask_buf.download(mask_.data(), sizeof(gl_idx_val) * mask_.size());
        return mask_;
    }
    aligned_dense_vec_t send_inout_device_to_host() {
        inout_.resize(inout_buf.size() / sizeof(float));
        inout_buf.download(inout_.data(), sizeof(float) * inout_.size());
        return inout_;
    }
    aligned_mask_t send_new_frontier_device_to_host() {
        if (!generate_new_frontier_) die_("[ERROR]: this->generate_new_frontier_ should be true");
        new_frontier_.resize(new_frontier_buf.size() / sizeof(gl_idx_val));
        new_frontier_buf.download(new_frontier_.data(), sizeof(gl_idx_val) * new_frontier_.size());
        return new_frontier_;
    }

    void compute_reference_results(graphlily::aligned_sparse_float_vec_t &mask, graphlily::aligned_dense_float_vec_t &inout,
                                   float val) {
        for (size_t k = 1; k <= mask[0].index; k++) inout[mask[k].index] = val;
    }

    void compute_reference_results(graphlily::aligned_sparse_float_vec_t &mask, graphlily::aligned_dense_float_vec_t &inout,
                                   graphlily::aligned_sparse_float_vec_t &new_frontier) {
        new_frontier.assign(1, graphlily::idx_float_t{0, 0});
        for (size_t k = 1; k <= mask[0].index; k++) {
            if (inout[mask[k].index] > mask[k].val) {
                inout[mask[k].index] = mask[k].val;
                new_frontier.push_back(mask[k]);
            }
        }
        new_frontier[0].index = (graphlily::idx_t)(new_frontier.size() - 1);
    }
};

}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_ASSIGN_VECTOR_SPARSE_MODULE_H_
