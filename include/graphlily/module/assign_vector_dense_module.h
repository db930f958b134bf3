// graphlily/module/assign_vector_dense_module.h -- AssignVectorDenseModule on MI355X (reference
// module/assign_vector_dense_module.h:19-246): masked inout[i] = val.
#ifndef GRAPHLILY_ASSIGN_VECTOR_DENSE_MODULE_H_
#define GRAPHLILY_ASSIGN_VECTOR_DENSE_MODULE_H_

#include <type_traits>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/module/base_module.h"

namespace graphlily {
namespace module {

template <typename vector_data_t>
class AssignVectorDenseModule : public BaseModule {
    static_assert(sizeof(vector_data_t) == 4, "one of the reference's 32-bit value types (float, unsigned, graphlily::ufixed_32_8)");
    typedef graphlily::value_kind<vector_data_t> VK;
    using aligned_dense_vec_t = std::vector<vector_data_t, aligned_allocator<vector_data_t>>;
    graphlily::MaskType mask_type_ = graphlily::kNoMask;
    aligned_dense_vec_t mask_, inout_;

public:
    DeviceBuffer mask_buf;
    DeviceBuffer inout_buf;

    AssignVectorDenseModule() : BaseModule("overlay") {}

    void set_mask_type(graphlily::MaskType mask_type) {
        if (mask_type == graphlily::kNoMask) {
            std::cerr << "Please set the mask type" << std::endl;
            exit(EXIT_FAILURE);
        }
        mask_type_ = mask_type;
    }

    void send_mask_host_to_device(aligned_dense_vec_t &mask) {
        barrier_();
        mask_buf = DeviceBuffer(sizeof(float) * mask.size());
        mask_buf.upload(mask.data(), sizeof(float) * mask.size());
    }
    void send_inout_host_to_device(aligned_dense_vec_t &inout) {
        barrier_();
        inout_buf = DeviceBuffer(sizeof(float) * inout.size());
        inout_buf.upload(inout.data(), sizeof(float) * inout.size());
    }
    void bind_mask_buf(DeviceBuffer src_buf) {
        settle_deferred_();
        mask_buf = src_buf;
    }
    void bind_inout_buf(DeviceBuffer src_buf) {
        settle_deferred_();
        inout_buf = src_buf;
    }

    void run(uint32_t len, vector_data_t val) {
        if (mask_type_ != graphlily::kMaskWriteToZero && mask_type_ != graphlily::kMaskWriteToOne) {
            std::cout << "Invalid mask type" << std::endl;
            exit(EXIT_FAILURE);
        }
        // the third call of a BFS pull iteration whose first two are deferred: the three run as one fused step (module/fusion.h)
        if (!blocking_ && VK::kind == GL_VAL_FLOAT && detail::fusion().fire(mask_buf, inout_buf, len, (float)val, (int)mask_type_)) return;
        barrier_();
        GRAPHLILY_CHECK(gl_assign_dense_typed(mask_buf.ptr(), inout_buf.ptr(), len, VK::bits(val), (int)mask_type_, VK::kind));
        finish_();
    }

    aligned_dense_vec_t send_mask_device_to_host() {
        mask_.resize(mask_buf.size() / sizeof(float));
        mask_buf.download(mask_.data(), sizeof(float) * mask_.size());
        return mask_;
    }
    aligned_dense_vec_t send_inout_device_to_host() {
        inout_.resize(inout_buf.size() / sizeof(float));
        inout_buf.download(inout_.data(), sizeof(float) * inout_.size());
        return inout_;
    }

    void compute_reference_results(graphlily::aligned_dense_float_vec_t &mask, graphlily::aligned_dense_float_vec_t &inout,
                                   uint32_t len, float val) {
        if (mask_type_ != graphlily::kMaskWriteToZero && mask_type_ != graphlily::kMaskWriteToOne) {
            std::cout << "Invalid mask type" << std::endl;
            exit(EXIT_FAILURE);
        }
        const bool on_zero = (mask_type_ == graphlily::kMaskWriteToZero);
        for (uint32_t i = 0; i < len; i++)
            if ((mask[i] == 0) == on_zero) inout[i] = val;
    }
};

}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_ASSIGN_VECTOR_DENSE_MODULE_H_
