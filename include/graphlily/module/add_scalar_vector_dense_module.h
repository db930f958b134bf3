// graphlily/module/add_scalar_vector_dense_module.h -- eWiseAddModule on MI355X (reference
// module/add_scalar_vector_dense_module.h:19-204): out[i] = in[i] + val.
#ifndef GRAPHLILY_EWISE_ADD_MODULE_H_
#define GRAPHLILY_EWISE_ADD_MODULE_H_

#include <type_traits>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/module/base_module.h"

namespace graphlily {
namespace module {

template <typename vector_data_t>
class eWiseAddModule : public BaseModule {
    static_assert(sizeof(vector_data_t) == 4, "one of the reference's 32-bit value types (float, unsigned, graphlily::ufixed_32_8)");
    typedef graphlily::value_kind<vector_data_t> VK;
    using aligned_dense_vec_t = std::vector<vector_data_t, aligned_allocator<vector_data_t>>;
    aligned_dense_vec_t out_;   // host staging for send_out_device_to_host

    void run_now_(uint32_t len, vector_data_t val) {
        GRAPHLILY_CHECK(gl_ewise_add_typed(in_buf.ptr(), out_buf.ptr(), len, VK::bits(val), VK::kind));
    }

public:
    DeviceBuffer in_buf;
    DeviceBuffer out_buf;

    eWiseAddModule() : BaseModule("overlay") {}

    void send_in_host_to_device(aligned_dense_vec_t &in) {
        barrier_();
        in_buf = DeviceBuffer(sizeof(float) * in.size());
        in_buf.upload(in.data(), sizeof(float) * in.size());
    }
    void allocate_out_buf(uint32_t len) {
        settle_deferred_();
        out_buf = DeviceBuffer(sizeof(float) * len);
    }
    void bind_in_buf(DeviceBuffer src_buf) {
        settle_deferred_();
        in_buf = src_buf;
    }
    void bind_out_buf(DeviceBuffer src_buf) {
        settle_deferred_();
        out_buf = src_buf;
    }

    void run(uint32_t len, vector_data_t val) {
        // the results -> vector copy of a BFS pull iteration whose SpMV is deferred?  Then it waits too (module/fusion.h)
        if (!blocking_ && VK::kind == GL_VAL_FLOAT &&
            detail::fusion().defer_ewise(in_buf, out_buf, len, (float)val, [this, len, val] { run_now_(len, val); }))
            return;
        barrier_();
        run_now_(len, val);
        finish_();
    }

    aligned_dense_vec_t send_out_device_to_host() {
        out_.resize(out_buf.size() / sizeof(float));
        out_buf.download(out_.data(), sizeof(float) * out_.size());
        return out_;
    }

    graphlily::aligned_dense_float_vec_t compute_reference_results(graphlily::aligned_dense_float_vec_t const &in,
                                                                    uint32_t len, float val) {
        graphlily::aligned_dense_float_vec_t out(len);
        for (uint32_t i = 0; i < len; i++) out[i] = in[i] + val;
        return out;
    }
};

}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_EWISE_ADD_MODULE_H_
