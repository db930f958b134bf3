// graphlily/synthesizer/overlay_synthesizer.h -- the "build the kernel" step of the reference's drivers, on this backend.
//
// The reference's OverlaySynthesizer (graphlily/synthesizer/overlay_synthesizer.h:10-73 over base_synthesizer.h:42-135) writes a
// kernel header, a Makefile and a connectivity .ini into ./proj and runs the Vitis build: what `bench_spmspv <target> build`
// (benchmark/bench_spmspv.cpp:294-306) and tests/test_module_*.cpp's fixtures call before anything runs.  Here the kernels are
// hand-written HIP, compiled ahead of time into libgraphlily_hip.so for gfx950: there is nothing to generate or to synthesise.
// This header keeps the class, its constructor, set_target() and synthesize() -- the surface the drivers name -- so that they
// compile UNMODIFIED; synthesize() checks that the library is there and says which one.  The HLS code generation itself is out of
// scope.
#ifndef GRAPHLILY_HIP_OVERLAY_SYNTHESIZER_H_
#define GRAPHLILY_HIP_OVERLAY_SYNTHESIZER_H_

#include <cassert>
#include <cstdint>
#include <fstream>      // (the reference's header brings it in, and bench_spmspv.cpp:308 counts on that)
#include <iostream>
#include <string>

#include "graphlily/global.h"

namespace graphlily {
namespace synthesizer {

class BaseSynthesizer {
protected:
    std::string kernel_name_;
    std::string target_;

public:
    explicit BaseSynthesizer(std::string kernel_name) : kernel_name_(kernel_name) {}
    virtual ~BaseSynthesizer() {}
    std::string get_kernel_name() { return kernel_name_; }
    // base_synthesizer.h:72-75: the same three targets, the same assert
    void set_target(std::string target) {
        assert(target == "sw_emu" || target == "hw_emu" || target == "hw");
        target_ = target;
    }
    virtual void generate_kernel_header() {}
    virtual void generate_kernel_ini() {}
    virtual void link_kernel_code() {}
    virtual void generate_makefile() {}
    // base_synthesizer.h:102-104: "synthesize the kernel according to target_" -- here: the prebuilt library answers
    virtual void synthesize() {
        std::cout << "INFO: [graphlily-hip] kernel '" << kernel_name_ << "' is prebuilt for gfx950 in libgraphlily_hip.so (" << gl_version()
                  << "); nothing to synthesise for target '" << target_ << "'" << std::endl;
    }
};

class OverlaySynthesizer : public BaseSynthesizer {
    uint32_t num_channels_, spmv_out_buf_len_, spmspv_out_buf_len_, vec_buf_len_;   // (the FPGA's buffer geometry: not used)

public:
    OverlaySynthesizer(uint32_t num_channels, uint32_t spmv_out_buf_len, uint32_t spmspv_out_buf_len, uint32_t vec_buf_len)
        : BaseSynthesizer("overlay"), num_channels_(num_channels), spmv_out_buf_len_(spmv_out_buf_len),
          spmspv_out_buf_len_(spmspv_out_buf_len), vec_buf_len_(vec_buf_len) {
        (void)num_channels_; (void)spmv_out_buf_len_; (void)spmspv_out_buf_len_; (void)vec_buf_len_;
    }
};

}  // namespace synthesizer
}  // namespace graphlily

#endif  // GRAPHLILY_HIP_OVERLAY_SYNTHESIZER_H_
