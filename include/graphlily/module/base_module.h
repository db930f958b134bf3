// graphlily/module/base_module.h -- common part of the operator modules (reference
// module/base_module.h:10-133) on the HIP C ABI: no OpenCL device/context/kernel/queue objects; the
// "runtime" is the library's device + stream (gl_init / gl_sync).
#ifndef GRAPHLILY_BASE_MODULE_H_
#define GRAPHLILY_BASE_MODULE_H_

#include <cassert>
#include <string>

#include "graphlily/global.h"
#include "graphlily/module/fusion.h"

namespace graphlily {
namespace module {

class BaseModule {
protected:
    std::string kernel_name_;
    std::string target_ = "hw";
    int device_ = 0;
    // Every reference call ends in command_queue_.finish().  A stand-alone module keeps that; modules owned by a
    // ModuleCollection run non-blocking (module_collection.h): all work goes to one in-order stream and every call
    // that hands data to the host (send_*_device_to_host, get_results_nnz, uploads) waits for it anyway, so the
    // callers see the same values without a host round trip per launch.
    bool blocking_ = true;
    const void *owner_ = nullptr;   // the ModuleCollection this module belongs to (module/fusion.h pairs modules of one owner)

    void finish_() {
        if (blocking_) GRAPHLILY_CHECK(gl_sync());
    }
    // every module call that is not the expected continuation of a deferred BFS pull iteration first lets the deferred
    // calls run (module/fusion.h)
    static void barrier_() {
        ++calls_();
        detail::fusion().flush();
    }
    // a setter or binder: deferred calls were made with the OLD state and must run with it (they read the module's members
    // when they run, not when they were deferred); not a "call" for the bookkeeping below
    static void settle_deferred_() { detail::fusion().flush(); }
    // module calls so far, whichever module: a module that has learnt something about a device buffer from the host (the
    // SpMSpV module: the work of the vector it uploaded) trusts it only while no other call has run in between
    static uint64_t &calls_() {
        static uint64_t n = 0;
        return n;
    }

public:
    explicit BaseModule(std::string kernel_name) : kernel_name_(kernel_name) {}
    virtual ~BaseModule() { detail::fusion().forget(this); }

    std::string get_kernel_name() { return kernel_name_; }

    void set_target(std::string target) {
        assert(target == "sw_emu" || target == "hw_emu" || target == "hw");
        target_ = target;
    }

    // extension: which GPU this module's buffers and kernels live on (default 0)
    void set_device(int device) { device_ = device; }

    // extension: let a driver enqueue several module calls and synchronise once (gl_sync)
    void set_blocking(bool blocking) { blocking_ = blocking; }
    void set_owner(const void *owner) { owner_ = owner; }

    void copy_buffer_device_to_device(DeviceBuffer src, DeviceBuffer dst, size_t bytes) {
        barrier_();
        GRAPHLILY_CHECK(gl_buf_d2d(dst.ptr(), src.ptr(), bytes));
        finish_();
    }

    // The fused overlay needed its unused ports tied off and a mode selected (reference :90-96);
    // separate HIP kernels need neither.  Kept virtual so caller code that invokes them still links.
    virtual void set_unused_args() {}
    virtual void set_mode() {}

    // The bitstream path is accepted and ignored.
    void set_up_runtime(std::string /*xclbin_file_path*/) {
        GRAPHLILY_CHECK(gl_init(device_));
        set_unused_args();
        set_mode();
    }
};

}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_BASE_MODULE_H_
