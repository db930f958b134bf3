// graphlily/cl_buffers.h -- the buffer handles of the module API as the reference's callers spell them.
//
// The reference's modules expose their device buffers as public `cl::Buffer` members and take `cl::Buffer` in bind_*_buf /
// copy_buffer_device_to_device (module/base_module.h:82-85, assign_vector_dense_module.h:118-126); one caller in the tree also
// MAKES such a buffer itself and reads it back through a command queue of its own -- tests/test_module_apply.cpp:236-256:
//     cl_mem_ext_ptr_t x_ext{...HBM[...]};  cl::Buffer x_buf(context, CL_MEM_EXT_PTR_XILINX | CL_MEM_USE_HOST_PTR, bytes, &x_ext);
//     module.bind_inout_buf(x_buf);  module.run(...);  command_queue.enqueueMigrateMemObjects({x_buf}, CL_MIGRATE_MEM_OBJECT_HOST);
// On this backend a module buffer is a graphlily::DeviceBuffer (global.h: a shared handle of a block of HBM from the library's
// pool).  This header gives callers of that kind the four names they use, with the meaning they rely on and nothing of OpenCL
// behind them: `cl::Buffer` IS a DeviceBuffer that remembers the host block it mirrors (CL_MEM_USE_HOST_PTR: created holding the
// host block's contents), `cl::CommandQueue::enqueueMigrateMemObjects` copies between the two in the direction asked for,
// `finish()` is gl_sync, `cl::Device` / `cl::Context` are tags (there is one device context per process: gl_init).  The HBM / DDR
// bank words (global.h:42-54) are kept as values only -- every block lives in the one HBM pool.
#ifndef GRAPHLILY_HIP_CL_BUFFERS_H_
#define GRAPHLILY_HIP_CL_BUFFERS_H_

#include <vector>

#include "graphlily/global.h"

// CL/cl_ext_xilinx.h: the extension pointer a caller passes with CL_MEM_EXT_PTR_XILINX
struct cl_mem_ext_ptr_t {
    unsigned flags;
    void *obj;
    void *param;
};
#ifndef CL_MEM_USE_HOST_PTR
#define CL_MEM_READ_WRITE (1u << 0)
#define CL_MEM_WRITE_ONLY (1u << 1)
#define CL_MEM_READ_ONLY (1u << 2)
#define CL_MEM_USE_HOST_PTR (1u << 3)
#define CL_MEM_EXT_PTR_XILINX (1u << 31)
#define CL_MIGRATE_MEM_OBJECT_HOST (1u << 0)
#define XCL_MEM_TOPOLOGY (1u << 31)
#endif

namespace cl {

class Device {};
class Context {
public:
    Context() {}
    Context(const Device &, const void *, const void *, const void *) {}
};

class Buffer : public graphlily::DeviceBuffer {
    void *host_ = nullptr;     // the block this buffer mirrors (CL_MEM_USE_HOST_PTR), or null

public:
    Buffer() {}
    Buffer(const graphlily::DeviceBuffer &b) : graphlily::DeviceBuffer(b) {}   // (a module's own buffer member, named as a cl::Buffer)
    // cl::Buffer(context, flags, size, host_ptr): with CL_MEM_EXT_PTR_XILINX host_ptr is a cl_mem_ext_ptr_t whose obj is the
    // host block; with CL_MEM_USE_HOST_PTR the buffer starts out holding that block's contents
    Buffer(const Context &, unsigned flags, size_t bytes, void *host_ptr, int *err = nullptr) : graphlily::DeviceBuffer(bytes) {
        void *h = host_ptr;
        if ((flags & CL_MEM_EXT_PTR_XILINX) && host_ptr) h = static_cast<cl_mem_ext_ptr_t *>(host_ptr)->obj;
        if ((flags & CL_MEM_USE_HOST_PTR) && h) {
            host_ = h;
            upload(h, bytes);
        }
        if (err) *err = 0;
    }
    void *host_block() const { return host_; }
};

class CommandQueue {
public:
    CommandQueue() {}
    CommandQueue(const Context &, const Device &, unsigned = 0, int *err = nullptr) {
        if (err) *err = 0;
    }
    // flags == CL_MIGRATE_MEM_OBJECT_HOST: device -> the mirrored host block; 0: host block -> device (OpenCL 1.2 5.4.4)
    int enqueueMigrateMemObjects(const std::vector<Buffer> &bufs, unsigned flags) const {
        for (const Buffer &b : bufs) {
            if (!b.valid() || !b.host_block()) continue;
            if (flags & CL_MIGRATE_MEM_OBJECT_HOST) b.download(b.host_block(), b.size());
            else b.upload(b.host_block(), b.size());
        }
        return 0;
    }
    int finish() const {
        GRAPHLILY_CHECK(gl_sync());
        return 0;
    }
};

}  // namespace cl

namespace graphlily {

// global.h:42-54: bank words of the Alveo U280's 32 HBM pseudo-channels and two DDR banks (values only here)
#define GRAPHLILY_CHANNEL_NAME(n) (int)((unsigned)(n) | XCL_MEM_TOPOLOGY)
const int HBM[32] = {
    GRAPHLILY_CHANNEL_NAME(0),  GRAPHLILY_CHANNEL_NAME(1),  GRAPHLILY_CHANNEL_NAME(2),  GRAPHLILY_CHANNEL_NAME(3),
    GRAPHLILY_CHANNEL_NAME(4),  GRAPHLILY_CHANNEL_NAME(5),  GRAPHLILY_CHANNEL_NAME(6),  GRAPHLILY_CHANNEL_NAME(7),
    GRAPHLILY_CHANNEL_NAME(8),  GRAPHLILY_CHANNEL_NAME(9),  GRAPHLILY_CHANNEL_NAME(10), GRAPHLILY_CHANNEL_NAME(11),
    GRAPHLILY_CHANNEL_NAME(12), GRAPHLILY_CHANNEL_NAME(13), GRAPHLILY_CHANNEL_NAME(14), GRAPHLILY_CHANNEL_NAME(15),
    GRAPHLILY_CHANNEL_NAME(16), GRAPHLILY_CHANNEL_NAME(17), GRAPHLILY_CHANNEL_NAME(18), GRAPHLILY_CHANNEL_NAME(19),
    GRAPHLILY_CHANNEL_NAME(20), GRAPHLILY_CHANNEL_NAME(21), GRAPHLILY_CHANNEL_NAME(22), GRAPHLILY_CHANNEL_NAME(23),
    GRAPHLILY_CHANNEL_NAME(24), GRAPHLILY_CHANNEL_NAME(25), GRAPHLILY_CHANNEL_NAME(26), GRAPHLILY_CHANNEL_NAME(27),
    GRAPHLILY_CHANNEL_NAME(28), GRAPHLILY_CHANNEL_NAME(29), GRAPHLILY_CHANNEL_NAME(30), GRAPHLILY_CHANNEL_NAME(31)};
const int DDR[2] = {GRAPHLILY_CHANNEL_NAME(32), GRAPHLILY_CHANNEL_NAME(33)};
#undef GRAPHLILY_CHANNEL_NAME

// global.h:30-40 looks for the Alveo board among the OpenCL devices; here a tag: the device is the one set_up_runtime / gl_init
// selected, and buffer calls of a process that never did fail with GL_ERR_NOT_INITIALIZED
inline cl::Device find_device() { return cl::Device(); }

}  // namespace graphlily

#endif  // GRAPHLILY_HIP_CL_BUFFERS_H_
