// Runtime plumbing of libgraphlily_hip.so: device/stream, error reporting, buffers.
// Replaces the OpenCL/XRT set-up of module/base_module.h:106-133 and the
// cl::Buffer migrate/copy calls of the reference modules.
#include "gl_common.h"

#include <cstring>

namespace gl {

Context &ctx() {
    static Context c;
    return c;
}

Profiler &prof() {
    static Profiler p;
    return p;
}

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

__global__ void fill_f32_kernel(float *__restrict__ dst, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = v;
}

}  // namespace gl

extern "C" {

const char *gl_last_error(void) { return gl::g_err; }

const char *gl_version(void) { return "graphlily_hip 0.1 (gfx950)"; }

int gl_device_count(int *count) {
    GL_ARG(count != nullptr);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return gl::set_error(GL_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return GL_OK;
}

int gl_init(int device) {
    gl::Context &c = gl::ctx();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return gl::set_error(GL_ERR_HIP, "gl_init: no HIP device available (%s)",
                             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    GL_ARG(device >= 0 && device < n);
    GL_HIP(hipSetDevice(device));
    if (c.initialized && c.device == device) return GL_OK;
    if (c.own_stream) {
        (void)hipStreamDestroy(c.own_stream);
        c.own_stream = nullptr;
    }
    GL_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    GL_HIP(hipGetDeviceProperties(&prop, device));
    c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c.stream = c.own_stream;
    c.device = device;
    c.initialized = true;
    return GL_OK;
}

int gl_set_stream(void *hip_stream) {
    GL_REQUIRE_INIT();
    gl::Context &c = gl::ctx();
    c.stream = (hipStream_t)hip_stream;
    return GL_OK;
}

int gl_reset_stream(void) {
    GL_REQUIRE_INIT();
    gl::ctx().stream = gl::ctx().own_stream;
    return GL_OK;
}

int gl_prof_begin(uint32_t max_launches) {
    GL_REQUIRE_INIT();
    gl::Profiler &p = gl::prof();
    while (p.events.size() < 2ull * max_launches) {
        hipEvent_t e;
        GL_HIP(hipEventCreate(&e));
        p.events.push_back(e);
    }
    p.used = 0;
    p.seen = 0;
    p.on = true;
    return GL_OK;
}

int gl_prof_sample_every(uint32_t n) {
    gl::prof().every = n ? n : 1u;
    return GL_OK;
}

int gl_prof_end(double *total_ms, uint32_t *launches) {
    GL_REQUIRE_INIT();
    gl::Profiler &p = gl::prof();
    p.on = false;
    GL_HIP(hipStreamSynchronize(gl::ctx().stream));
    double sum = 0.0;
    for (uint32_t i = 0; i < p.used; i++) {
        float ms = 0.0f;
        GL_HIP(hipEventElapsedTime(&ms, p.events[2 * i], p.events[2 * i + 1]));
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = p.used;
    return GL_OK;
}

int gl_sync(void) {
    GL_REQUIRE_INIT();
    GL_HIP(hipStreamSynchronize(gl::ctx().stream));
    return GL_OK;
}

int gl_buf_alloc(void **d_ptr, size_t bytes) {
    GL_REQUIRE_INIT();
    GL_ARG(d_ptr != nullptr);
    *d_ptr = nullptr;
    GL_HIP(hipMalloc(d_ptr, bytes ? bytes : 4));
    return GL_OK;
}

int gl_buf_free(void *d_ptr) {
    GL_REQUIRE_INIT();
    if (d_ptr) GL_HIP(hipFree(d_ptr));
    return GL_OK;
}

int gl_buf_h2d(void *d_dst, const void *h_src, size_t bytes) {
    GL_REQUIRE_INIT();
    if (bytes == 0) return GL_OK;
    GL_ARG(d_dst != nullptr && h_src != nullptr);
    hipStream_t s = gl::ctx().stream;
    GL_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
    GL_HIP(hipStreamSynchronize(s));
    return GL_OK;
}

int gl_buf_d2h(void *h_dst, const void *d_src, size_t bytes) {
    GL_REQUIRE_INIT();
    if (bytes == 0) return GL_OK;
    GL_ARG(h_dst != nullptr && d_src != nullptr);
    hipStream_t s = gl::ctx().stream;
    GL_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    GL_HIP(hipStreamSynchronize(s));
    return GL_OK;
}

int gl_buf_d2d(void *d_dst, const void *d_src, size_t bytes) {
    GL_REQUIRE_INIT();
    if (bytes == 0) return GL_OK;
    GL_ARG(d_dst != nullptr && d_src != nullptr);
    GL_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, gl::ctx().stream));
    return GL_OK;
}

int gl_host_alloc(void **h_ptr, size_t bytes) {
    GL_REQUIRE_INIT();
    GL_ARG(h_ptr != nullptr);
    *h_ptr = nullptr;
    GL_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 4, hipHostMallocDefault));
    return GL_OK;
}

int gl_host_free(void *h_ptr) {
    GL_REQUIRE_INIT();
    if (h_ptr) GL_HIP(hipHostFree(h_ptr));
    return GL_OK;
}

int gl_buf_fill_f32(float *d_dst, float value, size_t count) {
    GL_REQUIRE_INIT();
    if (count == 0) return GL_OK;
    GL_ARG(d_dst != nullptr);
    unsigned blocks = gl::cdiv(count, 256);
    if (blocks > 2048) blocks = 2048;
    gl::fill_f32_kernel<<<blocks, 256, 0, gl::ctx().stream>>>(d_dst, value, count);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

}  // extern "C"
