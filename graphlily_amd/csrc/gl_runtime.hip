// Runtime plumbing of libgraphlily_hip.so: device/stream, error reporting, buffers.
// Replaces the OpenCL/XRT set-up of module/base_module.h:106-133 and the
// cl::Buffer migrate/copy calls of the reference modules.
#include "gl_common.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace gl {

// ------------------------------------------------------------------------------------------ block pools
// The reference's drivers create a fresh cl::Buffer and a fresh page-aligned host vector for every
// send_*_host_to_device / send_*_device_to_host (e.g. app/bfs.h:107-113: two n-element vectors and two buffers per
// BFS::pull call).  hipMalloc / hipFree / hipHostMalloc cost 0.1-3 ms each for 12 MB -- more than the SpMV they
// serve -- so both kinds of block are recycled by exact (rounded) size:
//   device blocks: gl_buf_alloc / gl_buf_free.  Reuse is ordered by the library's stream: a block freed while kernels
//     that read it are still queued is only ever handed to work enqueued later on the same stream (switching
//     streams drains the old one first, gl_set_stream).
//   pinned host blocks: gl_host_pool_alloc / gl_host_pool_free, behind the C++ layer's aligned_allocator.  Large
//     blocks are page-locked (uploads / downloads are then one DMA at PCIe rate instead of a staged pageable copy);
//     small ones, and everything before gl_init, come from posix_memalign.
// Cached bytes are capped (GRAPHLILY_POOL_MAX_MB, default 8192 device / 2048 host); gl_pool_trim releases them.
struct BlockPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;             // block -> rounded size (pinned blocks only, for the host pool)
    std::multimap<size_t, void *> cached;                // rounded size -> free block
    size_t cached_bytes = 0, cap_bytes = 0;
};

static BlockPool &device_pool() {
    static BlockPool *p = new BlockPool();   // never destroyed: blocks may be returned during process teardown
    return *p;
}
static BlockPool &host_pool() {
    static BlockPool *p = new BlockPool();
    return *p;
}

static size_t pool_cap(const char *what, size_t dflt_mb) {
    const char *e = getenv("GRAPHLILY_POOL_MAX_MB");
    (void)what;
    return (size_t)(e ? atol(e) : (long)dflt_mb) << 20;
}

static inline size_t round_block(size_t bytes) {
    const size_t q = bytes >= (1u << 20) ? 4096u : 256u;
    return (bytes + q - 1) / q * q;
}

constexpr size_t kPinThreshold = 64u << 10;   // host blocks below this are not worth a page-locked mapping

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
    // recycled device blocks are ordered by the stream they were last used on: drain it before moving on
    if (c.stream != (hipStream_t)hip_stream) GL_HIP(hipStreamSynchronize(c.stream));
    c.stream = (hipStream_t)hip_stream;
    return GL_OK;
}

int gl_reset_stream(void) {
    GL_REQUIRE_INIT();
    gl::Context &c = gl::ctx();
    if (c.stream != c.own_stream) GL_HIP(hipStreamSynchronize(c.stream));
    c.stream = c.own_stream;
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
    gl::BlockPool &P = gl::device_pool();
    const size_t want = gl::round_block(bytes ? bytes : 4);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.cached.find(want);
        if (it != P.cached.end()) {
            *d_ptr = it->second;
            P.cached.erase(it);
            P.cached_bytes -= want;
            P.live[*d_ptr] = want;
            return GL_OK;
        }
    }
    hipError_t e = hipMalloc(d_ptr, want);
    if (e != hipSuccess) {   // out of memory with blocks parked in the pool: release them and try once more
        (void)hipGetLastError();
        gl_pool_trim();
        e = hipMalloc(d_ptr, want);
    }
    if (e != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_buf_alloc: hipMalloc(%zu): %s", want, hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[*d_ptr] = want;
    return GL_OK;
}

int gl_buf_free(void *d_ptr) {
    GL_REQUIRE_INIT();
    if (!d_ptr) return GL_OK;
    gl::BlockPool &P = gl::device_pool();
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("device", 8192);
        auto it = P.live.find(d_ptr);
        if (it != P.live.end()) {
            sz = it->second;
            P.live.erase(it);
            if (P.cached_bytes + sz <= P.cap_bytes) {
                P.cached.emplace(sz, d_ptr);
                P.cached_bytes += sz;
                return GL_OK;
            }
        }
    }
    GL_HIP(hipFree(d_ptr));   // not from gl_buf_alloc, or the pool is full
    return GL_OK;
}

int gl_pool_trim(void) {
    for (int host = 0; host < 2; host++) {
        gl::BlockPool &P = host ? gl::host_pool() : gl::device_pool();
        std::multimap<size_t, void *> drop;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            drop.swap(P.cached);
            P.cached_bytes = 0;
        }
        if (drop.empty()) continue;
        if (!host && gl::ctx().initialized) (void)hipStreamSynchronize(gl::ctx().stream);   // queued work may still use them
        for (auto &kv : drop) {
            if (host) (void)hipHostFree(kv.second);
            else (void)hipFree(kv.second);
        }
    }
    return GL_OK;
}

int gl_host_pool_alloc(void **h_ptr, size_t bytes) {
    GL_ARG(h_ptr != nullptr);
    *h_ptr = nullptr;
    if (bytes == 0) bytes = 1;
    if (bytes >= gl::kPinThreshold && gl::ctx().initialized) {
        gl::BlockPool &P = gl::host_pool();
        const size_t want = gl::round_block(bytes);
        {
            std::lock_guard<std::mutex> lk(P.mu);
            auto it = P.cached.find(want);
            if (it != P.cached.end()) {
                *h_ptr = it->second;
                P.cached.erase(it);
                P.cached_bytes -= want;
                P.live[*h_ptr] = want;
                return GL_OK;
            }
        }
        if (hipHostMalloc(h_ptr, want, hipHostMallocDefault) == hipSuccess) {
            std::lock_guard<std::mutex> lk(P.mu);
            P.live[*h_ptr] = want;
            return GL_OK;
        }
        (void)hipGetLastError();   // cannot pin (limits): plain pages below
        *h_ptr = nullptr;
    }
    if (posix_memalign(h_ptr, 4096, bytes) != 0) {
        *h_ptr = nullptr;
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_host_pool_alloc: out of host memory (%zu bytes)", bytes);
    }
    return GL_OK;
}

int gl_host_pool_free(void *h_ptr) {
    if (!h_ptr) return GL_OK;
    gl::BlockPool &P = gl::host_pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("host", 2048);
        auto it = P.live.find(h_ptr);
        if (it == P.live.end()) {
            free(h_ptr);   // a posix_memalign block
            return GL_OK;
        }
        const size_t sz = it->second;
        P.live.erase(it);
        if (P.cached_bytes + sz <= P.cap_bytes) {
            P.cached.emplace(sz, h_ptr);
            P.cached_bytes += sz;
            return GL_OK;
        }
    }
    (void)hipHostFree(h_ptr);
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
