// Runtime plumbing of libgraphlily_hip.so: device/stream, error reporting, buffers.
// Replaces the OpenCL/XRT set-up of module/base_module.h:106-133 and the
// cl::Buffer migrate/copy calls of the reference modules.
#include "gl_common.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace gl {

// ------------------------------------------------------------------------------------------ block pools
// The reference's drivers create a fresh cl::Buffer and a fresh page-aligned host vector for every
// send_*_host_to_device / send_*_device_to_host (e.g. app/bfs.h:107-113: two n-element vectors and two buffers per
// BFS::pull call).  hipMalloc / hipFree / hipHostMalloc cost 0.1-3 ms each for 12 MB -- more than the SpMV they
// serve -- so both kinds of block are recycled by exact (rounded) size:
//   device blocks: gl_buf_alloc / gl_buf_free.  Reuse is ordered by the library's stream: a block freed while kernels
//     that read it are still queued is only ever handed to work enqueued later on the same stream (switching
//     streams drains the old one first, gl_set_stream).
//   host blocks: gl_host_pool_alloc / gl_host_pool_free, behind the C++ layer's aligned_allocator.  Blocks of
//     64 KiB and more are recycled, so the drivers' per-call vectors are not mmap'ed and page-faulted in again every
//     time (glibc gives a 12 MB vector fresh pages: ~3000 faults per fill).  They are NOT page-locked by default:
//     measured on the MI355X box (scripts/ubench_host.hip, profiles/r02_ubench_host.txt) a 12 MB copy takes 0.23 ms
//     in either direction from pageable and from page-locked memory alike (56 GB/s), while hipHostMalloc of 12 MB costs
//     2-2.5 ms -- and a driver whose previous result is still alive needs a new block on its first calls.
//     GRAPHLILY_HOST_PIN=1 page-locks them for platforms where pageable copies are staged slowly.
// Cached bytes are capped (GRAPHLILY_POOL_MAX_MB, default 8192 device / 2048 host); gl_pool_trim releases them.
struct BlockPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;             // block -> rounded size | pinned flag (bit 0, host pool)
    std::multimap<size_t, void *> cached;                // rounded size -> free block
    std::unordered_map<void *, size_t> pinned_cached;    // host pool: cached blocks that are page-locked
    std::unordered_map<size_t, bool> grown;              // host pool: sizes that got their one extra block (see gl_host_pool_alloc)
    struct Slab {
        char *base;
        size_t size, used;
    };
    std::vector<Slab> slabs;                             // device pool: blocks are carved from these
    size_t slab_live = 0;                                // carved blocks currently handed out
    bool in_slab(const void *p) const {
        for (const Slab &s : slabs)
            if ((const char *)p >= s.base && (const char *)p < s.base + s.size) return true;
        return false;
    }
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

constexpr size_t kHostPoolThreshold = 64u << 10;   // smaller host blocks go straight to the C library
constexpr size_t kSpareThreshold = 1u << 20;       // a host-pool miss on a block this large also parks one spare
constexpr size_t kSlabBytes = 256u << 20;          // device blocks up to a quarter of this are carved from slabs
static bool pool_trace() {
    static const bool on = getenv("GRAPHLILY_POOL_TRACE") && atoi(getenv("GRAPHLILY_POOL_TRACE")) != 0;
    return on;
}
static bool spare_on_miss() {
    static const bool on = !(getenv("GRAPHLILY_POOL_SPARE") && atoi(getenv("GRAPHLILY_POOL_SPARE")) == 0);
    return on;
}

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

__global__ void fill_u32_gated_kernel(uint32_t *__restrict__ dst, uint32_t v, size_t n, const uint32_t *__restrict__ gate, uint32_t gate_value) {
    if (gate && *gate != gate_value) return;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
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
    int rc;
    if ((rc = gl::preload_spmv()) != GL_OK || (rc = gl::preload_spmv_bool()) != GL_OK || (rc = gl::preload_spmspv()) != GL_OK ||
        (rc = gl::preload_apply()) != GL_OK || (rc = gl::preload_format()) != GL_OK) {
        c.initialized = false;
        return rc;
    }
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

// ---- hipGraph capture of a launch sequence on the library's stream
int gl_graph_begin_capture(void) {
    GL_REQUIRE_INIT();
    if (gl::ctx().stream == nullptr)   // an adopted NULL stream: HIP's legacy default stream cannot be captured
        return gl::set_error(GL_ERR_UNSUPPORTED, "gl_graph_begin_capture: the library is on the NULL stream (gl_set_stream(NULL)); "
                             "capture needs a created stream (gl_reset_stream)");
    GL_HIP(hipStreamBeginCapture(gl::ctx().stream, hipStreamCaptureModeThreadLocal));
    return GL_OK;
}

int gl_graph_end_capture(gl_graph *graph) {
    GL_REQUIRE_INIT();
    GL_ARG(graph != nullptr);
    *graph = nullptr;
    hipGraph_t g = nullptr;
    GL_HIP(hipStreamEndCapture(gl::ctx().stream, &g));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_graph_end_capture: hipGraphInstantiate: %s", hipGetErrorString(e));
    *graph = reinterpret_cast<gl_graph>(exec);
    return GL_OK;
}

int gl_graph_launch(gl_graph graph) {
    GL_REQUIRE_INIT();
    GL_ARG(graph != nullptr);
    GL_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph), gl::ctx().stream));
    return GL_OK;
}

int gl_graph_destroy(gl_graph graph) {
    if (graph) GL_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph)));
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
            P.live[*d_ptr] = want | (P.in_slab(*d_ptr) ? 1u : 0u);
            return GL_OK;
        }
        // Not cached: carve it from a slab.  A genuine hipMalloc costs 0.2 ms (rocprofv3 trace of the reference's
        // bench_bfs on this backend) and a driver that re-sends its vectors on every call (app/bfs.h:107-113) holds
        // the previous call's buffers while it allocates the new ones, so its first calls all miss the cache;
        // carving makes a miss as cheap as a hit.  Blocks above a quarter slab get their own allocation.
        if (want <= gl::kSlabBytes / 4) {
            if (P.slabs.empty() || P.slabs.back().size - P.slabs.back().used < want) {
                void *base = nullptr;
                if (gl::pool_trace()) fprintf(stderr, "[pool] new device slab (%zu MB) for a %zu-byte block\n", gl::kSlabBytes >> 20, want);
                if (hipMalloc(&base, gl::kSlabBytes) == hipSuccess) P.slabs.push_back(gl::BlockPool::Slab{(char *)base, gl::kSlabBytes, 0});
                else (void)hipGetLastError();   // no room for a slab: plain allocation below
            }
            if (!P.slabs.empty() && P.slabs.back().size - P.slabs.back().used >= want) {
                gl::BlockPool::Slab &sl = P.slabs.back();
                *d_ptr = sl.base + sl.used;
                sl.used += want;
                P.live[*d_ptr] = want | 1u;
                P.slab_live++;
                return GL_OK;
            }
        }
    }
    if (gl::pool_trace()) fprintf(stderr, "[pool] device miss %zu bytes\n", want);
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
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("device", 8192);
        auto it = P.live.find(d_ptr);
        if (it != P.live.end()) {
            const size_t sz = it->second & ~(size_t)1;
            const bool slab = (it->second & 1u) != 0;
            P.live.erase(it);
            if (slab) P.slab_live--;
            if (slab || P.cached_bytes + sz <= P.cap_bytes) {   // slab blocks can only be recycled
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
        std::unordered_map<void *, size_t> pinned;
        std::vector<gl::BlockPool::Slab> slabs;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            if (!host && P.slab_live) {   // carved blocks are still out: only the separately allocated ones can go
                for (auto it = P.cached.begin(); it != P.cached.end();) {
                    if (P.in_slab(it->second)) { ++it; continue; }
                    drop.emplace(it->first, it->second);
                    P.cached_bytes -= it->first;
                    it = P.cached.erase(it);
                }
            } else {
                drop.swap(P.cached);
                pinned.swap(P.pinned_cached);
                slabs.swap(P.slabs);
                P.cached_bytes = 0;
            }
        }
        if (drop.empty() && slabs.empty()) continue;
        if (!host && gl::ctx().initialized) (void)hipStreamSynchronize(gl::ctx().stream);   // queued work may still use them
        for (auto &kv : drop) {
            bool carved = false;
            for (const gl::BlockPool::Slab &sl : slabs) carved = carved || ((char *)kv.second >= sl.base && (char *)kv.second < sl.base + sl.size);
            if (carved) continue;
            if (!host) (void)hipFree(kv.second);
            else if (pinned.count(kv.second)) (void)hipHostFree(kv.second);
            else free(kv.second);
        }
        for (const gl::BlockPool::Slab &sl : slabs) (void)hipFree(sl.base);
    }
    return GL_OK;
}

int gl_host_pool_alloc(void **h_ptr, size_t bytes) {
    GL_ARG(h_ptr != nullptr);
    *h_ptr = nullptr;
    if (bytes == 0) bytes = 1;
    if (bytes >= gl::kHostPoolThreshold) {
        gl::BlockPool &P = gl::host_pool();
        const size_t want = gl::round_block(bytes);   // multiple of 4096: bit 0 is free for the pinned flag
        bool hit = false, grow = false;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            auto it = P.cached.find(want);
            if (it != P.cached.end()) {
                *h_ptr = it->second;
                P.cached.erase(it);
                P.cached_bytes -= want;
                P.live[*h_ptr] = want | P.pinned_cached[*h_ptr];
                P.pinned_cached.erase(*h_ptr);
                // A driver call needs one block more from its second call on: the previous call's result is still
                // alive while the new one is allocated (kernel_results = bfs.pull(...), benchmark/bench_bfs.cpp:60).
                // The first time a large size class runs empty it therefore gets one extra paged-in block -- once;
                // the cost (a 12 MB block: ~2 ms of page faults) lands in the call that emptied it, not in the next.
                grow = want >= gl::kSpareThreshold && gl::spare_on_miss() && P.cached.find(want) == P.cached.end() && !P.grown[want];
                if (grow) P.grown[want] = true;
                hit = true;
            }
        }
        if (hit) {
            void *extra = nullptr;
            if (grow && posix_memalign(&extra, 4096, want) == 0) {
                memset(extra, 0, want);
                std::lock_guard<std::mutex> lk(P.mu);
                P.cached.emplace(want, extra);
                P.cached_bytes += want;
            }
            return GL_OK;
        }
        if (gl::pool_trace()) fprintf(stderr, "[pool] host miss %zu bytes\n", want);
        static const bool pin = getenv("GRAPHLILY_HOST_PIN") && atoi(getenv("GRAPHLILY_HOST_PIN")) != 0;
        size_t pinned = 0;
        if (pin && gl::ctx().initialized) {
            if (hipHostMalloc(h_ptr, want, hipHostMallocDefault) == hipSuccess) pinned = 1;
            else {
                (void)hipGetLastError();   // cannot pin (limits): plain pages below
                *h_ptr = nullptr;
            }
        }
        if (!pinned && posix_memalign(h_ptr, 4096, want) != 0) {
            *h_ptr = nullptr;
            return gl::set_error(GL_ERR_INVALID_ARG, "gl_host_pool_alloc: out of host memory (%zu bytes)", want);
        }
        // a miss on a large block parks one spare, and both are paged in now (a fresh 12 MB block costs ~3000 page
        // faults on first touch: 2 ms inside a 2 ms BFS, tests/cpp/api_breakdown.cpp)
        void *spare = nullptr;
        if (!pinned && want >= gl::kSpareThreshold && gl::spare_on_miss()) {
            memset(*h_ptr, 0, want);
            if (posix_memalign(&spare, 4096, want) == 0) memset(spare, 0, want);
            else spare = nullptr;
        }
        std::lock_guard<std::mutex> lk(P.mu);
        P.live[*h_ptr] = want | pinned;
        if (spare) {
            if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("host", 2048);
            if (P.cached_bytes + want <= P.cap_bytes) {
                P.cached.emplace(want, spare);
                P.cached_bytes += want;
            } else {
                free(spare);
            }
        }
        return GL_OK;
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
    size_t tag;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("host", 2048);
        auto it = P.live.find(h_ptr);
        if (it == P.live.end()) {
            free(h_ptr);   // a small block
            return GL_OK;
        }
        tag = it->second;
        P.live.erase(it);
        const size_t sz = tag & ~(size_t)1;
        if (P.cached_bytes + sz <= P.cap_bytes) {
            P.cached.emplace(sz, h_ptr);
            if (tag & 1) P.pinned_cached[h_ptr] = 1;
            P.cached_bytes += sz;
            return GL_OK;
        }
    }
    if (tag & 1) (void)hipHostFree(h_ptr);
    else free(h_ptr);
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

int gl_buf_d2h_async(void *h_dst, const void *d_src, size_t bytes) {
    GL_REQUIRE_INIT();
    if (bytes == 0) return GL_OK;
    GL_ARG(h_dst != nullptr && d_src != nullptr);
    GL_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, gl::ctx().stream));
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

int gl_buf_fill_u32_gated(uint32_t *d_dst, uint32_t value, size_t count, const uint32_t *d_gate, uint32_t gate_value) {
    GL_REQUIRE_INIT();
    if (count == 0) return GL_OK;
    GL_ARG(d_dst != nullptr);
    unsigned blocks = gl::cdiv(count, 256);
    if (blocks > 2048) blocks = 2048;
    gl::fill_u32_gated_kernel<<<blocks, 256, 0, gl::ctx().stream>>>(d_dst, value, count, d_gate, gate_value);
    GL_LAUNCH_CHECK();
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
