// Runtime plumbing of libgraphlily_hip.so: device/stream, error reporting, buffers.
// Replaces the OpenCL/XRT set-up of module/base_module.h:106-133 and the
// cl::Buffer migrate/copy calls of the reference modules.
#include "gl_common.h"
#include "gl_bfs_shard.h"

#include <chrono>
#include <omp.h>
#include <unistd.h>
#include <emmintrin.h>
#include <sys/mman.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sched.h>
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
//     (page-locking them buys nothing on this platform).
// Cached bytes are capped (GRAPHLILY_POOL_MAX_MB, default 8192 device / 2048 host); gl_pool_trim releases them.
struct BlockPool {
    std::mutex mu;
    // Device blocks up to a quarter slab are carved from 256 MB slabs (a genuine hipMalloc costs 0.2 ms).  A slab counts
    // the blocks it has handed out; when the last one comes back its cached pieces are dropped and carving restarts at
    // offset 0 (that is the only coalescing there is, and the only point at which a slab can be released).  A slab is
    // RETIRED when the library moves to another device: its blocks are then never cached again and it is released with
    // its last live block.
    struct Slab {
        char *base;
        size_t size, used, live;
        int device;
        bool retired;
    };
    struct Live {
        size_t size;     // the block's own (rounded) size, which may exceed what the caller asked for (best fit)
        Slab *slab;      // nullptr: its own allocation
        int device;
        bool pinned;     // host pool: page-locked
    };
    std::unordered_map<void *, Live> live;
    std::multimap<size_t, void *> cached;                // rounded size -> free block
    std::unordered_map<void *, size_t> pinned_cached;    // host pool: cached blocks that are page-locked
    std::unordered_map<size_t, bool> grown;              // host pool: sizes that got their one extra block (see gl_host_pool_alloc)
    std::vector<Slab *> slabs;
    Slab *slab_of(const void *p) const {
        for (Slab *s : slabs)
            if ((const char *)p >= s->base && (const char *)p < s->base + s->size) return s;
        return nullptr;
    }
    // drop the cached pieces of a slab (caller holds mu)
    void purge_slab(const Slab *sl) {
        for (auto it = cached.begin(); it != cached.end();) {
            if ((char *)it->second >= sl->base && (char *)it->second < sl->base + sl->size) {
                cached_bytes -= it->first;
                it = cached.erase(it);
            } else {
                ++it;
            }
        }
    }
    size_t slab_bytes() const {
        size_t b = 0;
        for (const Slab *s : slabs) b += s->size;
        return b;
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
static bool pool_trace() { return false; }      // (compile-time switch for allocation debugging)
static bool spare_on_miss() { return true; }

void device_pool_leave_device();

Context &ctx() {
    static Context c;
    return c;
}

Profiler &prof() {
    static Profiler p;
    return p;
}

// ---- GRAPHLILY_TRACE_API
namespace {
struct TraceRec {
    const char *name;
    double t0, t1;
    unsigned long long arg;
};
std::vector<TraceRec> &trace_recs() {
    static std::vector<TraceRec> *v = new std::vector<TraceRec>();
    return *v;
}
std::mutex g_trace_mu;
void trace_dump() {
    const char *path = getenv("GRAPHLILY_TRACE_API");
    FILE *f = (path && strcmp(path, "1") != 0 && strcmp(path, "stderr") != 0) ? fopen(path, "w") : stderr;
    if (!f) return;
    std::lock_guard<std::mutex> lk(g_trace_mu);
    double prev = 0.0;
    for (const TraceRec &r : trace_recs()) {
        fprintf(f, "%12.1f  dur %9.1f  gap %9.1f  %-36s %llu\n", r.t0, r.t1 - r.t0, r.t0 - prev, r.name, r.arg);
        prev = r.t1;
    }
    if (f != stderr) fclose(f);
}
}  // namespace

bool ApiTrace::on() {
    static const bool v = [] {
        const char *e = getenv("GRAPHLILY_TRACE_API");
        const bool yes = e && *e && strcmp(e, "0") != 0;
        if (yes) atexit(trace_dump);
        return yes;
    }();
    return v;
}
double ApiTrace::now_us() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
void ApiTrace::record(const char *name, double t0, double t1, unsigned long long arg) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    trace_recs().push_back(TraceRec{name, t0, t1, arg});
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

// gl_levels_pack: BFS levels (floats holding small integers) as bytes or nibbles, one 32-bit word per thread and store
// (BITS = 8: 4 levels per word, BITS = 4: 8 levels); workgroup 0 also copies `tail_words` raw words behind the packed levels
template <int BITS>
__global__ __launch_bounds__(256) void levels_pack_kernel(const float4 *__restrict__ src, uint32_t *__restrict__ dst, uint32_t nwords,
                                                          const uint32_t *__restrict__ tail, uint32_t tail_words, uint32_t tail_at) {
    constexpr uint32_t M = (1u << BITS) - 1u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nwords; i += gridDim.x * 256u) {
        if (BITS == 8) {
            const float4 v = src[i];
            dst[i] = ((uint32_t)v.x & M) | (((uint32_t)v.y & M) << 8) | (((uint32_t)v.z & M) << 16) | (((uint32_t)v.w & M) << 24);
        } else {
            const float4 v = src[2u * i], w = src[2u * i + 1u];
            dst[i] = ((uint32_t)v.x & M) | (((uint32_t)v.y & M) << 4) | (((uint32_t)v.z & M) << 8) | (((uint32_t)v.w & M) << 12) |
                     (((uint32_t)w.x & M) << 16) | (((uint32_t)w.y & M) << 20) | (((uint32_t)w.z & M) << 24) | (((uint32_t)w.w & M) << 28);
        }
    }
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < tail_words; i += 256u) dst[tail_at + i] = tail[i];
}

constexpr uint32_t kLevelsChunkWords = GL_LEVELS_CHUNK_WORDS;
// gl_levels_pack_stream (the device functions are gl_bfs_shard.h's): workgroup c packs chunk c, the last one delivers the tail words
template <int BITS>
__global__ __launch_bounds__(256) void levels_pack_stream_kernel(LevelsPack k) {
    if (blockIdx.x + 1u < k.nchunks) levels_pack_chunk<BITS>(k, blockIdx.x);
    else levels_pack_tail(k);
}

// gl_buf_d2h_levels: the same packing for a buffer that is only EXPECTED to hold small integers -- every value is checked
// (a non-negative integer no larger than the field allows), a violation raises flag[0] and the caller falls back to the floats
template <int BITS>
__global__ __launch_bounds__(256) void levels_pack_checked_kernel(const float4 *__restrict__ src, uint32_t *__restrict__ dst, uint32_t nwords,
                                                                  uint32_t *__restrict__ flag) {
    constexpr uint32_t M = (1u << BITS) - 1u;
    bool bad = false;
    auto field = [&](float v) -> uint32_t {
        const uint32_t u = v >= 0.0f && v <= (float)M ? (uint32_t)v : 0u;
        bad = bad || __float_as_uint((float)u) != __float_as_uint(v);   // (bit for bit: -0.0 is not a level either)
        return u;
    };
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nwords; i += gridDim.x * 256u) {
        if (BITS == 8) {
            const float4 v = src[i];
            dst[i] = field(v.x) | (field(v.y) << 8) | (field(v.z) << 16) | (field(v.w) << 24);
        } else {
            const float4 v = src[2u * i], w = src[2u * i + 1u];
            dst[i] = field(v.x) | (field(v.y) << 4) | (field(v.z) << 8) | (field(v.w) << 12) | (field(w.x) << 16) | (field(w.y) << 20) |
                     (field(w.z) << 24) | (field(w.w) << 28);
        }
    }
    if (__any(bad) && (threadIdx.x & 63u) == 0u) flag[0] = 1u;
}

// gl_buf_d2h_async: device -> page-locked host memory by stores over PCIe (16 bytes per lane, grid-stride)
__global__ __launch_bounds__(256) void copy_out_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}

__global__ void fill_u32_gated_kernel(uint32_t *__restrict__ dst, uint32_t v, size_t n, const uint32_t *__restrict__ gate, uint32_t gate_value) {
    if (gate && *gate != gate_value) return;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = v;
}

// gl_init is moving to another device: drain the stream, release what is cached, and retire the slabs that still have
// live blocks (they are released with their last block, gl_buf_free)
void device_pool_leave_device() {
    if (ctx().stream || ctx().own_stream) (void)hipStreamSynchronize(ctx().stream);
    gl_pool_trim();
    BlockPool &P = device_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    for (BlockPool::Slab *sl : P.slabs) {
        sl->retired = true;
        P.purge_slab(sl);
    }
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

// ---- host threads next to the device.  A blocking call of this library ends on the host -- the wait for the stream, the
// expansion of a packed BFS result on a few threads, copies out of page-locked memory -- and on a two-socket host that half costs
// 40-100 us more from the far socket than from the GPU's own (orkut's BFS, same box, process by process: 0.41-0.49 ms unbound, 0.36-
// 0.39 bound; profiles/r06_bfs_numa_and_streamed_readback.txt).  gl_init therefore restricts the CALLING thread (and, through
// inheritance, the threads it creates later: the OpenMP team of the expansion) to the CPUs of the device's NUMA node, within the
// affinity mask the process had at the first call; GRAPHLILY_BIND_NUMA=0 leaves the affinity alone.
namespace gl {
struct NearCpus {
    bool have_original = false, bound = false;
    cpu_set_t original, near;
    int node = -1, count = 0;
    unsigned generation = 0;
};
static NearCpus &near_cpus() {
    static NearCpus n;
    return n;
}
// team threads that existed before the binding follow it at their next use
static void bind_this_thread_once() {
    NearCpus &n = near_cpus();
    static thread_local unsigned seen = 0;
    if (!n.bound || seen == n.generation) return;
    seen = n.generation;
    (void)sched_setaffinity(0, sizeof(n.near), &n.near);
}
}  // namespace gl

int gl_host_bind_near_device(int *numa_node, int *cpus) {
    GL_REQUIRE_INIT();
    if (numa_node) *numa_node = -1;
    if (cpus) *cpus = 0;
    gl::NearCpus &n = gl::near_cpus();
    if (!n.have_original) {
        CPU_ZERO(&n.original);
        if (sched_getaffinity(0, sizeof(n.original), &n.original) != 0) return GL_OK;
        n.have_original = true;
    }
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof(id) - 1, gl::ctx().device) != hipSuccess) {
        (void)hipGetLastError();
        return GL_OK;
    }
    for (char *q = id; *q; q++) *q = (char)tolower((unsigned char)*q);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id);
    int node = -1;
    if (FILE *f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return GL_OK;   // (one node, or the platform does not say)
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    char list[4096] = {0};
    if (FILE *f = fopen(path, "r")) {
        if (!fgets(list, sizeof(list), f)) list[0] = 0;
        fclose(f);
    }
    cpu_set_t want;
    CPU_ZERO(&want);
    for (const char *q = list; *q;) {   // "0-63,128-191"
        char *end = nullptr;
        const long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        if (*end == '-') {
            q = end + 1;
            b = strtol(q, &end, 10);
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &n.original)) CPU_SET((int)c, &want);
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',' ) break;
    }
    const int count = CPU_COUNT(&want);
    if (count == 0 || count == CPU_COUNT(&n.original)) return GL_OK;   // nothing of the node is allowed, or nothing else is
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return GL_OK;
    n.near = want;
    n.node = node;
    n.count = count;
    n.bound = true;
    n.generation++;
    if (numa_node) *numa_node = node;
    if (cpus) *cpus = count;
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
    if (c.initialized) gl::device_pool_leave_device();   // cached blocks of the old device must not be handed out on the new one
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
    if (!(getenv("GRAPHLILY_BIND_NUMA") && atoi(getenv("GRAPHLILY_BIND_NUMA")) == 0)) (void)gl_host_bind_near_device(nullptr, nullptr);
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

int gl_prof_begin(uint32_t max_launches, uint32_t every) {
    GL_REQUIRE_INIT();
    gl::Profiler &p = gl::prof();
    p.every = every ? every : 1u;
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

// ---- GPU time of whatever is enqueued between the two calls (one event pair on the library's stream)
static hipEvent_t g_span[2] = {nullptr, nullptr};
int gl_span_begin(void) {
    GL_REQUIRE_INIT();
    for (hipEvent_t &e : g_span)
        if (!e) GL_HIP(hipEventCreate(&e));
    GL_HIP(hipEventRecord(g_span[0], gl::ctx().stream));
    return GL_OK;
}

int gl_span_end(double *ms) {
    GL_REQUIRE_INIT();
    GL_ARG(ms != nullptr && g_span[1] != nullptr);
    GL_HIP(hipEventRecord(g_span[1], gl::ctx().stream));
    GL_HIP(hipEventSynchronize(g_span[1]));
    float t = 0.0f;
    GL_HIP(hipEventElapsedTime(&t, g_span[0], g_span[1]));
    *ms = t;
    return GL_OK;
}

// ---- hipGraph capture of a launch sequence on the library's stream
struct gl_graph_s {
    hipGraphExec_t exec = nullptr;
    std::vector<int *> comm_refs;   // counters of the communicators whose exchanges this graph replays
};

int gl_graph_begin_capture(void) {
    GL_REQUIRE_INIT();
    if (gl::ctx().stream == nullptr)   // an adopted NULL stream: HIP's legacy default stream cannot be captured
        return gl::set_error(GL_ERR_UNSUPPORTED, "gl_graph_begin_capture: the library is on the NULL stream (gl_set_stream(NULL)); "
                             "capture needs a created stream (gl_reset_stream)");
    GL_HIP(hipStreamBeginCapture(gl::ctx().stream, hipStreamCaptureModeThreadLocal));
    gl::ctx().capturing = true;
    gl::ctx().capture_refs.clear();
    return GL_OK;
}

int gl_graph_end_capture(gl_graph *graph) {
    GL_REQUIRE_INIT();
    GL_ARG(graph != nullptr);
    *graph = nullptr;
    gl::ctx().capturing = false;
    hipGraph_t g = nullptr;
    GL_HIP(hipStreamEndCapture(gl::ctx().stream, &g));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_graph_end_capture: hipGraphInstantiate: %s", hipGetErrorString(e));
    // the first launch of a fresh executable graph otherwise uploads it: the reference's bench drivers time exactly ONE call after
    // one warm-up (benchmark/bench_bfs.cpp:59-66), i.e. the first replay of what the warm-up call recorded
    if (gl::debug_knob("graph_upload", 1) != 0 && hipGraphUpload(exec, gl::ctx().stream) != hipSuccess) (void)hipGetLastError();
    gl_graph_s *G = new gl_graph_s;
    G->exec = exec;
    G->comm_refs.swap(gl::ctx().capture_refs);
    for (int *r : G->comm_refs) ++*r;
    *graph = G;
    return GL_OK;
}

int gl_graph_launch(gl_graph graph) {
    GL_REQUIRE_INIT();
    GL_ARG(graph != nullptr);
    GL_HIP(hipGraphLaunch(graph->exec, gl::ctx().stream));
    gl::ctx().graph_launches++;   // completion records of runs enqueued before the replay no longer describe their plans' last run
    return GL_OK;
}

int gl_graph_destroy(gl_graph graph) {
    if (!graph) return GL_OK;
    const hipError_t e = hipGraphExecDestroy(graph->exec);
    for (int *r : graph->comm_refs) --*r;
    delete graph;
    if (e != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_graph_destroy: %s", hipGetErrorString(e));
    return GL_OK;
}

int gl_sync(void) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_HIP(hipStreamSynchronize(gl::ctx().stream));
    return GL_OK;
}

int gl_buf_alloc(void **d_ptr, size_t bytes) {
    GL_TRACE(bytes);
    GL_REQUIRE_INIT();
    GL_ARG(d_ptr != nullptr);
    *d_ptr = nullptr;
    gl::BlockPool &P = gl::device_pool();
    const size_t want = gl::round_block(bytes ? bytes : 4);
    const int dev = gl::ctx().device;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("device", 8192);
        // best fit with bounded waste: the smallest cached block of want .. want + 25 %
        auto it = P.cached.lower_bound(want);
        if (it != P.cached.end() && it->first <= want + want / 4) {
            const size_t have = it->first;
            *d_ptr = it->second;
            P.cached.erase(it);
            P.cached_bytes -= have;
            gl::BlockPool::Slab *sl = P.slab_of(*d_ptr);
            if (sl) sl->live++;
            P.live[*d_ptr] = gl::BlockPool::Live{have, sl, dev, false};
            return GL_OK;
        }
        // Not cached: carve it from a slab.  A driver that re-sends its vectors on every call (app/bfs.h:107-113) holds
        // the previous call's buffers while it allocates the new ones, so its first calls all miss the cache;
        // carving makes a miss as cheap as a hit.  Blocks above a quarter slab get their own allocation, and so does
        // everything once the slabs have reached the pool's cap.
        if (want <= gl::kSlabBytes / 4) {
            gl::BlockPool::Slab *sl = nullptr;
            for (gl::BlockPool::Slab *c : P.slabs)
                if (!c->retired && c->device == dev && c->size - c->used >= want) sl = c;
            if (!sl && P.slab_bytes() + gl::kSlabBytes <= P.cap_bytes) {
                void *base = nullptr;
                if (gl::pool_trace()) fprintf(stderr, "[pool] new device slab (%zu MB) for a %zu-byte block\n", gl::kSlabBytes >> 20, want);
                if (hipMalloc(&base, gl::kSlabBytes) == hipSuccess) {
                    sl = new gl::BlockPool::Slab{(char *)base, gl::kSlabBytes, 0, 0, dev, false};
                    P.slabs.push_back(sl);
                } else {
                    (void)hipGetLastError();   // no room for a slab: plain allocation below
                }
            }
            if (sl) {
                *d_ptr = sl->base + sl->used;
                sl->used += want;
                sl->live++;
                P.live[*d_ptr] = gl::BlockPool::Live{want, sl, dev, false};
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
    P.live[*d_ptr] = gl::BlockPool::Live{want, nullptr, dev, false};
    return GL_OK;
}

int gl_buf_free(void *d_ptr) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    if (!d_ptr) return GL_OK;
    gl::BlockPool &P = gl::device_pool();
    char *release_slab = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("device", 8192);
        auto it = P.live.find(d_ptr);
        if (it != P.live.end()) {
            const gl::BlockPool::Live b = it->second;
            P.live.erase(it);
            if (gl::BlockPool::Slab *sl = b.slab) {
                sl->live--;
                if (sl->live == 0) {
                    // every block of the slab is back: forget its pieces and start carving from the front again;
                    // a retired slab (another device's) or a surplus one goes back to the driver
                    P.purge_slab(sl);
                    sl->used = 0;
                    size_t same_device = 0;
                    for (const gl::BlockPool::Slab *c : P.slabs) same_device += (!c->retired && c->device == sl->device) ? 1u : 0u;
                    if (sl->retired || same_device > 1) {
                        for (size_t i = 0; i < P.slabs.size(); i++)
                            if (P.slabs[i] == sl) { P.slabs.erase(P.slabs.begin() + i); break; }
                        release_slab = sl->base;
                        delete sl;
                    }
                } else if (!sl->retired) {
                    P.cached.emplace(b.size, d_ptr);   // a piece of a slab can only be recycled
                    P.cached_bytes += b.size;
                }
                d_ptr = nullptr;
            } else if (b.device == gl::ctx().device && P.cached_bytes + b.size <= P.cap_bytes) {
                P.cached.emplace(b.size, d_ptr);
                P.cached_bytes += b.size;
                return GL_OK;
            }
        }
    }
    if (release_slab) {
        (void)hipStreamSynchronize(gl::ctx().stream);   // work queued on the library's stream may still use its blocks
        GL_HIP(hipFree(release_slab));
    }
    if (d_ptr) GL_HIP(hipFree(d_ptr));   // not from gl_buf_alloc, another device's, or the pool is full
    return GL_OK;
}

int gl_pool_trim(void) {
    for (int host = 0; host < 2; host++) {
        gl::BlockPool &P = host ? gl::host_pool() : gl::device_pool();
        std::multimap<size_t, void *> drop;
        std::unordered_map<void *, size_t> pinned;
        std::vector<char *> slabs;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            // slabs without a live block go (with their cached pieces); the pieces of slabs that are still in use stay
            // cached -- they cannot be returned one by one
            for (size_t i = 0; i < P.slabs.size();) {
                gl::BlockPool::Slab *sl = P.slabs[i];
                if (sl->live == 0) {
                    P.purge_slab(sl);
                    slabs.push_back(sl->base);
                    delete sl;
                    P.slabs.erase(P.slabs.begin() + i);
                } else {
                    i++;
                }
            }
            for (auto it = P.cached.begin(); it != P.cached.end();) {
                if (P.slab_of(it->second)) { ++it; continue; }
                drop.emplace(it->first, it->second);
                P.cached_bytes -= it->first;
                it = P.cached.erase(it);
            }
            pinned.swap(P.pinned_cached);
        }
        if (drop.empty() && slabs.empty()) continue;
        if (!host && gl::ctx().initialized) (void)hipStreamSynchronize(gl::ctx().stream);   // queued work may still use them
        for (auto &kv : drop) {
            if (!host) (void)hipFree(kv.second);
            else if (pinned.count(kv.second)) (void)hipHostFree(kv.second);
            else free(kv.second);
        }
        for (char *base : slabs) (void)hipFree(base);
    }
    return GL_OK;
}

/* how many device blocks are out, how many bytes are parked, how many slabs exist (tests, leak hunting) */
int gl_pool_stats(uint64_t *live_blocks, uint64_t *cached_bytes, uint32_t *slabs) {
    gl::BlockPool &P = gl::device_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (live_blocks) *live_blocks = P.live.size();
    if (cached_bytes) *cached_bytes = P.cached_bytes;
    if (slabs) *slabs = (uint32_t)P.slabs.size();
    return GL_OK;
}

namespace gl {
// A fresh host block of `want` bytes, paged in.  Blocks of 2 MB and more are 2 MB-aligned and advised as huge pages: a 12 MB
// vector is then 6 faults instead of 3000 (first touch 0.3 ms instead of 2 ms where transparent huge pages are available).
static void *fresh_host_block(size_t want, bool touch) {
    void *p = nullptr;
    const size_t huge = 2u << 20;
    if (want >= huge) {
        const size_t sz = (want + huge - 1) / huge * huge;
        if (posix_memalign(&p, huge, sz) != 0) return nullptr;
        (void)madvise(p, sz, MADV_HUGEPAGE);
    } else if (posix_memalign(&p, 4096, want) != 0) {
        return nullptr;
    }
    if (touch) {
        const size_t nthreads = want >= (4u << 20) ? 4 : 1;
        (void)nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (long long off = 0; off < (long long)want; off += (1 << 20)) {   // one store per page: the kernel zero-fills on the fault
            const size_t end = std::min<size_t>((size_t)off + (1u << 20), want);
            for (size_t q = (size_t)off; q < end; q += 4096) ((volatile char *)p)[q] = 0;
        }
    }
    return p;
}
}  // namespace gl

int gl_host_pool_reserve(size_t bytes, uint32_t count) {
    if (bytes < gl::kHostPoolThreshold) return GL_OK;
    gl::BlockPool &P = gl::host_pool();
    const size_t want = gl::round_block(bytes);
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(P.mu);
            if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("host", 2048);
            if (P.cached.count(want) >= count || P.cached_bytes + want > P.cap_bytes) return GL_OK;
        }
        void *p = gl::fresh_host_block(want, true);
        if (!p) return gl::set_error(GL_ERR_INVALID_ARG, "gl_host_pool_reserve: out of host memory (%zu bytes)", want);
        std::lock_guard<std::mutex> lk(P.mu);
        P.cached.emplace(want, p);
        P.cached_bytes += want;
    }
}

int gl_host_fill_u32(void *h_dst, uint32_t word, size_t count) {
    GL_ARG(h_dst != nullptr || count == 0);
    uint32_t *d = static_cast<uint32_t *>(h_dst);
    const int nt = count >= (1u << 20) ? 8 : 1;
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (long long i = 0; i < (long long)count; i++) d[i] = word;
    return GL_OK;
}

int gl_host_sparse_to_dense(const gl_idx_val *h_sparse, uint32_t range, uint32_t zero_bits, void *h_dense) {
    GL_ARG(h_sparse != nullptr && (h_dense != nullptr || range == 0));
    uint32_t *d = static_cast<uint32_t *>(h_dense);
    const uint32_t nnz = h_sparse[0].index;
    const int nt = range >= (1u << 20) ? 8 : 1;
    (void)nt;
    // every thread fills its slice of the vector and then stores the entries that fall into it.  That needs the list in
    // strictly ascending order (a SpMSpV result is: binary search for the slice's first entry); any other list -- where a
    // repeated index means "the last one wins" -- is stored sequentially after the parallel fill.
    int ascending = 1;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(&& : ascending)
    for (long long k = 2; k <= (long long)nnz; k++) ascending = ascending && (h_sparse[k].index > h_sparse[k - 1].index);
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const uint32_t lo = (uint32_t)((uint64_t)range * t / T), hi = (uint32_t)((uint64_t)range * (t + 1) / T);
        for (uint32_t i = lo; i < hi; i++) d[i] = zero_bits;
        if (ascending) {
            uint32_t a = 1, b = nnz + 1;
            while (a < b) {
                const uint32_t m = a + (b - a) / 2;
                if (h_sparse[m].index < lo) a = m + 1; else b = m;
            }
            for (uint32_t k = a; k <= nnz && h_sparse[k].index < hi; k++) d[h_sparse[k].index] = __builtin_bit_cast(uint32_t, h_sparse[k].val);
        }
    }
    if (!ascending)
        for (uint32_t k = 1; k <= nnz; k++)
            if (h_sparse[k].index < range) d[h_sparse[k].index] = __builtin_bit_cast(uint32_t, h_sparse[k].val);
    return GL_OK;
}

int gl_host_pool_alloc(void **h_ptr, size_t bytes) {
    GL_TRACE(bytes);
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
                P.live[*h_ptr] = gl::BlockPool::Live{want, nullptr, -1, P.pinned_cached.count(*h_ptr) != 0};
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
            void *extra = grow ? gl::fresh_host_block(want, true) : nullptr;
            if (extra) {
                std::lock_guard<std::mutex> lk(P.mu);
                P.cached.emplace(want, extra);
                P.cached_bytes += want;
            }
            return GL_OK;
        }
        if (gl::pool_trace()) fprintf(stderr, "[pool] host miss %zu bytes\n", want);
        const bool pin = false;   // (page-locking 12 MB costs 2.5 ms and buys nothing on this platform: profiles/r02_ubench_host.txt)
        size_t pinned = 0;
        if (pin && gl::ctx().initialized) {
            if (hipHostMalloc(h_ptr, want, hipHostMallocDefault) == hipSuccess) pinned = 1;
            else {
                (void)hipGetLastError();   // cannot pin (limits): plain pages below
                *h_ptr = nullptr;
            }
        }
        const bool page_in = !pinned && want >= gl::kSpareThreshold && gl::spare_on_miss();
        if (!pinned && (*h_ptr = gl::fresh_host_block(want, page_in)) == nullptr)
            return gl::set_error(GL_ERR_INVALID_ARG, "gl_host_pool_alloc: out of host memory (%zu bytes)", want);
        // a miss on a large block parks one spare, and both are paged in now (a fresh 12 MB block costs ~3000 page
        // faults on first touch: 2 ms inside a 2 ms BFS, tests/cpp/api_breakdown.cpp)
        void *spare = nullptr;
        if (page_in) spare = gl::fresh_host_block(want, true);
        std::lock_guard<std::mutex> lk(P.mu);
        P.live[*h_ptr] = gl::BlockPool::Live{want, nullptr, -1, pinned != 0};
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
    GL_TRACE();
    if (!h_ptr) return GL_OK;
    gl::BlockPool &P = gl::host_pool();
    bool pinned;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.cap_bytes) P.cap_bytes = gl::pool_cap("host", 2048);
        auto it = P.live.find(h_ptr);
        if (it == P.live.end()) {
            free(h_ptr);   // a small block
            return GL_OK;
        }
        pinned = it->second.pinned;
        const size_t sz = it->second.size;
        P.live.erase(it);
        if (P.cached_bytes + sz <= P.cap_bytes) {
            P.cached.emplace(sz, h_ptr);
            if (pinned) P.pinned_cached[h_ptr] = 1;
            P.cached_bytes += sz;
            return GL_OK;
        }
    }
    if (pinned) (void)hipHostFree(h_ptr);
    else free(h_ptr);
    return GL_OK;
}

int gl_buf_h2d(void *d_dst, const void *h_src, size_t bytes) {
    GL_TRACE(bytes);
    GL_REQUIRE_INIT();
    if (bytes == 0) return GL_OK;
    GL_ARG(d_dst != nullptr && h_src != nullptr);
    hipStream_t s = gl::ctx().stream;
    GL_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, s));
    GL_HIP(hipStreamSynchronize(s));
    return GL_OK;
}

int gl_buf_d2h(void *h_dst, const void *d_src, size_t bytes) {
    GL_TRACE(bytes);
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
    // A read-back behind a kernel: hipMemcpyAsync starts ~18 us after the kernel in front of it has ended (measured behind every
    // BFS schedule: the runtime's own copy kernel follows a system-scope barrier), a kernel of this library that stores to the
    // page-locked destination itself starts after the usual ~4 us.  Only for page-locked, 16-byte aligned destinations.
    const bool by_kernel = true;
    if (by_kernel && bytes >= (64u << 10) && (((uintptr_t)h_dst | (uintptr_t)d_src) & 15u) == 0) {
        hipPointerAttribute_t at;
        void *dev_view = nullptr;
        if (hipPointerGetAttributes(&at, h_dst) == hipSuccess && at.type == hipMemoryTypeHost &&
            hipHostGetDevicePointer(&dev_view, h_dst, 0) == hipSuccess && dev_view != nullptr) {
            const size_t n16 = bytes / 16u;
            const unsigned grid = (unsigned)std::min<size_t>((size_t)gl::ctx().num_cus * 4u, (n16 + 255u) / 256u);
            gl::copy_out_kernel<<<grid, 256, 0, gl::ctx().stream>>>(static_cast<uint4 *>(dev_view), static_cast<const uint4 *>(d_src), n16);
            GL_LAUNCH_CHECK();
            const size_t done = n16 * 16u;
            if (done < bytes)
                GL_HIP(hipMemcpyAsync(static_cast<char *>(h_dst) + done, static_cast<const char *>(d_src) + done, bytes - done,
                                      hipMemcpyDeviceToHost, gl::ctx().stream));
            return GL_OK;
        }
        (void)hipGetLastError();   // (a pageable destination: not an error, the plain copy below serves it)
    }
    GL_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, gl::ctx().stream));
    return GL_OK;
}

int gl_levels_pack(const float *d_levels, uint32_t n, int bits, const uint32_t *d_tail, uint32_t tail_words, void *d_out) {
    GL_REQUIRE_INIT();
    GL_ARG(d_levels != nullptr && d_out != nullptr && (bits == 4 || bits == 8) && (n & 7u) == 0);
    GL_ARG((((uintptr_t)d_levels | (uintptr_t)d_out) & 15u) == 0 && (tail_words == 0 || d_tail != nullptr));
    const uint32_t nwords = n / (32u / (uint32_t)bits);
    const uint32_t tail_at = (nwords + 3u) & ~3u;      // (the tail starts on a 16-byte boundary)
    const unsigned grid = std::max(1u, std::min<unsigned>(gl::cdiv(nwords, 256), (unsigned)gl::ctx().num_cus * 8u));
    if (bits == 8)
        gl::levels_pack_kernel<8><<<grid, 256, 0, gl::ctx().stream>>>(reinterpret_cast<const float4 *>(d_levels), static_cast<uint32_t *>(d_out),
                                                                      nwords, d_tail, tail_words, tail_at);
    else
        gl::levels_pack_kernel<4><<<grid, 256, 0, gl::ctx().stream>>>(reinterpret_cast<const float4 *>(d_levels), static_cast<uint32_t *>(d_out),
                                                                      nwords, d_tail, tail_words, tail_at);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// the host half: packed levels -> floats on a few cores (a 12 MB result is 1.5 or 3 MB over PCIe and ~30 us of this).
// gl_host_threads_warm wakes the same threads up front -- call it between enqueueing the GPU work and waiting for it, so
// that the wake-up of a sleeping OpenMP team (tens of microseconds) overlaps the kernels.
static int allowed_cpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) return std::min(c, omp_get_num_procs());
    }
    return omp_get_num_procs();
}

static int host_expand_threads(size_t n) {
    // half of the processors this PROCESS may run on (its affinity mask -- a 2-CPU container on a 256-core host must not start 16
    // spinning threads), at most 16
    static const int hw = getenv("GRAPHLILY_HOST_THREADS") ? std::max(1, atoi(getenv("GRAPHLILY_HOST_THREADS")))
                                                             : std::max(1, std::min(16, allowed_cpus() / 2));   // (measured: 12 MB of
    // floats from 1.5 MB of nibbles in 106 / 71 / 53 / 86 / 190 us on 4 / 8 / 16 / 32 / 64 threads of a box under load)
    return n >= (1u << 18) ? hw : (n >= (1u << 16) ? std::min(hw, 8) : 1);
}

int gl_host_unpack_threads(void) { return host_expand_threads(1u << 20); }

// 16 level bytes -> 16 floats.  How they are stored (levels_destination_is_warm below picks per call):
//   cached  ordinary stores (round 6): at cache speed when the destination's lines are still in the cores' caches,
//           which a result array that is recycled call after call is -- 12 MB over 16 threads are 0.8 MB each, an L2's worth; a
//           line that is not cached costs a read-for-ownership first;
//   stream  non-temporal stores past the caches: no read, but ~10 GB/s per core whatever the caches hold.
// EPYC 9575F, orkut's 3 M nibbles into two alternating page-locked arrays, 16 threads: 61 us streamed, 24 us cached (8 threads
// 65 / 32, 32 threads 38 / 16; profiles/r06_host_unpack_stores.txt).  AMD's CLZERO (claim a line without reading it, then fill
// it) was tried for the uncached case and is two orders of magnitude slower on page-locked memory: not kept.
extern "C++" {
enum { kStoreStream = 0, kStoreCached = 1 };
template <int MODE>
static inline void expand16(const __m128i v, float *dst) {
    const __m128i z = _mm_setzero_si128();
    const __m128i lo = _mm_unpacklo_epi8(v, z), hi = _mm_unpackhi_epi8(v, z);
    const __m128 a = _mm_cvtepi32_ps(_mm_unpacklo_epi16(lo, z)), b = _mm_cvtepi32_ps(_mm_unpackhi_epi16(lo, z));
    const __m128 c = _mm_cvtepi32_ps(_mm_unpacklo_epi16(hi, z)), d = _mm_cvtepi32_ps(_mm_unpackhi_epi16(hi, z));
    if (MODE == kStoreStream) {
        _mm_stream_ps(dst, a), _mm_stream_ps(dst + 4, b), _mm_stream_ps(dst + 8, c), _mm_stream_ps(dst + 12, d);
    } else {
        _mm_store_ps(dst, a), _mm_store_ps(dst + 4, b), _mm_store_ps(dst + 8, c), _mm_store_ps(dst + 12, d);
    }
}

// 32-level blocks [b0, b1) of the packed source -> floats (destination 16-byte aligned)
template <int MODE>
static inline void levels_expand_blocks_mode(float *h_dst, const uint8_t *src, size_t b0, size_t b1, int bits) {
    if (bits == 8) {
        for (size_t b = b0; b < b1; b++) {
            expand16<MODE>(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 32u * b)), h_dst + 32u * b);
            expand16<MODE>(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 32u * b + 16u)), h_dst + 32u * b + 16u);
        }
    } else {
        const __m128i m = _mm_set1_epi8(0x0f);
        for (size_t b = b0; b < b1; b++) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 16u * b));
            const __m128i lo = _mm_and_si128(v, m), hi = _mm_and_si128(_mm_srli_epi16(v, 4), m);
            expand16<MODE>(_mm_unpacklo_epi8(lo, hi), h_dst + 32u * b);          // levels 0 ... 15 of the block, in order
            expand16<MODE>(_mm_unpackhi_epi8(lo, hi), h_dst + 32u * b + 16u);
        }
    }
}
}  // extern "C++"

static inline void levels_expand_blocks(float *h_dst, const uint8_t *src, size_t b0, size_t b1, int bits, bool cached) {
    if (cached) levels_expand_blocks_mode<kStoreCached>(h_dst, src, b0, b1, bits);
    else levels_expand_blocks_mode<kStoreStream>(h_dst, src, b0, b1, bits);
}

// Which stores a destination gets (one decision per call, by the calling thread): ordinary ones when it is one of the last few
// destinations -- an array the caller recycles, whose lines the team's caches still hold -- and streaming ones for memory seen for
// the first time (the reference's bench_bfs times ONE call into a vector allocated for it: 0.45 ms streamed, 0.50 with ordinary
// stores and their read-for-ownership).  GRAPHLILY_HOST_STORES=cached|stream pins either.
static bool levels_destination_is_warm(const float *h_dst) {
    static const char *pin = getenv("GRAPHLILY_HOST_STORES");
    static std::mutex lock;
    static const float *recent[4] = {nullptr, nullptr, nullptr, nullptr};
    static unsigned next = 0;
    if (pin && !strcmp(pin, "stream")) return false;
    if (pin && !strcmp(pin, "cached")) return true;
    std::lock_guard<std::mutex> hold(lock);
    for (const float *q : recent)
        if (q == h_dst) return true;
    recent[next++ & 3u] = h_dst;
    return false;
}

// wait_stream: the team is started FIRST and its master waits for the library's stream while the other threads spin on a
// flag -- a team woken before the wait has gone back to sleep by the time a 0.3 ms schedule ends (measured in the reference's
// bench_bfs: 187 us for the unpack of 12 MB instead of ~55), one woken after it costs its wake-up in full
static int levels_unpack_impl(float *h_dst, const void *h_src, size_t n, int bits, bool wait_stream) {
    const int nt = host_expand_threads(n);
    (void)nt;
    const bool warm = levels_destination_is_warm(h_dst);
    const uint8_t *src = static_cast<const uint8_t *>(h_src);
    int ready = wait_stream ? 0 : 1;
    hipError_t waited = hipSuccess;
    // blocks of 32 levels (16 or 32 source bytes) through SSE2 -- the x86-64 baseline -- when the destination allows aligned
    // streaming stores; the rest (and any other destination) by the plain loop
    const size_t nblk = (((uintptr_t)h_dst & 15u) == 0) ? n / 32u : 0u;
#pragma omp parallel num_threads(nt)
    {
        const size_t T = (size_t)omp_get_num_threads(), t = (size_t)omp_get_thread_num();
        gl::bind_this_thread_once();
        if (wait_stream) {
            if (t == 0) {
                waited = hipStreamSynchronize(gl::ctx().stream);
                __atomic_store_n(&ready, 1, __ATOMIC_RELEASE);
            } else {
                // (spinning is for waits of a fraction of a millisecond; past ~2 ms the thread naps between looks)
                for (uint32_t spins = 0; !__atomic_load_n(&ready, __ATOMIC_ACQUIRE); spins++) {
                    if (spins < (1u << 16)) _mm_pause();
                    else usleep(50);
                }
            }
        }
        const size_t b0 = nblk * t / T, b1 = nblk * (t + 1) / T;
        if (waited == hipSuccess) levels_expand_blocks(h_dst, src, b0, b1, bits, warm);
        _mm_sfence();
    }
    if (waited != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_sync_levels_unpack: %s", hipGetErrorString(waited));
    for (size_t i = nblk * 32u; i < n; i++)
        h_dst[i] = bits == 8 ? (float)src[i] : (float)((src[i / 2u] >> (4u * (i & 1u))) & 15u);
    return GL_OK;
}

int gl_host_levels_unpack(float *h_dst, const void *h_src, size_t n, int bits) {
    GL_ARG(((h_dst != nullptr && h_src != nullptr) || n == 0) && (bits == 4 || bits == 8) && (bits == 8 || (n & 1u) == 0));
    return levels_unpack_impl(h_dst, h_src, n, bits, false);
}

int gl_sync_levels_unpack(float *h_dst, const void *h_src, size_t n, int bits) {
    GL_REQUIRE_INIT();
    GL_ARG(((h_dst != nullptr && h_src != nullptr) || n == 0) && (bits == 4 || bits == 8) && (bits == 8 || (n & 1u) == 0));
    return levels_unpack_impl(h_dst, h_src, n, bits, true);
}

// ---- the streamed form: pack kernel -> page-locked host block in flagged chunks -> the team expands chunks as they land
// block layout (32-bit words): [0, nwords) packed levels | tail_at = nwords rounded up to 4: tail_words raw words | flags_at =
// the next multiple of 16 words: one flag per chunk + one for the tail, GL_LEVELS_FLAG_STRIDE_WORDS apart (a cache line each)
struct LevelsStream {
    uint32_t nwords, tail_at, flags_at, nchunks;   // nchunks counts the tail's
    size_t bytes;
};
static LevelsStream levels_stream_layout(uint32_t n, int bits, uint32_t tail_words) {
    LevelsStream L;
    L.nwords = n / (32u / (uint32_t)bits);
    L.tail_at = (L.nwords + 3u) & ~3u;
    L.flags_at = (L.tail_at + tail_words + 15u) & ~15u;
    L.nchunks = (L.nwords + gl::kLevelsChunkWords - 1u) / gl::kLevelsChunkWords + 1u;
    L.bytes = 4u * ((size_t)L.flags_at + (size_t)L.nchunks * GL_LEVELS_FLAG_STRIDE_WORDS);
    return L;
}

int gl_levels_stream_bytes(uint32_t n, int bits, uint32_t tail_words, size_t *bytes) {
    GL_ARG(bytes != nullptr && (bits == 4 || bits == 8) && (n & 7u) == 0);
    *bytes = levels_stream_layout(n, bits, tail_words).bytes;
    return GL_OK;
}

int gl_levels_stream_arm(void *h_block, uint32_t n, int bits, uint32_t tail_words) {
    GL_ARG(h_block != nullptr && (bits == 4 || bits == 8) && (n & 7u) == 0);
    const LevelsStream L = levels_stream_layout(n, bits, tail_words);
    uint32_t *flags = static_cast<uint32_t *>(h_block) + L.flags_at;
    for (uint32_t c = 0; c < L.nchunks; c++) __atomic_store_n(flags + (size_t)c * GL_LEVELS_FLAG_STRIDE_WORDS, 0u, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);   // (the launch that follows is what orders these against the kernel's stores)
    return GL_OK;
}

}  // extern "C"
namespace gl {
int levels_stream_describe(const float *d_levels, uint32_t n, int bits, const uint32_t *d_tail, uint32_t tail_words, void *h_block,
                           LevelsPack *out, const char *who) {
    if (!(d_levels != nullptr && h_block != nullptr && (bits == 4 || bits == 8) && (n & 7u) == 0 &&
          (((uintptr_t)d_levels | (uintptr_t)h_block) & 15u) == 0 && (tail_words == 0 || d_tail != nullptr)))
        return set_error(GL_ERR_INVALID_ARG, "%s: n levels (a multiple of 8) in a 16-byte aligned buffer, bits 4 or 8, a 16-byte aligned block", who);
    hipPointerAttribute_t at;
    void *dev_view = nullptr;
    if (hipPointerGetAttributes(&at, h_block) != hipSuccess || at.type != hipMemoryTypeHost ||
        hipHostGetDevicePointer(&dev_view, h_block, 0) != hipSuccess || dev_view == nullptr) {
        (void)hipGetLastError();
        return set_error(GL_ERR_INVALID_ARG, "%s: the block must be page-locked host memory (gl_host_alloc)", who);
    }
    const LevelsStream L = levels_stream_layout(n, bits, tail_words);
    out->src = reinterpret_cast<const float4 *>(d_levels);
    out->dst = static_cast<uint32_t *>(dev_view);
    out->flags = out->dst + L.flags_at;
    out->tail = d_tail;
    out->nwords = L.nwords, out->bits = (uint32_t)bits, out->tail_words = tail_words, out->tail_at = L.tail_at, out->nchunks = L.nchunks;
    return GL_OK;
}
}  // namespace gl
extern "C" {

int gl_levels_pack_stream(const float *d_levels, uint32_t n, int bits, const uint32_t *d_tail, uint32_t tail_words, void *h_block) {
    GL_REQUIRE_INIT();
    gl::LevelsPack k;
    const int rc = gl::levels_stream_describe(d_levels, n, bits, d_tail, tail_words, h_block, &k, "gl_levels_pack_stream");
    if (rc != GL_OK) return rc;
    if (bits == 8) gl::levels_pack_stream_kernel<8><<<k.nchunks, 256, 0, gl::ctx().stream>>>(k);
    else gl::levels_pack_stream_kernel<4><<<k.nchunks, 256, 0, gl::ctx().stream>>>(k);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_sync_levels_unpack_stream(float *h_dst, const void *h_block, size_t n, int bits, uint32_t *h_tail, uint32_t tail_words) {
    GL_REQUIRE_INIT();
    GL_ARG(((h_dst != nullptr && h_block != nullptr) || n == 0) && (bits == 4 || bits == 8) && (n & 7u) == 0 && n <= 0xffffffffull);
    GL_ARG(tail_words == 0 || h_tail != nullptr);
    const LevelsStream L = levels_stream_layout((uint32_t)n, bits, tail_words);
    const uint32_t *words = static_cast<const uint32_t *>(h_block);
    const uint8_t *src = static_cast<const uint8_t *>(h_block);
    const uint32_t *flags = words + L.flags_at;
    const int nt = host_expand_threads(n);
    (void)nt;
    const bool aligned = ((uintptr_t)h_dst & 15u) == 0;
    const bool warm = levels_destination_is_warm(h_dst);
    const uint32_t levels_per_word = 32u / (uint32_t)bits;
    int drained = 0, missing = 0;
    hipError_t waited = hipSuccess;
    const bool stamps = gl::debug_knob("levels_stream_stamps", 0) != 0;   // scratch: when the chunks were done / the stream had drained
    const auto t_in = std::chrono::steady_clock::now();
    double t_sync = 0.0, t_last = 0.0, t_first = 0.0;
    // thread 0 waits for the stream (and so learns of a failed launch: the flags of a schedule that died never come); the others
    // take the chunks round-robin, each as soon as its flag is up.  With one thread: wait, then expand everything.
#pragma omp parallel num_threads(nt)
    {
        const uint32_t T = (uint32_t)omp_get_num_threads(), t = (uint32_t)omp_get_thread_num();
        gl::bind_this_thread_once();
        if (t == 0) {
            waited = hipStreamSynchronize(gl::ctx().stream);
            __atomic_store_n(&drained, 1, __ATOMIC_RELEASE);
            if (stamps) t_sync = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count();
        }
        const uint32_t W = T > 1u ? T - 1u : 1u, w = T > 1u ? t - 1u : 0u;
        if (T == 1u || t > 0u) {
            for (uint32_t c = w; c + 1u < L.nchunks; c += W) {
                const uint32_t *f = flags + (size_t)c * GL_LEVELS_FLAG_STRIDE_WORDS;
                bool up = false;
                for (uint32_t spins = 0;; spins++) {
                    if (__atomic_load_n(f, __ATOMIC_ACQUIRE) != 0u) { up = true; break; }
                    if (__atomic_load_n(&drained, __ATOMIC_ACQUIRE)) {   // the stream is done: whatever it wrote is visible now
                        up = __atomic_load_n(f, __ATOMIC_ACQUIRE) != 0u;
                        break;
                    }
                    if (spins < (1u << 16)) _mm_pause();
                    else usleep(50);
                }
                if (!up) {
                    __atomic_store_n(&missing, 1, __ATOMIC_RELAXED);
                    break;
                }
                if (stamps && c == 0) t_first = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count();
                const size_t w0 = (size_t)c * gl::kLevelsChunkWords, w1 = std::min<size_t>(L.nwords, w0 + gl::kLevelsChunkWords);
                const size_t l0 = w0 * levels_per_word, l1 = w1 * levels_per_word;     // (a chunk holds whole 32-level blocks)
                if (aligned) {
                    levels_expand_blocks(h_dst, src, l0 / 32u, l1 / 32u, bits, warm);
                    for (size_t i = l1 / 32u * 32u; i < l1; i++)
                        h_dst[i] = bits == 8 ? (float)src[i] : (float)((src[i / 2u] >> (4u * (i & 1u))) & 15u);
                } else {
                    for (size_t i = l0; i < l1; i++) h_dst[i] = bits == 8 ? (float)src[i] : (float)((src[i / 2u] >> (4u * (i & 1u))) & 15u);
                }
            }
            _mm_sfence();
            if (stamps && (L.nchunks - 2u) % W == w) t_last = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count();
        }
    }
    if (stamps)
        fprintf(stderr, "levels stream: first chunk up at %.1f us, last chunk expanded at %.1f us, stream drained at %.1f us (from the call)\n", t_first,
                t_last, t_sync);
    if (waited != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_sync_levels_unpack_stream: %s", hipGetErrorString(waited));
    if (missing || __atomic_load_n(flags + (size_t)(L.nchunks - 1u) * GL_LEVELS_FLAG_STRIDE_WORDS, __ATOMIC_ACQUIRE) == 0u)
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_sync_levels_unpack_stream: the stream ended without delivering every chunk "
                                                 "(was gl_levels_pack_stream enqueued for this block?)");
    if (tail_words) memcpy(h_tail, words + L.tail_at, 4u * (size_t)tail_words);
    return GL_OK;
}

int gl_buf_d2h_levels(float *h_dst, const float *d_src, size_t n, float max_level, int *packed) {
    GL_TRACE(n * sizeof(float));
    GL_REQUIRE_INIT();
    if (packed) *packed = 0;
    if (n == 0) return GL_OK;
    GL_ARG(h_dst != nullptr && d_src != nullptr);
    // GRAPHLILY_BFS_U8=0: levels always cross PCIe as floats (the Python drivers read the same switch)
    static const bool on = !(getenv("GRAPHLILY_BFS_U8") && atoi(getenv("GRAPHLILY_BFS_U8")) == 0);
    const int bits = max_level <= 15.0f ? 4 : 8;
    // (with fewer than four host threads the expansion costs more than the PCIe time it saves)
    if (!on || host_expand_threads(1u << 20) < 4 || !(max_level >= 0.0f && max_level <= 255.0f) || n < (1u << 16) || (n & 7u) != 0 ||
        ((uintptr_t)d_src & 15u) != 0)
        return gl_buf_d2h(h_dst, d_src, n * sizeof(float));
    hipStream_t s = gl::ctx().stream;
    const uint32_t nwords = (uint32_t)(n / (32u / (uint32_t)bits));
    const size_t pbytes = ((size_t)nwords * 4u + 15u) & ~(size_t)15u, total = pbytes + 16u;   // packed levels, then the flag word
    // one page-locked staging block of the library, grown on demand (this call is blocking: nobody else uses it meanwhile)
    static void *stage = nullptr;
    static size_t stage_bytes = 0;
    static std::mutex stage_lock;                       // (two host threads downloading at once take turns)
    std::lock_guard<std::mutex> hold(stage_lock);
    if (stage_bytes < total) {
        if (stage) (void)hipHostFree(stage);
        stage = nullptr, stage_bytes = 0;
        if (hipHostMalloc(&stage, total, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return gl_buf_d2h(h_dst, d_src, n * sizeof(float));
        }
        stage_bytes = total;
    }
    void *d_pack = nullptr;
    int rc = gl_buf_alloc(&d_pack, total);
    if (rc != GL_OK) return rc;
    uint32_t *d_flag = reinterpret_cast<uint32_t *>(static_cast<char *>(d_pack) + pbytes);
    hipError_t e = hipMemsetAsync(d_flag, 0, 16, s);
    if (e == hipSuccess) {
        const unsigned grid = std::max(1u, std::min<unsigned>(gl::cdiv(nwords, 256), (unsigned)gl::ctx().num_cus * 8u));
        if (bits == 8)
            gl::levels_pack_checked_kernel<8><<<grid, 256, 0, s>>>(reinterpret_cast<const float4 *>(d_src), static_cast<uint32_t *>(d_pack), nwords, d_flag);
        else
            gl::levels_pack_checked_kernel<4><<<grid, 256, 0, s>>>(reinterpret_cast<const float4 *>(d_src), static_cast<uint32_t *>(d_pack), nwords, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        rc = gl_buf_d2h_async(stage, d_pack, total);
        // (the team starts now and unpacks as soon as the stream has drained; a raised flag makes that wasted work)
        if (rc == GL_OK) rc = levels_unpack_impl(h_dst, stage, n, bits, true);
        else (void)hipStreamSynchronize(s);
    } else {
        (void)hipStreamSynchronize(s);
    }
    (void)gl_buf_free(d_pack);
    if (e != hipSuccess) return gl::set_error(GL_ERR_HIP, "gl_buf_d2h_levels: %s", hipGetErrorString(e));
    if (rc != GL_OK) return rc;
    if (*reinterpret_cast<const uint32_t *>(static_cast<const char *>(stage) + pbytes) != 0u)
        return gl_buf_d2h(h_dst, d_src, n * sizeof(float));      // not levels after all: the floats themselves
    if (packed) *packed = 1;
    return GL_OK;
}

int gl_buf_d2d(void *d_dst, const void *d_src, size_t bytes) {
    GL_TRACE(bytes);
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

int gl_buf_fill_u32(uint32_t *d_dst, uint32_t value, size_t count) {
    const uint32_t *d_gate = nullptr;
    const uint32_t gate_value = 0u;
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
    GL_TRACE(count);
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
