// Internal helpers shared by the translation units of libgraphlily_hip.so.
// gfx950 (MI355X, CDNA4) only: wavefront = 64 lanes everywhere.
#ifndef GL_COMMON_H_
#define GL_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "graphlily_hip.h"

// launch predicates and step flags of the kernels' internal interfaces (what is left of the gated schedules of rounds 2-3:
// the compaction and the boolean pull step still take a predicate, nobody outside the library passes one)
#define GL_GATE_EQ 0
#define GL_GATE_GT 1
#define GL_GATE_LE 2
#define GL_BFS_DEFERRED 4
#define GL_STEP_PULL_FLAGS 8

namespace gl {

static inline long env_long(const char *name, long dflt) {
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
}

// Planner overrides for tests and same-box A/B runs -- NOT feature switches: GRAPHLILY_DEBUG="key=value,key=value" forces a
// decision the planner would otherwise take from the matrix (row blocks x column segments, hot table size, helper mode, load
// width, SpMSpV tile height, the BFS schedule's work thresholds ...), so that a test can drive every code path on one small
// matrix.  Read at the call that takes the decision (plan creation, mostly).
static inline long debug_knob(const char *key, long dflt) {
    const char *e = getenv("GRAPHLILY_DEBUG");
    if (!e) return dflt;
    const size_t klen = strlen(key);
    for (const char *p = e; *p;) {
        const char *end = strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : strlen(p);
        if (len > klen && strncmp(p, key, klen) == 0 && p[klen] == '=') return atol(p + klen + 1);
        if (!end) break;
        p = end + 1;
    }
    return dflt;
}


constexpr int kWave = 64;

// graphlily/global.h:80 / hw/math_constants.h: FLOAT_INF
constexpr float kFloatInf = 999999999.0f;

struct Context {
    bool initialized = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // current stream (own or adopted)
    int num_cus = 256;
    uint32_t *pinned_word = nullptr;   // page-locked staging word for small device->host control reads
    // gl_graph_begin_capture .. gl_graph_end_capture: the communicators (their counters of recorded graphs) whose exchanges
    // were recorded -- RCCL's communicator destroy WAITS for every graph holding its operations, so gl_dist_destroy refuses
    // while such a graph is alive instead of hanging
    bool capturing = false;
    std::vector<int *> capture_refs;
    uint64_t graph_launches = 0;       // gl_graph_launch calls so far (gl_spmspv_wait: is a plan's completion record still its last run's?)
};

Context &ctx();
inline uint64_t graph_launches() { return ctx().graph_launches; }

// bench.py's per-kernel HIP-event timing (gl_prof_begin / gl_prof_end)
struct Profiler {
    bool on = false;
    uint32_t used = 0;
    uint32_t every = 1, seen = 0;    // bracket every `every`-th launch (gl_prof_sample_every)
    std::vector<hipEvent_t> events;  // pairs: start, stop
};
Profiler &prof();
// should this launch of the dominant kernel be bracketed by an event pair?
inline bool prof_take(Profiler &pf) {
    if (!pf.on) return false;
    return (pf.seen++ % pf.every) == 0u && 2ull * (pf.used + 1) <= pf.events.size();
}

int set_error(int code, const char *fmt, ...);
// csr2csc on the host (OpenMP; gl_npz.cpp): what gl_csr2csc does without a device or for small matrices
int host_csr2csc(uint32_t num_rows, uint32_t num_cols, const uint32_t *indptr, const uint32_t *indices, const float *data,
                 uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data);

// GRAPHLILY_TRACE_API=<file>: host-side timeline of the C ABI calls a process makes (name, start, duration in
// microseconds since the first call, an optional size), written at exit -- where an UNMODIFIED reference driver spends its
// time between its module calls (profiles/r03_api_timeline_*.txt).  Off: one predictable branch per call.
struct ApiTrace {
    const char *name;
    double t0;
    unsigned long long arg;
    static bool on();
    static double now_us();
    static void record(const char *name, double t0, double t1, unsigned long long arg);
    explicit ApiTrace(const char *n, unsigned long long a = 0) : name(n), t0(on() ? now_us() : 0.0), arg(a) {}
    ~ApiTrace() {
        if (on()) record(name, t0, now_us(), arg);
    }
};
#define GL_TRACE(...) gl::ApiTrace gl_trace_(__func__, ##__VA_ARGS__)

// one per translation unit with kernels: force the unit's code object onto the device (gl_init)
int preload_spmv();
int preload_spmv_bool();
int preload_spmspv();
int preload_apply();
int preload_format();

#define GL_REQUIRE_INIT()                                                           \
    do {                                                                            \
        if (!gl::ctx().initialized)                                                 \
            return gl::set_error(GL_ERR_NOT_INITIALIZED,                            \
                                 "%s: gl_init() has not succeeded (no HIP device?)", __func__); \
    } while (0)

#define GL_HIP(expr)                                                                \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess)                                                       \
            return gl::set_error(GL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                                 hipGetErrorString(e_));                            \
    } while (0)

#define GL_ARG(cond)                                                                \
    do {                                                                            \
        if (!(cond))                                                                \
            return gl::set_error(GL_ERR_INVALID_ARG, "%s: invalid argument: %s", __func__, #cond); \
    } while (0)

#define GL_LAUNCH_CHECK() GL_HIP(hipGetLastError())

// one device word into a host (stack) variable: the stream is waited for whether or not the copy could be enqueued, so no
// return path leaves a copy into a dead stack frame pending
static inline hipError_t d2h_word_sync(uint32_t *h_dst, const void *d_src, hipStream_t s) {
    const hipError_t e = hipMemcpyAsync(h_dst, d_src, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    const hipError_t w = hipStreamSynchronize(s);
    return e != hipSuccess ? e : w;
}

static inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------ semirings
// The three semirings of graphlily/global.h:96-100 as compile-time policies.
// mul = the (x) ALU, add = the (+) ALU (hw/float_pe.h:36-77).  `ident` is the
// value used for padding lanes inside a reduction (neutral for add), and
// finish() folds the runtime SemiringType::zero in exactly once, reproducing
// "accumulator initialised to zero" of the reference loops.
template <int OP>
struct Semiring;

template <>
struct Semiring<GL_OP_MULADD> {
    __device__ static float ident(float) { return 0.0f; }
    __device__ static float mul(float a, float b) { return a * b; }
    __device__ static float add(float a, float b) { return a + b; }
    __device__ static float finish(float zero, float s) { return zero + s; }
};

template <>
struct Semiring<GL_OP_ANDOR> {
    __device__ static float ident(float) { return 0.0f; }
    __device__ static float mul(float a, float b) { return (a != 0.0f && b != 0.0f) ? 1.0f : 0.0f; }
    __device__ static float add(float a, float b) { return (a != 0.0f || b != 0.0f) ? 1.0f : 0.0f; }
    __device__ static float finish(float zero, float s) { return (zero != 0.0f || s != 0.0f) ? 1.0f : 0.0f; }
};

template <>
struct Semiring<GL_OP_ADDMIN> {
    __device__ static float ident(float) { return __builtin_inff(); }
    __device__ static float mul(float a, float b) { return a + b; }
    // std::min(acc, t) == (t < acc) ? t : acc
    __device__ static float add(float a, float b) { return (b < a) ? b : a; }
    __device__ static float finish(float zero, float s) { return (s < zero) ? s : zero; }
};

// ------------------------------------------------------------------ the reference's other value types
// graphlily/global.h:62-64 lets val_t be `unsigned` or ap_ufixed<32, 8, AP_RND, AP_SAT> (the shipped default) instead of
// float.  Both are 32-bit words, so they travel through the same kernels, buffers and plan layouts as BIT PATTERNS
// inside `float` variables (loads, stores and register moves never change bits); only the semiring policies below and
// the mask tests look inside.  Internally the value type is folded into the op code: opx = op + 3 * val_type.
//   unsigned        C arithmetic: + and * wrap mod 2^32 (associative, so even (+,x) is exact in any order),
//                   a && b / a || b give 1, MIN is the unsigned minimum (hw/ufixed_pe_fwd.h:23-65 with ValT = unsigned);
//   ufixed<32,8>    value = bits / 2^24; a + b saturates at 2^32 - 1 (AP_SAT), a && b / a || b give 1.0 = 1 << 24, MIN is
//                   the unsigned minimum of the bits.  (+,x): the PRODUCT is rounded to 24 fraction bits (AP_RND, half up)
//                   and saturated when it is assigned to ValT (hw/ufixed_pe_fwd.h:29-31: `out = a * b`), the add only
//                   saturates (:53-55).  Every term is non-negative, so a clamped running sum equals
//                   min(sum of the terms, 2^32 - 1) in ANY order: the kernels add the rounded products exactly (64-bit
//                   accumulators in LDS, clamped compare-and-swap adds in global memory) and clamp once.
constexpr int kOpU32MulAdd = 3, kOpU32AndOr = 4, kOpU32AddMin = 5, kOpFixMulAdd = 6, kOpFixAndOr = 7, kOpFixAddMin = 8;
constexpr uint32_t kFixOne = 1u << 24;

__device__ __forceinline__ uint32_t fbits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ float bitsf(uint32_t v) { return __uint_as_float(v); }
__device__ __forceinline__ uint32_t sat_add_u32(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;
    return s < a ? 0xffffffffu : s;
}

template <>
struct Semiring<kOpU32MulAdd> {
    __device__ static float ident(float) { return bitsf(0u); }
    __device__ static float mul(float a, float b) { return bitsf(fbits(a) * fbits(b)); }
    __device__ static float add(float a, float b) { return bitsf(fbits(a) + fbits(b)); }
    __device__ static float finish(float zero, float s) { return bitsf(fbits(zero) + fbits(s)); }
};

// a * b of two ap_ufixed<32,8> assigned to one: 48 fraction bits rounded half up to 24 (AP_RND), saturated (AP_SAT)
__device__ __forceinline__ uint32_t fix_mul_u32(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * (unsigned long long)b;
    const unsigned long long r = (p + (1ull << 23)) >> 24;   // p < 2^64 - 2^33: the rounding add cannot wrap
    return r > 0xffffffffull ? 0xffffffffu : (uint32_t)r;
}
template <>
struct Semiring<kOpFixMulAdd> {
    __device__ static float ident(float) { return bitsf(0u); }
    __device__ static float mul(float a, float b) { return bitsf(fix_mul_u32(fbits(a), fbits(b))); }
    __device__ static float add(float a, float b) { return bitsf(sat_add_u32(fbits(a), fbits(b))); }
    __device__ static float finish(float zero, float s) { return add(zero, s); }
};

template <uint32_t ONE>
struct SemiringBitsAndOr {
    __device__ static float ident(float) { return bitsf(0u); }
    __device__ static float mul(float a, float b) { return bitsf((fbits(a) != 0u && fbits(b) != 0u) ? ONE : 0u); }
    __device__ static float add(float a, float b) { return bitsf((fbits(a) != 0u || fbits(b) != 0u) ? ONE : 0u); }
    __device__ static float finish(float zero, float s) { return add(zero, s); }
};
template <>
struct Semiring<kOpU32AndOr> : SemiringBitsAndOr<1u> {};
template <>
struct Semiring<kOpFixAndOr> : SemiringBitsAndOr<kFixOne> {};

template <bool SAT>
struct SemiringBitsAddMin {
    __device__ static float ident(float) { return bitsf(0xffffffffu); }
    __device__ static float mul(float a, float b) { return bitsf(SAT ? sat_add_u32(fbits(a), fbits(b)) : fbits(a) + fbits(b)); }
    __device__ static float add(float a, float b) { return bitsf(min(fbits(a), fbits(b))); }
    __device__ static float finish(float zero, float s) { return add(zero, s); }
};
template <>
struct Semiring<kOpU32AddMin> : SemiringBitsAddMin<false> {};
template <>
struct Semiring<kOpFixAddMin> : SemiringBitsAddMin<true> {};

// is this word "zero" for a mask test?  float: compares equal to 0.0 (so -0.0 is zero); the integer types: all bits clear
template <int OPX>
__device__ __forceinline__ bool value_is_zero(float v) {
    return OPX < 3 ? (v == 0.0f) : (__float_as_uint(v) == 0u);
}

// ------------------------------------------------------------------ device-resident BFS schedule, frontier as bits only
// (gl_bfs_bits_*): 16 control words + two per slot (the new-frontier count of slot s in word 16 + s, how it was evaluated behind them).  [0] first pull slot (0xffffffff while pushing), [1] push iterations of the first
// push phase (the reference's count), [2] source vertex, [3] pushes after a pull step handed the loop back, [4] the slot
// that handed back (0xffffffff: none), [5] new-frontier count of the running step, [6] workgroup ticket of the running
// step, [7] the threshold the hand-back used, [8] the slot whose PUSH goes row-wise (its frontier's columns hold more
// than `heavy` non-zeros; 0: none), [9] the slot whose pull goes bottom-up, [12..13] non-zeros in the rows reached so far
// (64 bits; ctl is 8-byte aligned).
// Every step of slot s ends with decide(): the reference's loop condition (do { push } while (it < num_iterations &&
// nnz / n < threshold), app/bfs.h:180-190) where the step pushed, the opposite decision where it pulled, and the
// direction of the next slot's push.
struct BfsBitsCtl {
    uint32_t *ctl = nullptr;
    uint32_t slot = 0, n = 1;
    uint32_t may_continue = 0;     // bit 0: the reference's loop may go on after this slot, bit 1: a slot follows
    float threshold = 0.0f;        // push while new frontier / n < threshold
    float back_threshold = 0.0f;   // pull hands back to push when new frontier / n < back_threshold (0: never)
    unsigned long long heavy = ~0ull;
    // bottom-up pull (bfs_bottom_up in gl_spmspv.hip): a slot that does not scatter visits only the rows not reached
    // yet, row-wise with an early exit, when those rows hold fewer than bu_limit non-zeros (ctl[9] = that slot;
    // ctl[12..13] = non-zeros in the rows reached so far); nnz_rows = all non-zeros.  bu_limit 0: never.
    unsigned long long nnz_rows = 0, bu_limit = 0;
    __device__ bool pushes() const { return ctl[0] > slot; }
    __device__ bool row_wise() const { return ctl[8] == slot; }
    __device__ bool scatters() const { return ctl[0] > slot && ctl[8] != slot; }
    __device__ bool bottom_up() const { return !scatters() && ctl[9] == slot; }
    // ctl[14]: a slot reached nothing -- the frontier is empty, no later slot can change a distance: the steps only keep
    // the books from then on (the reference's loops run their remaining iterations on an empty vector)
    __device__ bool finished() const { return ctl[14] != 0u; }
    // ctl[15] = words of ctl: the words behind the 16 control words are two arrays of S = (ctl[15] - 16) / 2 entries, the
    // vertices slot s reached (ctl[16 + s]) and HOW the slot was evaluated (ctl[16 + S + s]: 1 scattered, 2 streamed
    // row-wise, 3 bottom-up; 0: nothing ran) -- the host's "edges actually traversed" 
    // `local`: ctl points at a workgroup's PRIVATE copy of the 16 control words (the one-launch shard step, gl_bfs_shard.h:
    // every workgroup replays the previous slot's decision for itself); the per-slot records then go to `records` -- the
    // caller's control words -- from the one workgroup that is given the pointer, and nowhere from the others
    uint32_t *records = nullptr;
    bool local = false;
    __device__ uint32_t *rec_() const { return local ? records : ctl; }
    __device__ uint32_t slot_capacity() const {
        const uint32_t *r = rec_();
        return r ? (r[15] - 16u) >> 1 : 0u;
    }
    __device__ void record_mode(uint32_t mode) const {
        const uint32_t S = slot_capacity();
        if (slot < S) rec_()[16u + S + slot] = mode;
    }
    // called once per slot, when the step that ran is complete, with the step's totals: vertices reached, non-zeros in
    // their columns (what a push from them scatters) and in their rows (what a pull no longer has to look at)
    __device__ void decide(uint32_t fresh, unsigned long long work, unsigned long long work_rows) const {
        if (fresh == 0u) ctl[14] = 1u;
        if (slot < slot_capacity()) rec_()[16u + slot] = fresh;   // the slot's new-frontier size, for the host
        unsigned long long *visited = reinterpret_cast<unsigned long long *>(ctl + 12);
        const unsigned long long vis = *visited + work_rows;
        *visited = vis;
        if (ctl[0] > slot) {
            const bool again = ctl[4] != 0xffffffffu;
            ctl[again ? 3 : 1] += 1u;
            const bool cont = again ? (may_continue & 2u) != 0u : (may_continue & 1u) != 0u;
            const float thr = again ? __uint_as_float(ctl[7]) : threshold;
            if (!(cont && ((float)fresh / (float)n < thr))) ctl[0] = slot + 1u;
        } else if (back_threshold > 0.0f && (may_continue & 2u) != 0u && (float)fresh / (float)n < back_threshold) {
            ctl[0] = 0xffffffffu;
            ctl[4] = slot;
            ctl[7] = __float_as_uint(back_threshold);
        }
        const bool next_scatters = ctl[0] > slot + 1u && work <= heavy;
        if (ctl[0] > slot + 1u && work > heavy) ctl[8] = slot + 1u;
        if (!next_scatters && nnz_rows - min(vis, nnz_rows) < bu_limit) ctl[9] = slot + 1u;
    }
};

// streamed-once 8-byte load (matrix streams): non-temporal so the stream does not
// evict the dense vector from L2
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 load_stream_nt(const uint2 *p) {
#ifdef GL_STREAM_PLAIN   // A/B builds only (scripts/build_variant.sh): ordinary loads, so that small matrices may stay in the caches
    return *p;
#else
    u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t *>(p));
    return make_uint2(v.x, v.y);
#endif
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream_nt16(const uint4 *p) {
#ifdef GL_STREAM_PLAIN
    return *p;
#else
    u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

// the same loads WITHOUT the non-temporal hint, for matrices small enough to stay in the 256 MB Infinity Cache from one run
// to the next (iterative algorithms stream the same matrix every iteration): same-box, the pokec stand-in's pattern
// layout (155 MB) 0.059 -> 0.046 ms, the googleplus general layout (88 MB) 51 -> 57 % of the HBM peak -- and orkut (1.7 GB)
// 70 -> 62 %, which is what the hint is for
__device__ __forceinline__ uint2 load_stream_keep(const uint2 *p) { return *p; }
__device__ __forceinline__ uint4 load_stream_keep16(const uint4 *p) { return *p; }

// Plan metadata is read-only for the whole launch: loading it through the constant address space lets the
// compiler keep wave-uniform loads on the scalar unit even though the span loop contains barriers
// (a fence makes ordinary global loads "clobbered", which turns them into vector loads + vmcnt(0) waits).
__device__ __forceinline__ uint32_t load_const(const uint32_t *p) {
    return *(const __attribute__((address_space(4))) uint32_t *)(p);
}
__device__ __forceinline__ uint4 load_const(const uint4 *p) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    return make_uint4(load_const(q), load_const(q + 1), load_const(q + 2), load_const(q + 3));
}

// the same test for a mask compared with 0 in the value type of opx (SpMV epilogue, dense assign)
template <int MASK, int OPX>
__device__ __forceinline__ bool mask_allows_zero(float m) {
    if (MASK == GL_MASK_WRITETOZERO) return value_is_zero<OPX>(m);
    if (MASK == GL_MASK_WRITETOONE) return !value_is_zero<OPX>(m);
    return true;
}

// mask test shared by SpMV epilogue and dense assign: "does the mask allow a write here?"
template <int MASK>
__device__ __forceinline__ bool mask_allows(float m, float ref) {
    if (MASK == GL_MASK_WRITETOZERO) return m == ref;
    if (MASK == GL_MASK_WRITETOONE) return m != ref;
    return true;
}

}  // namespace gl

#endif  // GL_COMMON_H_
