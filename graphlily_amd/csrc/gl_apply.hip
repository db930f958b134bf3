// Element-wise "apply" operators of the overlay (modes 3-6, hw/overlay.cpp:380-411)
// and the sparse->dense conversion used at the push->pull switch.
// All are bandwidth-trivial, grid-stride, one launch each (the SSSP-mode sparse
// assign adds the ordered compaction of gl_compact.h for its new frontier).
#include "gl_common.h"
#include "gl_compact.h"

namespace gl {

static void *g_scratch = nullptr;
static size_t g_scratch_bytes = 0;

int scratch_reserve(size_t bytes, void **d_ptr) {
    if (bytes > g_scratch_bytes) {
        // previous users are ordered on the stream; drain before releasing
        GL_HIP(hipStreamSynchronize(ctx().stream));
        if (g_scratch) GL_HIP(hipFree(g_scratch));
        g_scratch = nullptr;
        g_scratch_bytes = 0;
        size_t want = bytes < (1u << 16) ? (1u << 16) : bytes;
        GL_HIP(hipMalloc(&g_scratch, want));
        // word 0 is the compaction's ticket (gl_compact.h); on the launch stream: the library's own stream does not
        // synchronise with the null stream a plain hipMemset runs on
        GL_HIP(hipMemsetAsync(g_scratch, 0, sizeof(uint32_t), ctx().stream));
        g_scratch_bytes = want;
    }
    *d_ptr = g_scratch;
    return GL_OK;
}

static inline unsigned stream_grid(uint64_t items_per_thread_total) {
    unsigned blocks = cdiv(items_per_thread_total, 256);
    unsigned cap = (unsigned)ctx().num_cus * 8u;
    if (blocks > cap) blocks = cap;
    return blocks ? blocks : 1u;
}

// hw/kernel_add_scalar_vector_dense_impl.h:6-27
__global__ __launch_bounds__(256) void ewise_add_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        uint32_t len, float val, bool vec4) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    if (vec4) {
        const uint32_t n4 = len >> 2;
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        float4 *out4 = reinterpret_cast<float4 *>(out);
        for (uint32_t i = tid; i < n4; i += stride) {
            float4 v = in4[i];
            v.x += val; v.y += val; v.z += val; v.w += val;
            out4[i] = v;
        }
        for (uint32_t i = (n4 << 2) + tid; i < len; i += stride) out[i] = in[i] + val;
    } else {
        for (uint32_t i = tid; i < len; i += stride) out[i] = in[i] + val;
    }
}

// hw/kernel_assign_vector_dense_impl.h:8-47
template <int MASK>
__global__ __launch_bounds__(256) void assign_dense_kernel(const float *__restrict__ mask, float *__restrict__ inout,
                                                           uint32_t len, float val) {
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < len; i += stride) {
        if (mask_allows<MASK>(mask[i], 0.0f)) inout[i] = val;
    }
}

// hw/kernel_assign_vector_sparse_no_new_frontier_impl.h:4-55
__global__ __launch_bounds__(256) void assign_sparse_kernel(const gl_idx_val *__restrict__ mask, float *__restrict__ inout,
                                                            float val) {
    const uint32_t n = mask[0].index;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) inout[mask[1u + i].index] = val;
}

// hw/kernel_assign_vector_sparse_new_frontier_impl.h:4-78: candidates are the mask
// entries; kept (and relaxed) when inout[idx] > v.  Mask indices are unique (they
// come from a SpMSpV result), so entries do not interact.
struct RelaxSource {
    const gl_idx_val *mask;
    float *inout;
    __device__ uint32_t size() const { return mask[0].index; }
    __device__ bool get(uint32_t i, gl_idx_val &out) const {
        out = mask[1u + i];
        return inout[out.index] > out.val;
    }
    __device__ void consumed(uint32_t i) const {
        gl_idx_val m = mask[1u + i];
        if (inout[m.index] > m.val) inout[m.index] = m.val;
    }
    __device__ void emitted(const gl_idx_val &) const {}
    __device__ void begin_chunk(uint32_t) const {}
};

// ---- the apply kernels for the integer value types (gl_common.h): SAT = ap_ufixed<32,8,AP_RND,AP_SAT>, else unsigned
template <bool SAT>
__global__ __launch_bounds__(256) void ewise_add_bits_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t len, uint32_t val) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < len; i += gridDim.x * 256u) out[i] = SAT ? sat_add_u32(in[i], val) : in[i] + val;
}

template <int MASK>
__global__ __launch_bounds__(256) void assign_dense_bits_kernel(const uint32_t *__restrict__ mask, uint32_t *__restrict__ inout, uint32_t len,
                                                                uint32_t val) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < len; i += gridDim.x * 256u) {
        const bool z = mask[i] == 0u;
        if (MASK == GL_MASK_WRITETOZERO ? z : !z) inout[i] = val;
    }
}

struct RelaxSourceBits {   // RelaxSource with the unsigned ordering of the bits (both integer types order like that)
    const gl_idx_val *mask;
    uint32_t *inout;
    __device__ uint32_t size() const { return mask[0].index; }
    __device__ bool get(uint32_t i, gl_idx_val &out) const {
        out = mask[1u + i];
        return inout[out.index] > __float_as_uint(out.val);
    }
    __device__ void consumed(uint32_t i) const {
        const gl_idx_val m = mask[1u + i];
        if (inout[m.index] > __float_as_uint(m.val)) inout[m.index] = __float_as_uint(m.val);
    }
    __device__ void emitted(const gl_idx_val &) const {}
    __device__ void begin_chunk(uint32_t) const {}
};

// graphlily/global.h:153-164
__global__ __launch_bounds__(256) void sparse_scatter_kernel(const gl_idx_val *__restrict__ sv, float *__restrict__ dense,
                                                             uint32_t range) {
    const uint32_t n = sv[0].index;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
        gl_idx_val e = sv[1u + i];
        if (e.index < range) dense[e.index] = e.val;
    }
}

// gl_bfs_bits_begin: distances, the bit vectors (vector 1 = {source}: slot s reads vector s and writes vector s + 1, slot 1
// is the first) and the control words (BfsBitsCtl, gl_common.h)
__global__ __launch_bounds__(256) void bfs_bits_begin_kernel(uint32_t *__restrict__ ctl, uint32_t ctl_words, float *__restrict__ distance,
                                                             uint32_t n, uint32_t *__restrict__ bits, uint32_t words, uint32_t nvec,
                                                             uint32_t first_pull_slot) {
    const uint32_t src = ctl[2];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    for (uint32_t i = tid; i < n; i += stride) distance[i] = (i == src) ? 1.0f : 0.0f;      // app/bfs.h:168-171
    for (uint32_t w = tid; w < nvec * words; w += stride) bits[w] = (w == words + (src >> 5)) ? (1u << (src & 31u)) : 0u;
    for (uint32_t w = tid; w < ctl_words; w += stride)     // (grid-stride: more slots than a small graph's launch has threads)
        if (w != 2u) ctl[w] = w == 0u ? first_pull_slot : (w == 4u ? 0xffffffffu : (w == 15u ? ctl_words : 0u));
}

// the set bits of a frontier bit vector as list candidates (gl_bfs_pull_step_back): entry {row, 1}
struct BitsSource {
    const uint32_t *bits;
    uint32_t n;
    __device__ uint32_t size() const { return n; }
    __device__ bool get(uint32_t i, gl_idx_val &out) const {
        out.index = i;
        out.val = 1.0f;
        return (bits[i >> 5] >> (i & 31u)) & 1u;
    }
    __device__ void consumed(uint32_t) const {}
    __device__ void emitted(const gl_idx_val &) const {}
    __device__ void begin_chunk(uint32_t) const {}
};

int bits_to_sparse_gated(const uint32_t *d_bits, uint32_t n, gl_idx_val *d_out, uint32_t *d_counts, const uint32_t *gate_word,
                         uint32_t gate_value, hipStream_t s) {
    BitsSource src{d_bits, n};
    Gate gate;
    gate.word = gate_word;
    gate.value = gate_value;
    gate.op = GL_GATE_EQ;
    return run_compaction(src, n, d_counts, d_out, 0.0f, s, nullptr, gate);
}

}  // namespace gl

extern "C" {


int gl_bfs_bits_begin(uint32_t *d_ctl, uint32_t ctl_words, float *d_distance, uint32_t n, uint32_t *d_bits, uint32_t bits_words,
                      uint32_t nvec, uint32_t first_pull_slot) {
    GL_REQUIRE_INIT();
    GL_ARG(d_ctl != nullptr && d_distance != nullptr && d_bits != nullptr && n > 0 && (uint64_t)bits_words * 32u >= n);
    GL_ARG(((uintptr_t)d_ctl & 7u) == 0 && ctl_words >= 18u && ctl_words <= 65536u && nvec >= 3u);
    gl::bfs_bits_begin_kernel<<<gl::stream_grid(n), 256, 0, gl::ctx().stream>>>(d_ctl, ctl_words, d_distance, n, d_bits, bits_words, nvec,
                                                                                  first_pull_slot);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_ewise_add(const float *d_in, float *d_out, uint32_t len, float val) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    if (len == 0) return GL_OK;
    GL_ARG(d_in != nullptr && d_out != nullptr);
    const bool vec4 = ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u) == 0;
    gl::ewise_add_kernel<<<gl::stream_grid(vec4 ? (len + 3) / 4 : len), 256, 0, gl::ctx().stream>>>(d_in, d_out, len, val, vec4);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_assign_dense(const float *d_mask, float *d_inout, uint32_t len, float val, int mask_type) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    if (mask_type != GL_MASK_WRITETOZERO && mask_type != GL_MASK_WRITETOONE)
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_assign_dense: Invalid mask type %d", mask_type);
    if (len == 0) return GL_OK;
    GL_ARG(d_mask != nullptr && d_inout != nullptr);
    hipStream_t s = gl::ctx().stream;
    if (mask_type == GL_MASK_WRITETOZERO)
        gl::assign_dense_kernel<GL_MASK_WRITETOZERO><<<gl::stream_grid(len), 256, 0, s>>>(d_mask, d_inout, len, val);
    else
        gl::assign_dense_kernel<GL_MASK_WRITETOONE><<<gl::stream_grid(len), 256, 0, s>>>(d_mask, d_inout, len, val);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_assign_sparse(const gl_idx_val *d_mask, float *d_inout, float val, uint32_t max_entries) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_ARG(d_mask != nullptr && d_inout != nullptr);
    if (max_entries == 0) return GL_OK;
    gl::assign_sparse_kernel<<<gl::stream_grid(max_entries), 256, 0, gl::ctx().stream>>>(d_mask, d_inout, val);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_assign_sparse_new_frontier(const gl_idx_val *d_mask, float *d_inout, gl_idx_val *d_new_frontier,
                                  uint32_t max_entries) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_ARG(d_mask != nullptr && d_inout != nullptr && d_new_frontier != nullptr);
    GL_ARG((const void *)d_mask != (const void *)d_new_frontier);
    void *counts = nullptr;
    int rc = gl::scratch_reserve((size_t)(gl::cdiv(max_entries, gl::kCompactChunk) + 1) * sizeof(uint32_t), &counts);
    if (rc != GL_OK) return rc;
    gl::RelaxSource src{d_mask, d_inout};
    // head of the new frontier is {count, 0} (kernel_assign_vector_sparse_new_frontier_impl.h:73-77)
    return gl::run_compaction(src, max_entries, (uint32_t *)counts, d_new_frontier, 0.0f, gl::ctx().stream);
}

int gl_sparse_to_dense(const gl_idx_val *d_sparse, float *d_dense, uint32_t range, float zero, uint32_t max_entries) {
    GL_TRACE();
    GL_REQUIRE_INIT();
    GL_ARG(d_sparse != nullptr && d_dense != nullptr);
    int rc = gl_buf_fill_f32(d_dense, zero, range);
    if (rc != GL_OK) return rc;
    if (max_entries == 0) return GL_OK;
    gl::sparse_scatter_kernel<<<gl::stream_grid(max_entries), 256, 0, gl::ctx().stream>>>(d_sparse, d_dense, range);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

// ---- the same operators over the reference's other value types (bit patterns; see gl_common.h)
static int check_val_type(int val_type, const char *who) {
    if (val_type == GL_VAL_FLOAT || val_type == GL_VAL_UNSIGNED || val_type == GL_VAL_UFIXED_32_8) return GL_OK;
    return gl::set_error(GL_ERR_INVALID_ARG, "%s: unknown value type %d", who, val_type);
}

int gl_ewise_add_typed(const void *d_in, void *d_out, uint32_t len, uint32_t val_bits, int val_type) {
    GL_REQUIRE_INIT();
    int rc = check_val_type(val_type, "gl_ewise_add_typed");
    if (rc != GL_OK) return rc;
    if (val_type == GL_VAL_FLOAT) return gl_ewise_add((const float *)d_in, (float *)d_out, len, __builtin_bit_cast(float, val_bits));
    if (len == 0) return GL_OK;
    GL_ARG(d_in != nullptr && d_out != nullptr);
    if (val_type == GL_VAL_UFIXED_32_8)
        gl::ewise_add_bits_kernel<true><<<gl::stream_grid(len), 256, 0, gl::ctx().stream>>>((const uint32_t *)d_in, (uint32_t *)d_out, len, val_bits);
    else
        gl::ewise_add_bits_kernel<false><<<gl::stream_grid(len), 256, 0, gl::ctx().stream>>>((const uint32_t *)d_in, (uint32_t *)d_out, len, val_bits);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

int gl_assign_dense_typed(const void *d_mask, void *d_inout, uint32_t len, uint32_t val_bits, int mask_type, int val_type) {
    GL_REQUIRE_INIT();
    int rc = check_val_type(val_type, "gl_assign_dense_typed");
    if (rc != GL_OK) return rc;
    if (val_type == GL_VAL_FLOAT)
        return gl_assign_dense((const float *)d_mask, (float *)d_inout, len, __builtin_bit_cast(float, val_bits), mask_type);
    if (mask_type != GL_MASK_WRITETOZERO && mask_type != GL_MASK_WRITETOONE)
        return gl::set_error(GL_ERR_INVALID_ARG, "gl_assign_dense: Invalid mask type %d", mask_type);
    if (len == 0) return GL_OK;
    GL_ARG(d_mask != nullptr && d_inout != nullptr);
    hipStream_t s = gl::ctx().stream;
    if (mask_type == GL_MASK_WRITETOZERO)
        gl::assign_dense_bits_kernel<GL_MASK_WRITETOZERO><<<gl::stream_grid(len), 256, 0, s>>>((const uint32_t *)d_mask, (uint32_t *)d_inout, len, val_bits);
    else
        gl::assign_dense_bits_kernel<GL_MASK_WRITETOONE><<<gl::stream_grid(len), 256, 0, s>>>((const uint32_t *)d_mask, (uint32_t *)d_inout, len, val_bits);
    GL_LAUNCH_CHECK();
    return GL_OK;
}

/* value moves only: the float entry point with the bits passed through */
int gl_assign_sparse_typed(const void *d_mask, void *d_inout, uint32_t val_bits, uint32_t max_entries) {
    return gl_assign_sparse((const gl_idx_val *)d_mask, (float *)d_inout, __builtin_bit_cast(float, val_bits), max_entries);
}

int gl_sparse_to_dense_typed(const void *d_sparse, void *d_dense, uint32_t range, uint32_t zero_bits, uint32_t max_entries) {
    return gl_sparse_to_dense((const gl_idx_val *)d_sparse, (float *)d_dense, range, __builtin_bit_cast(float, zero_bits), max_entries);
}

int gl_assign_sparse_new_frontier_typed(const void *d_mask, void *d_inout, void *d_new_frontier, uint32_t max_entries, int val_type) {
    GL_REQUIRE_INIT();
    int rc = check_val_type(val_type, "gl_assign_sparse_new_frontier_typed");
    if (rc != GL_OK) return rc;
    if (val_type == GL_VAL_FLOAT)
        return gl_assign_sparse_new_frontier((const gl_idx_val *)d_mask, (float *)d_inout, (gl_idx_val *)d_new_frontier, max_entries);
    GL_ARG(d_mask != nullptr && d_inout != nullptr && d_new_frontier != nullptr && d_mask != d_new_frontier);
    void *counts = nullptr;
    rc = gl::scratch_reserve((size_t)(gl::cdiv(max_entries, gl::kCompactChunk) + 1) * sizeof(uint32_t), &counts);
    if (rc != GL_OK) return rc;
    gl::RelaxSourceBits src{(const gl_idx_val *)d_mask, (uint32_t *)d_inout};
    return gl::run_compaction(src, max_entries, (uint32_t *)counts, (gl_idx_val *)d_new_frontier, 0.0f, gl::ctx().stream);
}

}  // extern "C"

// gl_init loads this translation unit's code object up front (HIP defers that to the unit's first launch, which would put
// tens of milliseconds into somebody's timed call)
namespace gl {
int preload_apply() {
    hipFuncAttributes attr;
    GL_HIP(hipFuncGetAttributes(&attr, (const void *)sparse_scatter_kernel));
    return GL_OK;
}
}  // namespace gl
