// graphlily/global.h -- MI355X build of GraphLily's global definitions.
//
// Drop-in for the reference's graphlily/global.h:57-164 as far as the SpMV / SpMSpV hot path and
// its callers use it: element types, the three semirings, mask types, aligned host vectors and
// convert_sparse_vec_to_dense_vec.  The FPGA plumbing of the reference header (HBM bank map,
// find_device, makefile strings, :27-54, :110-146) has no counterpart: there is no bitstream.
//
// val_t is float here (the reference's third option, global.h:64; the shipped default is
// ap_ufixed<32,8>, :63, which needs Xilinx ap_fixed.h).
#ifndef GRAPHLILY_GLOBAL_H_
#define GRAPHLILY_GLOBAL_H_

#include <algorithm>
#include <cassert>
#include <cmath>        // (the reference's global.h brings <cmath> and <fstream> in through ap_fixed.h / xcl2.hpp: its drivers call floor(),
#include <fstream>      //  abs(float) and use std::ofstream without including either -- bench_spmspv.cpp:157,308, bench_spmv.cpp)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "graphlily_hip.h"

// Host vectors handed to the device are page aligned in the reference (xcl2.hpp:61-76,
// aligned_allocator at global scope); kept so that caller code naming it compiles unchanged.
// Blocks come from the library's host pool (gl_host_pool_alloc): 4 KiB aligned like the reference's and recycled
// by size, so the n-element vectors the drivers build for every pull() / push() call (app/bfs.h:107-113) are not
// mapped and page-faulted in afresh each time.
namespace graphlily_detail {
inline int &noinit_depth() {
    static thread_local int depth = 0;
    return depth;
}
struct NoInitScope {
    NoInitScope() { ++noinit_depth(); }
    ~NoInitScope() { --noinit_depth(); }
    NoInitScope(const NoInitScope &) = delete;
    NoInitScope &operator=(const NoInitScope &) = delete;
};
}  // namespace graphlily_detail

template <typename T>
struct aligned_allocator {
    using value_type = T;
    aligned_allocator() = default;
    template <typename U>
    aligned_allocator(const aligned_allocator<U> &) {}
    T *allocate(std::size_t num) {
        void *ptr = nullptr;
        if (gl_host_pool_alloc(&ptr, (num == 0 ? 1 : num) * sizeof(T)) != GL_OK || !ptr) throw std::bad_alloc();
        return reinterpret_cast<T *>(ptr);
    }
    void deallocate(T *p, std::size_t) { gl_host_pool_free(p); }
    // vector<T, aligned_allocator<T>>(n) VALUE-initialises its elements, like the reference's allocator (xcl2.hpp:61-76 has
    // no construct(): drivers rely on zeroed vectors, tests/test_module_spmv_spmspv.cpp:210-211 reads vector[0].index before
    // writing it).  Only inside a graphlily_detail::NoInitScope -- the module layer's download helpers, whose vectors are
    // overwritten at once -- the elements are left as they are (no 12 MB fill in front of a 12 MB copy).
    template <typename U>
    void construct(U *p) {
        if (graphlily_detail::noinit_depth()) ::new (static_cast<void *>(p)) U;
        else ::new (static_cast<void *>(p)) U();
    }
    template <typename U, typename A0, typename... Args>
    void construct(U *p, A0 &&a0, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<Args>(args)...); }
    template <typename U>
    bool operator==(const aligned_allocator<U> &) const { return true; }
    template <typename U>
    bool operator!=(const aligned_allocator<U> &) const { return false; }
};

namespace graphlily {

inline std::string get_root_path_() {
    const char *p = getenv("GRAPHLILY_ROOT_PATH");
    return p ? std::string(p) : std::string();
}
const std::string root_path = get_root_path_();
const std::string device_name = "AMD Instinct MI355X (gfx950)";

// kept for callers that size things with them; they no longer describe hardware lanes
const uint32_t pack_size = 8;
const uint32_t spmv_row_interleave_factor = 1;
const uint32_t num_hbm_channels = 16;

// ---- the value type.  The reference picks val_t in global.h:62-64: `unsigned`, ap_ufixed<32, 8, AP_RND, AP_SAT> (the shipped
// default) or float.  Here float is the default; compiling with -DGRAPHLILY_VAL_UFIXED / -DGRAPHLILY_VAL_UNSIGNED selects the
// other two, so that a driver written for the reference's default configuration instantiates the same module templates
// (the kernels serve all three bit for bit, graphlily_hip.h GL_VAL_*).  Xilinx' ap_fixed.h is not needed: ufixed_32_8 below is
// a plain 32-bit word with the conversions and the arithmetic the host side of the reference's drivers uses.
struct ufixed_32_8 {
    uint32_t bits;   // value = bits / 2^24
    ufixed_32_8() = default;
    // AP_RND (round half up to 24 fraction bits), AP_SAT (clamp to [0, 2^32 - 1]); every arithmetic type converts through double
    ufixed_32_8(double v) {
        if (!(v > 0.0)) bits = 0u;
        else {
            const double q = v * 16777216.0 + 0.5;
            bits = q >= 4294967296.0 ? 0xffffffffu : (uint32_t)q;
        }
    }
    operator float() const { return (float)((double)bits / 16777216.0); }
    static ufixed_32_8 from_bits(uint32_t b) {
        ufixed_32_8 r;
        r.bits = b;
        return r;
    }
};
static_assert(sizeof(ufixed_32_8) == 4, "ufixed_32_8 is one 32-bit word");

#if defined(GRAPHLILY_VAL_UFIXED)
using val_t = ufixed_32_8;
#elif defined(GRAPHLILY_VAL_UNSIGNED)
using val_t = unsigned;
#else
using val_t = float;
#endif

// what the C ABI needs to know about a value type: GL_VAL_* and the 32-bit word of a value
template <typename T>
struct value_kind;
template <>
struct value_kind<float> {
    static const int kind = GL_VAL_FLOAT;
    static uint32_t bits(float v) {
        uint32_t b;
        memcpy(&b, &v, 4);
        return b;
    }
    static float from_float(float v) { return v; }
};
template <>
struct value_kind<unsigned> {
    static const int kind = GL_VAL_UNSIGNED;
    static uint32_t bits(unsigned v) { return v; }
    // csr_matrix_convert_from_float<unsigned> (io/data_loader.h:75-84): the C conversion, clamped instead of undefined
    static unsigned from_float(float v) { return v <= 0.0f ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (unsigned)v); }
};
template <>
struct value_kind<ufixed_32_8> {
    static const int kind = GL_VAL_UFIXED_32_8;
    static uint32_t bits(ufixed_32_8 v) { return v.bits; }
    static ufixed_32_8 from_float(float v) { return ufixed_32_8((double)v); }
};

typedef uint32_t idx_t;
const uint32_t idx_marker = 0xffffffff;
typedef struct {idx_t data[pack_size];} packed_idx_t;

typedef struct {idx_t index; val_t val;} idx_val_t;
typedef struct {idx_t index; float val;} idx_float_t;
static_assert(sizeof(idx_val_t) == sizeof(gl_idx_val), "idx_val_t must match the C ABI element (32-bit index, 32-bit value word)");

using aligned_dense_vec_t = std::vector<val_t, aligned_allocator<val_t>>;
using aligned_sparse_vec_t = std::vector<idx_val_t, aligned_allocator<idx_val_t>>;
using aligned_dense_float_vec_t = std::vector<float, aligned_allocator<float>>;
using aligned_sparse_float_vec_t = std::vector<idx_float_t, aligned_allocator<idx_float_t>>;

const val_t UINT_INF = val_t(4294967295.0);
const val_t UFIXED_INF = 255;
const val_t FLOAT_INF = 999999999;

enum OperationType {
    kMulAdd = 0,
    kLogicalAndOr = 1,
    kAddMin = 2,
};

struct SemiringType {
    OperationType op;
    val_t one;   // identity of <x>
    val_t zero;  // identity of <+>; a RUNTIME value on this backend (the FPGA hard-codes it per op)
};

const SemiringType ArithmeticSemiring = {kMulAdd, 1, 0};
const SemiringType LogicalSemiring = {kLogicalAndOr, 1, 0};
// float: the float line of the reference (global.h:100); the integer value types: its shipped line (:99, zero = UFIXED_INF)
#if defined(GRAPHLILY_VAL_UFIXED) || defined(GRAPHLILY_VAL_UNSIGNED)
const SemiringType TropicalSemiring = {kAddMin, 0, UFIXED_INF};
#else
const SemiringType TropicalSemiring = {kAddMin, 0, FLOAT_INF};
#endif

enum MaskType {
    kNoMask = 0,
    kMaskWriteToZero = 1,
    kMaskWriteToOne = 2,
};

const std::string proj_folder_name = "proj";

template <typename sparse_vec_t, typename dense_vec_t, typename value_t>
dense_vec_t convert_sparse_vec_to_dense_vec(const sparse_vec_t &sparse_vector, uint32_t range, value_t zero) {
    dense_vec_t dense_vector(range);
    typedef typename sparse_vec_t::value_type elem_t;
    typedef typename dense_vec_t::value_type out_t;
    // large float vectors (the push -> pull switch of app/bfs.h:196-201: a 3 M-element fill and a million scattered
    // stores on the orkut stand-in) go to the library's few-thread host loop; same result, entry for entry
    // (float values only: the fast path copies the 32-bit value words as they are)
    if (range >= (1u << 18) && sizeof(elem_t) == sizeof(gl_idx_val) && sizeof(out_t) == 4 && std::is_same<out_t, float>::value &&
        std::is_same<decltype(elem_t::val), float>::value &&
        !sparse_vector.empty() && (size_t)sparse_vector[0].index + 1 <= sparse_vector.size()) {
        float z = (float)zero;
        uint32_t zb;
        memcpy(&zb, &z, 4);
        if (gl_host_sparse_to_dense(reinterpret_cast<const gl_idx_val *>(sparse_vector.data()), range, zb, dense_vector.data()) == GL_OK)
            return dense_vector;
    }
    std::fill(dense_vector.begin(), dense_vector.end(), zero);
    const int nnz = sparse_vector[0].index;
    for (int i = 1; i < nnz + 1; i++) dense_vector[sparse_vector[i].index] = sparse_vector[i].val;
    return dense_vector;
}

// ---------------------------------------------------------------------------------------------
// Error convention of the reference (xcl2.hpp:40-46 OCL_CHECK): report and exit, never throw.
#define GRAPHLILY_CHECK(call)                                                                      \
    do {                                                                                           \
        int gl_rc_ = (call);                                                                       \
        if (gl_rc_ != GL_OK) {                                                                     \
            printf("%s:%d Error calling " #call ", error code is: %d (%s)\n", __FILE__, __LINE__,  \
                   gl_rc_, gl_last_error());                                                       \
            exit(EXIT_FAILURE);                                                                    \
        }                                                                                          \
    } while (0)

// A reference-counted device allocation: the role cl::Buffer plays in the reference's module API.
// Copyable, shareable between modules through bind_*_buf, freed with its last owner.
class DeviceBuffer {
    struct Impl {
        void *ptr = nullptr;
        size_t bytes = 0;
        // Set while the buffer's contents are owed by a deferred / fused module call (module/fusion.h): the first access
        // through ptr() settles the debt -- runs the deferred calls, or writes the float vector a fused BFS pull iteration
        // kept as bits -- so every reader sees what the unfused call sequence would have left.
        std::function<void()> on_access;
        // >= 0: the last writer was a fused BFS iteration that wrote this level (module/fusion.h) -- the buffer probably holds
        // levels only, and a download tries the packed read-back (gl_buf_d2h_levels checks every value).  Any other access
        // through ptr() forgets it.
        float levels_max = -1.0f;
        // number of the last access through ptr() (any of them may be a write; rptr() is the read-only access that leaves it):
        // SpMSpVModule::get_results_nnz trusts its run's completion record only while nobody has touched results_buf since
        uint64_t touched = 0;
        // Runs once before the next access that may WRITE the buffer (ptr(), not rptr() / download()): a buffer that owes "a copy of
        // this one, taken when somebody reads it" (module/fusion.h: results after a results -> vector swap) takes its copy now.
        std::function<void()> before_write;
        ~Impl() { if (ptr) gl_buf_free(ptr); }
    };
    std::shared_ptr<Impl> impl_;
    void settle_() const {
        if (impl_->on_access) {
            std::function<void()> f;
            f.swap(impl_->on_access);     // cleared first: the settling code uses the buffer itself
            f();
        }
    }

public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) : impl_(std::make_shared<Impl>()) {
        GRAPHLILY_CHECK(gl_buf_alloc(&impl_->ptr, bytes));
        impl_->bytes = bytes;
    }
    void *ptr() const {
        if (!impl_) return nullptr;
        settle_();
        if (impl_->before_write) {
            std::function<void()> f;
            f.swap(impl_->before_write);  // (once: whoever derived something from the old contents has it now)
            f();
        }
        impl_->levels_max = -1.0f;
        static uint64_t clock = 0;
        impl_->touched = ++clock;
        return impl_->ptr;
    }
    // the pointer for a call that only READS the buffer: its own debt is settled; the touch stamp, the levels hint and the
    // before-write hook stay
    const void *rptr() const {
        if (!impl_) return nullptr;
        settle_();
        return impl_->ptr;
    }
    uint64_t touched() const { return impl_ ? impl_->touched : 0; }
    size_t size() const { return impl_ ? impl_->bytes : 0; }
    bool valid() const { return impl_ && impl_->ptr != nullptr; }
    void upload(const void *host, size_t bytes) const {
        assert(bytes <= size());
        GRAPHLILY_CHECK(gl_buf_h2d(ptr(), host, bytes));
    }
    void download(void *host, size_t bytes) const {
        assert(bytes <= size());
        const float levels = impl_ ? impl_->levels_max : -1.0f;
        const void *p = rptr();                                      // (a download does not write)
        if (levels >= 0.0f && bytes == size() && (bytes & 3u) == 0) {
            GRAPHLILY_CHECK(gl_buf_d2h_levels(static_cast<float *>(host), static_cast<const float *>(p), bytes / 4u, levels, nullptr));
        } else {
            GRAPHLILY_CHECK(gl_buf_d2h(host, p, bytes));
        }
    }
    void mark_levels(float max_level) const { if (impl_) impl_->levels_max = max_level; }   // module/fusion.h
    // ---- for module/fusion.h
    const void *id() const { return impl_.get(); }                       // identity of the allocation behind the handle
    void *raw() const { return impl_ ? impl_->ptr : nullptr; }           // the pointer WITHOUT settling a debt
    bool owed() const { return impl_ && (bool)impl_->on_access; }
    void owe(std::function<void()> f) const { if (impl_) impl_->on_access = std::move(f); }
    void settle_quietly() const { if (impl_) impl_->on_access = nullptr; }
    void on_write(std::function<void()> f) const { if (impl_) impl_->before_write = std::move(f); }
    void clear_on_write() const { if (impl_) impl_->before_write = nullptr; }
    // exchange the device blocks of two buffers of equal size: every handle of `this` then names what `o`'s handles named and
    // vice versa (module/fusion.h: results -> vector "copies" of the pull loops become a swap)
    void swap_storage(const DeviceBuffer &o) const {
        assert(impl_ && o.impl_ && impl_->bytes == o.impl_->bytes);
        std::swap(impl_->ptr, o.impl_->ptr);
    }
    std::function<void()> take_debt() const {
        std::function<void()> f;
        if (impl_) f.swap(impl_->on_access);
        return f;
    }
};

}  // namespace graphlily

#include "graphlily/cl_buffers.h"   // cl::Buffer & co. as the reference's callers spell the module API's buffer handles

#endif  // GRAPHLILY_GLOBAL_H_
