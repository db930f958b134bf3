// LDS accumulator policies shared by the SpMV unit kernels (gl_spmv.hip) and the SpMSpV fold kernel (gl_spmspv.hip):
// one per semiring x value type, on top of the Semiring<OP> ALUs of gl_common.h.
#ifndef GL_TILE_H_
#define GL_TILE_H_

#include "gl_common.h"

namespace gl {

template <int OP>
struct Tile;  // LDS accumulator policy

template <>
struct Tile<GL_OP_MULADD> {
    // f64 on purpose: ds_add_f64 runs at full rate next to the stream, ds_add_f32 does not (a build with float accumulators
    // and the 2.7 x larger hot table they leave room for ran the orkut stand-in in 1.14 ms instead of 0.315), and the sum of
    // a row is correctly rounded whatever order the wavefronts add in
    using T = double;
    __device__ static T ident() { return 0.0; }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) {
        // float product as in the reference (spmv_module.h:495), f64 accumulation
        __hip_atomic_fetch_add(&t[r], (T)(a * xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T lift(float z) { return (T)z; }
    __device__ static void accz(T *t, uint32_t r, float z) {   // z = a (x) x already formed (pattern plans)
        __hip_atomic_fetch_add(&t[r], (T)z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T comb(T x, T y) { return x + y; }
    // a partial result formed in registers (the row-packed hot stream: comb() of a record's lifted table values) into row r
    __device__ static void accl(T *t, uint32_t r, T v) { __hip_atomic_fetch_add(&t[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ static float zident() { return 0.0f; }   // the z whose lift() is comb()'s identity (padding table slot)
    __device__ static float get(const T *t, uint32_t r) { return (float)t[r]; }
    __device__ static float init(float zero) { return zero; }
    __device__ static float finish(float zero, float s) { return zero + s; }
};

template <>
struct Tile<GL_OP_ANDOR> {
    using T = float;
    __device__ static T ident() { return 0.0f; }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) {
        if (a != 0.0f && xv != 0.0f) t[r] = 1.0f;   // every writer stores the same value
    }
    __device__ static T lift(float z) { return z != 0.0f ? 1.0f : 0.0f; }
    __device__ static void accz(T *t, uint32_t r, float z) {
        if (z != 0.0f) t[r] = 1.0f;
    }
    __device__ static T comb(T x, T y) { return (x != 0.0f || y != 0.0f) ? 1.0f : 0.0f; }
    __device__ static void accl(T *t, uint32_t r, T v) {
        if (v != 0.0f) t[r] = 1.0f;
    }
    __device__ static float zident() { return 0.0f; }
    __device__ static float get(const T *t, uint32_t r) { return t[r]; }
    __device__ static float init(float zero) { return zero != 0.0f ? 1.0f : 0.0f; }
    __device__ static float finish(float zero, float s) { return (zero != 0.0f || s != 0.0f) ? 1.0f : 0.0f; }
};

// ordered-integer min: floats >= 0 compare like int, floats < 0 like reversed uint
__device__ __forceinline__ void atomic_min_f32_as_int(float *addr, float v) {
    if (!(__float_as_uint(v) >> 31))   // by sign bit: -0.0 must take the negative path (v >= 0 is true for it)
        atomicMin((int *)addr, __float_as_int(v));
    else
        atomicMax((unsigned int *)addr, __float_as_uint(v));
}

template <>
struct Tile<GL_OP_ADDMIN> {
    using T = float;
    __device__ static T ident() { return __builtin_inff(); }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) { atomic_min_f32_as_int(&t[r], a + xv); }
    __device__ static T lift(float z) { return z; }
    __device__ static void accz(T *t, uint32_t r, float z) { atomic_min_f32_as_int(&t[r], z); }
    __device__ static T comb(T x, T y) { return (y < x) ? y : x; }
    __device__ static void accl(T *t, uint32_t r, T v) { atomic_min_f32_as_int(&t[r], v); }
    __device__ static float zident() { return __builtin_inff(); }
    __device__ static float get(const T *t, uint32_t r) { return t[r]; }
    __device__ static float init(float zero) { return zero; }
    __device__ static float finish(float zero, float s) { return (s < zero) ? s : zero; }
};

// ---- the integer value types (gl_common.h): 32-bit LDS accumulators holding the bits
template <>
struct Tile<kOpU32MulAdd> {
    using T = uint32_t;
    __device__ static T ident() { return 0u; }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) {
        __hip_atomic_fetch_add(&t[r], fbits(a) * fbits(xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T lift(float z) { return fbits(z); }
    __device__ static void accz(T *t, uint32_t r, float z) {
        __hip_atomic_fetch_add(&t[r], fbits(z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T comb(T x, T y) { return x + y; }
    __device__ static void accl(T *t, uint32_t r, T v) { __hip_atomic_fetch_add(&t[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ static float zident() { return bitsf(0u); }
    __device__ static float get(const T *t, uint32_t r) { return bitsf(t[r]); }
    __device__ static float init(float zero) { return zero; }
    __device__ static float finish(float zero, float s) { return bitsf(fbits(zero) + fbits(s)); }
};

// (+,x) over ap_ufixed<32,8,AP_RND,AP_SAT>: the rounded, saturated products (non-negative) are added EXACTLY in 64 bits
// (ds_add_u64; a row of 2^32 products of < 2^32 each still fits) and clamped once when the row is read -- the same word as the
// reference's clamped running sum in any order (gl_common.h)
template <>
struct Tile<kOpFixMulAdd> {
    using T = unsigned long long;
    __device__ static T ident() { return 0ull; }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) {
        __hip_atomic_fetch_add(&t[r], (T)fix_mul_u32(fbits(a), fbits(xv)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T lift(float z) { return (T)fbits(z); }
    __device__ static void accz(T *t, uint32_t r, float z) {   // z = colval (x) x, already rounded (pattern plans)
        __hip_atomic_fetch_add(&t[r], (T)fbits(z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T comb(T x, T y) { return x + y; }
    __device__ static void accl(T *t, uint32_t r, T v) { __hip_atomic_fetch_add(&t[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ static float zident() { return bitsf(0u); }
    __device__ static float get(const T *t, uint32_t r) { return bitsf(t[r] > 0xffffffffull ? 0xffffffffu : (uint32_t)t[r]); }
    __device__ static float init(float zero) { return zero; }
    __device__ static float finish(float zero, float s) { return bitsf(sat_add_u32(fbits(zero), fbits(s))); }
};

template <uint32_t ONE>
struct TileBitsAndOr {
    using T = uint32_t;
    __device__ static T ident() { return 0u; }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) {
        if (fbits(a) != 0u && fbits(xv) != 0u) t[r] = ONE;   // every writer stores the same value
    }
    __device__ static T lift(float z) { return fbits(z) != 0u ? ONE : 0u; }
    __device__ static void accz(T *t, uint32_t r, float z) {
        if (fbits(z) != 0u) t[r] = ONE;
    }
    __device__ static T comb(T x, T y) { return (x | y) ? ONE : 0u; }
    __device__ static void accl(T *t, uint32_t r, T v) {
        if (v) t[r] = ONE;
    }
    __device__ static float zident() { return bitsf(0u); }
    __device__ static float get(const T *t, uint32_t r) { return bitsf(t[r]); }
    __device__ static float init(float zero) { return bitsf(fbits(zero) != 0u ? ONE : 0u); }
    __device__ static float finish(float zero, float s) { return bitsf((fbits(zero) | fbits(s)) ? ONE : 0u); }
};
template <>
struct Tile<kOpU32AndOr> : TileBitsAndOr<1u> {};
template <>
struct Tile<kOpFixAndOr> : TileBitsAndOr<kFixOne> {};

template <int OPX>
struct TileBitsAddMin {
    using T = uint32_t;
    __device__ static T ident() { return 0xffffffffu; }
    __device__ static void acc(T *t, uint32_t r, float a, float xv) {
        __hip_atomic_fetch_min(&t[r], fbits(Semiring<OPX>::mul(a, xv)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T lift(float z) { return fbits(z); }
    __device__ static void accz(T *t, uint32_t r, float z) {
        __hip_atomic_fetch_min(&t[r], fbits(z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ static T comb(T x, T y) { return min(x, y); }
    __device__ static void accl(T *t, uint32_t r, T v) { __hip_atomic_fetch_min(&t[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ static float zident() { return bitsf(0xffffffffu); }
    __device__ static float get(const T *t, uint32_t r) { return bitsf(t[r]); }
    __device__ static float init(float zero) { return zero; }
    __device__ static float finish(float zero, float s) { return bitsf(min(fbits(zero), fbits(s))); }
};
template <>
struct Tile<kOpU32AddMin> : TileBitsAddMin<kOpU32AddMin> {};
template <>
struct Tile<kOpFixAddMin> : TileBitsAddMin<kOpFixAddMin> {};

}  // namespace gl

#endif  // GL_TILE_H_
