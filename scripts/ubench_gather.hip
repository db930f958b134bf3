// Scratch microbenchmark (not part of the product): what limits a gather-heavy stream on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_gather scripts/ubench_gather.hip && /tmp/ubench_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: stream only (8 B/lane); 1: stream + global gather; 2: stream(16 B/lane) only;
// 3: stream + LDS gather (x tile staged in LDS, index masked into the tile)
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_stream(const uint2 *__restrict__ s, const float *__restrict__ x,
                                                float *__restrict__ out, size_t n, uint32_t xmask, uint32_t lds_elems) {
    extern __shared__ float xt[];
    if (MODE == 3) {
        for (uint32_t i = threadIdx.x; i < lds_elems; i += 256) xt[i] = x[i];
        __syncthreads();
    }
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t base = tid; base < n; base += stride * U) {
        u32x2 c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t k = base + u * stride;
            c[u] = (k < n) ? __builtin_nontemporal_load((const u32x2 *)(s + k)) : (u32x2){0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float xv = 1.0f;
            if (MODE == 1) xv = x[c[u].x & xmask];
            if (MODE == 3) xv = xt[c[u].x & (lds_elems - 1)];
            acc += __uint_as_float(c[u].y) * xv;
        }
    }
    if (acc == 123.456f) out[tid] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_stream16(const uint4 *__restrict__ s, float *__restrict__ out, size_t n16) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t base = tid; base < n16; base += stride * U) {
        u32x4 c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t k = base + u * stride;
            c[u] = (k < n16) ? __builtin_nontemporal_load((const u32x4 *)(s + k)) : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += __uint_as_float(c[u].y) + __uint_as_float(c[u].w);
    }
    if (acc == 123.456f) out[tid] = acc;
}

// gather only: indices from a hash, no stream
template <int U>
__global__ __launch_bounds__(256) void k_gather_only(const float *__restrict__ x, float *__restrict__ out, size_t n,
                                                     uint32_t xmask) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t base = tid; base < n; base += stride * U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t h = (uint32_t)(base + u * stride) * 2654435761u;
            h ^= h >> 15;
            h *= 2246822519u;
            h ^= h >> 13;
            acc += x[h & xmask];
        }
    }
    if (acc == 123.456f) out[tid] = acc;
}

template <typename F>
static double time_ms(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const size_t n = 128ull << 20;  // 128 Mi entries = 1 GiB stream
    std::vector<uint2> h(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; i++) h[i] = make_uint2(rng(), 0x3f800000u);
    uint2 *s;
    float *x, *out;
    const size_t xn = 4u << 20;  // 4 Mi floats = 16 MiB
    CK(hipMalloc(&s, n * 8));
    CK(hipMalloc(&x, xn * 4));
    CK(hipMalloc(&out, 1 << 24));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, xn * 4));
    const int grid = 256 * 8;
    const double gb = n * 8 / 1e9;
    double t;
    t = time_ms([&] { k_stream<0, 4><<<grid, 256>>>(s, x, out, n, 0, 0); }, 10);
    printf("stream only 8B/lane U4       : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { k_stream<0, 8><<<grid, 256>>>(s, x, out, n, 0, 0); }, 10);
    printf("stream only 8B/lane U8       : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { k_stream16<4><<<grid, 256>>>((const uint4 *)s, out, n / 2); }, 10);
    printf("stream only 16B/lane U4      : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = time_ms([&] { k_stream16<4><<<grid * 4, 256>>>((const uint4 *)s, out, n / 2); }, 10);
    printf("stream only 16B/lane U4 g8192: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    for (uint32_t bits : {10u, 13u, 16u, 18u, 20u, 21u, 22u}) {
        uint32_t mask = (1u << bits) - 1;
        t = time_ms([&] { k_stream<1, 4><<<grid, 256>>>(s, x, out, n, mask, 0); }, 5);
        printf("stream + gather x[%4u KiB] U4 : %.3f ms  %.0f GB/s  %.1f Ggather/s\n", (4u << bits) >> 10, t,
               gb / t * 1e3, n / t / 1e6);
    }
    for (uint32_t bits : {13u, 20u, 22u}) {
        uint32_t mask = (1u << bits) - 1;
        t = time_ms([&] { k_stream<1, 8><<<grid, 256>>>(s, x, out, n, mask, 0); }, 5);
        printf("stream + gather x[%4u KiB] U8 : %.3f ms  %.0f GB/s  %.1f Ggather/s\n", (4u << bits) >> 10, t,
               gb / t * 1e3, n / t / 1e6);
    }
    for (uint32_t bits : {13u, 20u, 22u}) {
        uint32_t mask = (1u << bits) - 1;
        t = time_ms([&] { k_gather_only<8><<<grid, 256>>>(x, out, n, mask); }, 5);
        printf("gather only   x[%4u KiB] U8   : %.3f ms  %.1f Ggather/s\n", (4u << bits) >> 10, t, n / t / 1e6);
    }
    for (uint32_t le : {8192u, 16384u}) {
        CK(hipFuncSetAttribute((const void *)k_stream<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, le * 4));
        t = time_ms([&] { k_stream<3, 4><<<grid, 256, le * 4>>>(s, x, out, n, 0, le); }, 5);
        printf("stream + LDS gather [%u KiB] U4: %.3f ms  %.0f GB/s  %.1f Ggather/s\n", le * 4 >> 10, t, gb / t * 1e3,
               n / t / 1e6);
    }
    return 0;
}
