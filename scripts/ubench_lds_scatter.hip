// Scratch microbenchmark 2 (not part of the product): inner loop of a "row-block in LDS, column-sorted
// stream" SpMV: per entry one semi-coalesced global gather of x and one LDS atomic into a 32K-row tile.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_lds scripts/ubench_lds_scatter.hip
#include <hip/hip_runtime.h>

#include <algorithm>
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

constexpr int kRows = 32768;  // y tile rows (128 KiB of LDS)

// entry: x = col_off (17 bits) << 15 | row_local (15 bits), y = val; bases[g] = base column of 64-entry group g
// ACC: 0 none, 1 ds_add_f32, 2 plain store, 3 int min;  GATHER: 0 none, 1 global x[base+off]
template <int ACC, int GATHER, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void k_block(const uint2 *__restrict__ s, const uint32_t *__restrict__ bases,
                                                   const float *__restrict__ x, float *__restrict__ y,
                                                   size_t per_block) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    unsigned long long *t64 = reinterpret_cast<unsigned long long *>(__builtin_assume_aligned(tile, 16));
    double *td = reinterpret_cast<double *>(__builtin_assume_aligned(tile, 16));
    for (int i = threadIdx.x; i < kRows; i += THREADS) tile[i] = 0.f;
    __syncthreads();
    const size_t begin = (size_t)blockIdx.x * per_block;
    for (size_t base = begin + threadIdx.x; base < begin + per_block; base += (size_t)THREADS * U) {
        u32x2 c[U];
        uint32_t b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t k = base + (size_t)u * THREADS;
            c[u] = __builtin_nontemporal_load((const u32x2 *)(s + k));
            b[u] = GATHER ? bases[k >> 6] : 0u;
        }
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) xv[u] = GATHER ? x[b[u] + (c[u].x >> 15)] : 1.0f;
#pragma unroll
        for (int u = 0; u < U; u++) {
            float p = __uint_as_float(c[u].y) * xv[u];
            uint32_t r = c[u].x & 0x7fffu;
            if (ACC == 1) __hip_atomic_fetch_add(&tile[r], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ACC == 2) { if (p != 0.f) tile[r] = 1.0f; }
            if (ACC == 3) atomicMin((int *)&tile[r], __float_as_int(p));
            if (ACC == 0) { if (p == 123.456f) tile[r] = p; }
            if (ACC == 4) atomicAdd((unsigned int *)&tile[r], __float_as_uint(p));
            if (ACC == 5) atomicAdd(&t64[r & 0x3fffu], (unsigned long long)(long long)(p * 1048576.0f));
            if (ACC == 6) { uint32_t rr = (uint32_t)((threadIdx.x + u * 7) & 0x7fffu); __hip_atomic_fetch_add(&tile[rr], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (ACC == 7) { float old = tile[r]; tile[r] = old + p; }
            if (ACC == 8) __hip_atomic_fetch_add(&td[r & 0x3fffu], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ACC == 9) { float o = __hip_atomic_fetch_add(&tile[r], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (o == 123.456f) tile[0] = o; }
        }
    }
    __syncthreads();
    // flush: rows with a non-zero partial go to global y with one atomic each
    for (int i = threadIdx.x; i < kRows; i += THREADS) {
        float v = tile[i];
        if (v != 0.f) unsafeAtomicAdd(&y[(blockIdx.x % 75) * kRows + i], v);
    }
}

template <typename F>
static double time_ms(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
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

int main(int argc, char **argv) {
    const int nblocks = 512;
    const size_t per_block = 256 * 1024;              // entries per workgroup
    const size_t n = (size_t)nblocks * per_block;     // 128 Mi entries = 1 GiB
    const size_t xn = 2500000;
    std::vector<uint2> h(n);
    std::vector<uint32_t> hb(n / 64);
    std::mt19937 rng(1);
    for (int hub = 0; hub < 2; hub++) {
        // per block: sorted columns with mean gap 1.5 over the x range, random rows (optionally 10% on one hub row)
        for (int b = 0; b < nblocks; b++) {
            double col = 0, gap = (double)(xn - 200000) / 4.0 / per_block;  // each block sweeps 1/4 of x
            double start = (b % 4) * (xn / 4.0);
            for (size_t i = 0; i < per_block; i += 64) {
                uint32_t base = (uint32_t)(start + col);
                hb[(b * per_block + i) / 64] = base;
                for (int j = 0; j < 64; j++) {
                    col += gap * (0.5 + (rng() & 1023) / 1024.0);
                    uint32_t off = (uint32_t)(start + col) - base;
                    uint32_t row = (hub && (rng() % 10 == 0)) ? 777u : (rng() & 0x7fffu);
                    h[b * per_block + i + j] = make_uint2((off << 15) | row, 0x3f800000u);
                }
            }
        }
        uint2 *s;
        uint32_t *bases;
        float *x, *y;
        CK(hipMalloc(&s, n * 8));
        CK(hipMalloc(&bases, hb.size() * 4));
        CK(hipMalloc(&x, (xn + 1000000) * 4));
        CK(hipMalloc(&y, 75 * kRows * 4));
        CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(x, 0, (xn + 1000000) * 4));
        CK(hipMemset(y, 0, 75 * kRows * 4));
        const double gb = n * 8 / 1e9;
        const size_t lds = kRows * 4;
        printf("--- rows %s\n", hub ? "10% on one hub row" : "uniform");
#define RUN(ACC, G, T, U, name)                                                                           \
    {                                                                                                     \
        CK(hipFuncSetAttribute((const void *)k_block<ACC, G, T, U>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)lds));                                                                \
        double t = time_ms([&] { k_block<ACC, G, T, U><<<nblocks, T, lds>>>(s, bases, x, y, per_block); }, 5); \
        printf("%-44s: %.3f ms  %.0f GB/s  %.1f Gnnz/s\n", name, t, gb / t * 1e3, n / t / 1e6);           \
    }
        RUN(0, 0, 1024, 4, "stream only, 1024 thr U4");
        RUN(0, 0, 1024, 8, "stream only, 1024 thr U8");
        RUN(0, 1, 1024, 4, "stream + sorted gather, U4");
        RUN(1, 0, 1024, 4, "stream + ds_add_f32, U4");
        RUN(1, 1, 1024, 4, "stream + sorted gather + ds_add_f32, U4");
        RUN(1, 1, 1024, 8, "stream + sorted gather + ds_add_f32, U8");
        RUN(2, 1, 1024, 4, "stream + sorted gather + plain store, U4");
        RUN(3, 1, 1024, 4, "stream + sorted gather + ds_min_i32, U4");
        RUN(1, 1, 512, 8, "same, 512 thr U8");
        RUN(4, 1, 1024, 4, "stream + gather + ds_add_u32 random, U4");
        RUN(5, 1, 1024, 4, "stream + gather + ds_add_u64 fixed-point, U4");
        RUN(6, 1, 1024, 4, "stream + gather + ds_add_f32 conflict-free, U4");
        RUN(7, 1, 1024, 4, "stream + gather + non-atomic RMW, U4");
        RUN(8, 1, 1024, 4, "stream + gather + ds_add_f64, U4");
        RUN(9, 1, 1024, 4, "stream + gather + ds_add_rtn_f32, U4");
        CK(hipFree(s));
        CK(hipFree(bases));
        CK(hipFree(x));
        CK(hipFree(y));
    }
    return 0;
}
