// Scratch microbenchmark 5: decomposition of the SpMV inner loop.  GATHER: 0 none, 1 global (sorted, gap 3),
// 2 LDS table (32 KB).  ACC: 0 none, 1 ds_add_f64, 2 ds_min_i32, 3 ds_add_f64 into few rows (hub), 4 register sum.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int THREADS = 1024, WAVES = 16, ROWS = 12288, HOT = 8192;

template <int GATHER, int ACC, int U>
__global__ __launch_bounds__(THREADS) void k(const uint2 *__restrict__ s, const uint32_t *__restrict__ bases,
                                             const float *__restrict__ x, float *__restrict__ y, uint32_t groups_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *hot = reinterpret_cast<float *>(lds);
    double *tile = reinterpret_cast<double *>(lds + HOT * 4);
    for (int i = threadIdx.x; i < HOT; i += THREADS) hot[i] = x[i * 7];
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = blockIdx.x * groups_per_block;
    float racc = 0.f;
    for (uint32_t g = wave; g < groups_per_block; g += WAVES * U) {
        u32x2 e[U]; uint32_t b[U]; float xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t gi = g0 + g + u * WAVES;
            e[u] = __builtin_nontemporal_load((const u32x2 *)(s + (size_t)gi * 64u + lane));
            b[u] = (GATHER == 1) ? bases[gi] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (GATHER == 1) xv[u] = x[b[u] + (e[u].x >> 14)];
            else if (GATHER == 2) xv[u] = hot[(e[u].x >> 14) & (HOT - 1)];
            else xv[u] = 1.0f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float p = __uint_as_float(e[u].y) * xv[u];
            const uint32_t r = (e[u].x & 0x3fffu) % ROWS;
            if (ACC == 1) __hip_atomic_fetch_add(&tile[r], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ACC == 2) atomicMin((int *)&tile[r], __float_as_int(p));
            if (ACC == 3) __hip_atomic_fetch_add(&tile[r & 7u], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ACC == 4) racc += p;
            if (ACC == 0 && p == 123.456f) tile[r] = p;
        }
    }
    __syncthreads();
    if (racc == 123.456f) tile[0] = racc;
    for (int i = threadIdx.x; i < ROWS; i += THREADS) y[(size_t)blockIdx.x * ROWS + i] = (float)tile[i];
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main() {
    const size_t n = 128ull << 20, per_block = 512 * 1024;
    const int nblocks = (int)(n / per_block);
    const double gap = 3.0;
    std::vector<uint2> h(n); std::vector<uint32_t> hb(n / 64);
    std::mt19937 rng(1);
    for (int b = 0; b < nblocks; b++) {
        double col = 0;
        for (size_t i = 0; i < per_block; i += 64) {
            uint32_t base = (uint32_t)col; hb[(b * per_block + i) / 64] = base;
            for (int j = 0; j < 64; j++) {
                col += gap * (0.5 + (rng() & 1023) / 1024.0);
                h[b * per_block + i + j] = make_uint2((((uint32_t)col - base) << 14) | (rng() & 0x3fffu), 0x3f800000u);
            }
        }
    }
    uint2 *s; uint32_t *bases; float *x, *y;
    CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&bases, hb.size() * 4)); CK(hipMalloc(&x, 5000000 * 4)); CK(hipMalloc(&y, (size_t)nblocks * ROWS * 4));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, 5000000 * 4));
    const size_t lds = HOT * 4 + ROWS * 8;
#define RUN(G, A, U, name) { CK(hipFuncSetAttribute((const void *)k<G, A, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        double t = time_ms([&] { k<G, A, U><<<nblocks, THREADS, lds>>>(s, bases, x, y, (uint32_t)(per_block / 64)); }, 5); \
        printf("%-52s: %.3f ms  %.0f GB/s  %.1f Gnnz/s\n", name, t, n * 8 / 1e9 / t * 1e3, n / t / 1e6); }
    RUN(0, 0, 4, "stream");
    RUN(0, 4, 4, "stream + register sum");
    RUN(0, 1, 4, "stream + ds_add_f64");
    RUN(0, 2, 4, "stream + ds_min_i32");
    RUN(0, 3, 4, "stream + ds_add_f64 (8 rows only)");
    RUN(1, 0, 4, "stream + global gather");
    RUN(1, 4, 4, "stream + global gather + register sum");
    RUN(1, 1, 4, "stream + global gather + ds_add_f64");
    RUN(1, 2, 4, "stream + global gather + ds_min_i32");
    RUN(2, 0, 4, "stream + LDS-table gather");
    RUN(2, 4, 4, "stream + LDS-table gather + register sum");
    RUN(2, 1, 4, "stream + LDS-table gather + ds_add_f64");
    RUN(2, 2, 4, "stream + LDS-table gather + ds_min_i32");
    RUN(2, 1, 8, "stream + LDS-table gather + ds_add_f64, U8");
    RUN(1, 1, 8, "stream + global gather + ds_add_f64, U8");
    return 0;
}
