// Scratch microbenchmark 3: gather coalescing vs column gap for the row-block/column-sorted SpMV inner loop
// (ds_add_f64 into a double tile).  Tile rows and workgroup size are template parameters.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int ROWS, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void k(const uint2 *__restrict__ s, const uint32_t *__restrict__ bases,
                                             const float *__restrict__ x, float *__restrict__ y, size_t per_block) {
    extern __shared__ __attribute__((aligned(16))) double tile[];
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = 0.0;
    __syncthreads();
    const size_t begin = (size_t)blockIdx.x * per_block;
    for (size_t base = begin + threadIdx.x; base < begin + per_block; base += (size_t)THREADS * U) {
        u32x2 c[U]; uint32_t b[U]; float xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t kk = base + (size_t)u * THREADS; c[u] = __builtin_nontemporal_load((const u32x2 *)(s + kk)); b[u] = bases[kk >> 6]; }
#pragma unroll
        for (int u = 0; u < U; u++) xv[u] = x[b[u] + (c[u].x >> 14)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            double p = (double)(__uint_as_float(c[u].y) * xv[u]);
            __hip_atomic_fetch_add(&tile[c[u].x & (ROWS - 1)], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS; i += THREADS) y[(size_t)blockIdx.x * ROWS + i] = (float)tile[i];
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

template <int ROWS, int THREADS, int U>
static void run(const char *name, size_t n, size_t per_block, double gap, int xshare) {
    const size_t xn = 3000000;
    int nblocks = (int)(n / per_block);
    std::vector<uint2> h(n); std::vector<uint32_t> hb(n / 64);
    std::mt19937 rng(1);
    for (int b = 0; b < nblocks; b++) {
        double col = 0; double span = gap * per_block;            // columns swept by one block
        double start = (span >= xn) ? 0 : (double)((rng() % 1000) / 1000.0) * (xn - span);
        if (xshare == 1) start = 0;                                  // every block sweeps the same window
        if (xshare == 2) start = (b % 8) * (double)(xn - span) / 7;  // one window per XCD (block b -> XCD b%8)
        double g = (span >= xn) ? (double)(xn - 1000) / per_block : gap;
        for (size_t i = 0; i < per_block; i += 64) {
            uint32_t base = (uint32_t)(start + col);
            hb[(b * per_block + i) / 64] = base;
            for (int j = 0; j < 64; j++) {
                col += g * (0.5 + (rng() & 1023) / 1024.0);
                uint32_t off = (uint32_t)(start + col) - base;
                if (off > 0x3ffff) off = 0x3ffff;
                h[b * per_block + i + j] = make_uint2((off << 14) | (rng() & (ROWS - 1)), 0x3f800000u);
            }
        }
    }
    uint2 *s; uint32_t *bases; float *x, *y;
    CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&bases, hb.size() * 4)); CK(hipMalloc(&x, (xn + 2000000) * 4)); CK(hipMalloc(&y, (size_t)nblocks * ROWS * 4));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (xn + 2000000) * 4));
    size_t lds = ROWS * 8;
    CK(hipFuncSetAttribute((const void *)k<ROWS, THREADS, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double t = time_ms([&] { k<ROWS, THREADS, U><<<nblocks, THREADS, lds>>>(s, bases, x, y, per_block); }, 5);
    printf("%-28s gap %5.1f blocks %5d: %.3f ms  %.0f GB/s  %.1f Gnnz/s\n", name, gap, nblocks, t, n * 8 / 1e9 / t * 1e3, n / t / 1e6);
    CK(hipFree(s)); CK(hipFree(bases)); CK(hipFree(x)); CK(hipFree(y));
}

int main() {
    const size_t n = 128ull << 20;
    for (double gap : {1.5, 3.0, 6.0}) {
        for (int xs : {0, 1, 2}) {
            printf("xshare=%d (0 random windows, 1 same window, 2 window per XCD)\n", xs);
            run<16384, 1024, 4>("16K rows, 1024 thr, U4", n, 256 * 1024, gap, xs);
            run<16384, 1024, 8>("16K rows, 1024 thr, U8", n, 256 * 1024, gap, xs);
            run<8192, 512, 4>("8K rows, 512 thr, U4", n, 128 * 1024, gap, xs);
        }
    }
    return 0;
}
