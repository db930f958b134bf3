// Scratch microbenchmark 4: stage the x window of each round in LDS (global_load_lds, 16 B/lane) and gather
// from LDS, vs gathering from global.  Row-block tile in LDS as f64 (ds_add_f64), 1024 threads, U = 1.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int THREADS = 1024, WAVES = 16;

// MODE 0: global gather; MODE 1: LDS window via global_load_lds (double buffered, one barrier per round)
template <int ROWS, int WIN, int MODE>
__global__ __launch_bounds__(THREADS) void k(const uint2 *__restrict__ s, const uint32_t *__restrict__ bases,
                                             const float *__restrict__ x, float *__restrict__ y, uint32_t groups_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double *tile = reinterpret_cast<double *>(lds);
    float *win = reinterpret_cast<float *>(lds + (size_t)ROWS * 8);   // 2 x WIN floats
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = 0.0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = blockIdx.x * groups_per_block;
    const uint32_t nrounds = groups_per_block / WAVES;
    auto stage = [&](uint32_t r, float *dst) {
        const uint32_t cmin = bases[g0 + r * WAVES] & ~3u;
        uint32_t cmax = bases[g0 + (r + 1) * WAVES];                // first column of the next round (sorted)
        if (r + 1 >= nrounds || cmax < cmin) cmax = cmin + WIN - 256u;  // last round of the block: whole window
        uint32_t nchunk = (cmax - cmin + 256u) / 256u;              // 256 floats = 1 KiB per wave instruction
        if (nchunk > WIN / 256u) nchunk = WIN / 256u;
        for (uint32_t c = wave; c < nchunk; c += WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + cmin + c * 256u + lane * 4u),
                                             (__attribute__((address_space(3))) void *)(dst + c * 256u), 16, 0, 0);
    };
    if (MODE == 1) stage(0, win);
    __syncthreads();
    for (uint32_t r = 0; r < nrounds; r++) {
        const uint32_t gi = g0 + r * WAVES + wave;
        u32x2 c = __builtin_nontemporal_load((const u32x2 *)(s + (size_t)gi * 64u + lane));
        const uint32_t b = bases[gi];
        float xv;
        if (MODE == 1) {
            if (r + 1 < nrounds) stage(r + 1, win + ((r + 1) & 1u) * WIN);
            const uint32_t cmin = bases[g0 + r * WAVES] & ~3u;
            xv = win[(r & 1u) * WIN + (b - cmin) + (c.x >> 14)];
        } else {
            xv = x[b + (c.x >> 14)];
        }
        __hip_atomic_fetch_add(&tile[c.x & (ROWS - 1)], (double)(__uint_as_float(c.y) * xv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __syncthreads();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS; i += THREADS) y[(size_t)blockIdx.x * ROWS + i] = (float)tile[i];
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

template <int ROWS, int WIN, int MODE>
static void run(const char *name, size_t n, size_t per_block, double gap) {
    const size_t xn = 3000000;
    int nblocks = (int)(n / per_block);
    std::vector<uint2> h(n); std::vector<uint32_t> hb(n / 64 + 64);
    std::mt19937 rng(1);
    for (int b = 0; b < nblocks; b++) {
        double col = 0;
        for (size_t i = 0; i < per_block; i += 64) {
            uint32_t base = (uint32_t)col;
            hb[(b * per_block + i) / 64] = base;
            for (int j = 0; j < 64; j++) {
                col += gap * (0.5 + (rng() & 1023) / 1024.0);
                uint32_t off = (uint32_t)col - base;
                h[b * per_block + i + j] = make_uint2((off << 14) | (rng() & (ROWS - 1)), 0x3f800000u);
            }
        }
    }
    for (size_t i = n / 64; i < n / 64 + 64; i++) hb[i] = hb[n / 64 - 1] + 64;
    uint2 *s; uint32_t *bases; float *x, *y;
    CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&bases, hb.size() * 4)); CK(hipMalloc(&x, (xn + 2000000) * 4)); CK(hipMalloc(&y, (size_t)nblocks * ROWS * 4));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (xn + 2000000) * 4));
    size_t lds = (size_t)ROWS * 8 + 2 * WIN * 4;
    CK(hipFuncSetAttribute((const void *)k<ROWS, WIN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double t = time_ms([&] { k<ROWS, WIN, MODE><<<nblocks, THREADS, lds>>>(s, bases, x, y, (uint32_t)(per_block / 64)); }, 5);
    printf("%-36s gap %4.1f: %.3f ms  %.0f GB/s  %.1f Gnnz/s\n", name, gap, t, n * 8 / 1e9 / t * 1e3, n / t / 1e6);
    CK(hipFree(s)); CK(hipFree(bases)); CK(hipFree(x)); CK(hipFree(y));
}

int main() {
    const size_t n = 128ull << 20;
    for (double gap : {1.5, 3.0, 5.0}) {
        run<8192, 8192, 0>("8K rows, global gather", n, 512 * 1024, gap);
        run<8192, 8192, 1>("8K rows, LDS window (32 KB x2)", n, 512 * 1024, gap);
        run<16384, 4096, 1>("16K rows, LDS window (16 KB x2)", n, 512 * 1024, gap > 3.5 ? 3.5 : gap);
    }
    return 0;
}
