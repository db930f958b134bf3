// Scratch microbenchmark 6: like ubench_window2 but with a fully static software pipeline so that every
// s_waitcnt vmcnt(N) is exact: ONE round (16 groups) per window, windows fetched D rounds ahead into D
// register sets (float2 x NV per thread = NV*2048 floats), stream ring D deep, one barrier per round.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int THREADS = 1024, WAVES = 16;

template <int ROWS, int NV, int D, typename ACC>
__global__ __launch_bounds__(THREADS) void k(const uint2 *__restrict__ s, const uint32_t *__restrict__ los,
                                             const float *__restrict__ x, float *__restrict__ y, uint32_t rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    ACC *tile = reinterpret_cast<ACC *>(lds);
    constexpr uint32_t WIN = NV * 2048u;
    float *win = reinterpret_cast<float *>(lds + (size_t)ROWS * sizeof(ACC));   // 2 x WIN floats
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = (ACC)0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t *lo = los + (size_t)blockIdx.x * rounds;
    const uint2 *st = s + (size_t)blockIdx.x * rounds * (WAVES * 64u) + wave * 64u + lane;
    const uint32_t last = rounds - 1u;
    float2 stage[D][NV];
    u32x2 e[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t r = min((uint32_t)d, last);
        const float *src = x + lo[r] + threadIdx.x * 2u;
#pragma unroll
        for (int v = 0; v < NV; v++) stage[d][v] = *reinterpret_cast<const float2 *>(src + v * 2048u);
        e[d] = __builtin_nontemporal_load((const u32x2 *)(st + (size_t)r * (WAVES * 64u)));
    }
    for (uint32_t j = 0; j < rounds; j += D) {   // rounds is a multiple of D
#pragma unroll
        for (int d = 0; d < D; d++) {
            float *w = win + ((d & 1) ? WIN : 0u);   // D even
#pragma unroll
            for (int v = 0; v < NV; v++) *reinterpret_cast<float2 *>(w + v * 2048u + threadIdx.x * 2u) = stage[d][v];
            __syncthreads();
            const float xv = w[e[d].x >> 14];
            const float p = __uint_as_float(e[d].y) * xv;
            if (sizeof(ACC) == 8)
                __hip_atomic_fetch_add((double *)&tile[e[d].x & 0x3FFFu], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                atomicMin((int *)&tile[e[d].x & 0x3FFFu], __float_as_int(p));
            const uint32_t r = min(j + d + D, last);
            const float *src = x + lo[r] + threadIdx.x * 2u;
#pragma unroll
            for (int v = 0; v < NV; v++) stage[d][v] = *reinterpret_cast<const float2 *>(src + v * 2048u);
            e[d] = __builtin_nontemporal_load((const u32x2 *)(st + (size_t)r * (WAVES * 64u)));
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

template <int ROWS, int NV, int D, typename ACC>
static void run(const char *name, int nblocks, uint32_t rounds, double gap, int shift = 0) {
    const size_t xn = 3200000;
    const uint32_t WIN = NV * 2048u;
    rounds = rounds / D * D;
    const size_t n = (size_t)nblocks * rounds * 1024;
    std::vector<uint2> h(n);
    std::vector<uint32_t> hl((size_t)nblocks * rounds);
    std::mt19937 rng(1);
    double span = 1024.0 * gap;
    if (span > WIN - 32) { printf("%-26s gap %.1f: window too small\n", name, gap); return; }
    double step = std::min(span, (double)(xn - WIN) / rounds);
    for (int b = 0; b < nblocks; b++)
        for (uint32_t r = 0; r < rounds; r++) {
            hl[(size_t)b * rounds + r] = ((uint32_t)(r * step) + (shift ? (b * 1056u) % 16384u : 0u)) & ~31u;
            for (size_t i = 0; i < 1024; i++) {
                const uint32_t off = (uint32_t)((rng() % 100000) / 100000.0 * span);
                h[((size_t)b * rounds + r) * 1024 + i] = make_uint2((off << 14) | (rng() % ROWS), 0x3f800000u);
            }
        }
    uint2 *s; uint32_t *ld; float *x, *y;
    CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&ld, hl.size() * 4)); CK(hipMalloc(&x, (xn + 131072) * 4)); CK(hipMalloc(&y, (size_t)nblocks * ROWS * 4));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(ld, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (xn + 131072) * 4));
    size_t lds = (size_t)ROWS * sizeof(ACC) + 2 * WIN * 4;
    CK(hipFuncSetAttribute((const void *)k<ROWS, NV, D, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double t = time_ms([&] { k<ROWS, NV, D, ACC><<<nblocks, THREADS, lds>>>(s, ld, x, y, rounds); }, 5);
    printf("%-26s D%d gap %4.1f shift %d blocks %4d rounds %4u lds %3zuK: %.3f ms  %.0f GB/s matrix, x staged %.0f GB/s\n", name, D, gap, shift,
           nblocks, rounds, lds / 1024, t, n * 8 / 1e9 / t * 1e3, (double)nblocks * rounds * WIN * 4 / 1e9 / t * 1e3);
    CK(hipFree(s)); CK(hipFree(ld)); CK(hipFree(x)); CK(hipFree(y));
}

int main() {
    run<12000, 4, 4, double>("R12K W8K f64", 256, 816, 3.7);
    run<12000, 4, 4, double>("R12K W8K f64", 256, 816, 3.7, 1);
    run<12000, 4, 6, double>("R12K W8K f64", 256, 816, 3.7);
    run<12000, 4, 2, double>("R12K W8K f64", 256, 816, 3.7);
    run<12000, 3, 4, double>("R12K W6K f64", 256, 816, 3.7);
    run<12000, 2, 4, double>("R12K W4K f64", 256, 816, 3.7);
    run<12000, 2, 6, double>("R12K W4K f64", 256, 816, 3.7);
    run<9600, 5, 4, double>("R9.6K W10K f64", 256, 480, 5.0);
    run<9600, 5, 6, double>("R9.6K W10K f64", 256, 480, 5.0);
    run<16000, 6, 4, float>("R16K W12K i32min", 256, 816, 3.0);
    run<16000, 4, 4, float>("R16K W8K i32min", 256, 816, 3.0);
    run<12000, 4, 4, double>("R12K W8K f64 512blk", 512, 408, 3.7);
    return 0;
}
