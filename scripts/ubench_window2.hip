// Scratch microbenchmark 5: row-block SpMV inner loop with the x window of every CHUNK (K rounds of 16 groups)
// staged through registers into a double-buffered LDS window, stream prefetched D rounds ahead in a register
// ring, ONE barrier per chunk.  Compare with ubench_gap (global gather) and ubench_win (unpipelined, negative).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int THREADS = 1024, WAVES = 16;

// chunk table per block: {first round of the NEXT chunk, lo column (multiple of 32)}; windows hold WV*4096 floats
template <int ROWS, int WV, int D, typename ACC, int P, int ROT>
__global__ __launch_bounds__(THREADS) void k(const uint2 *__restrict__ s, const uint2 *__restrict__ chunks,
                                             const float *__restrict__ x, float *__restrict__ y, uint32_t rounds_per_block,
                                             uint32_t chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    ACC *tile = reinterpret_cast<ACC *>(lds);
    constexpr uint32_t WIN = WV * 4096u;
    float *win = reinterpret_cast<float *>(lds + (size_t)ROWS * sizeof(ACC));   // 2 x WIN floats
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = (ACC)0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint2 *ch = chunks + (size_t)blockIdx.x * chunks_per_block;
    constexpr uint32_t NC = P > 0 ? WAVES - 1 : WAVES;   // consumer waves; wave 15 warms L2 P chunks ahead
    const uint2 *st = s + (size_t)blockIdx.x * rounds_per_block * (NC * 64u) + wave * 64u + lane;
    const uint32_t rot = ROT ? ((threadIdx.x + blockIdx.x * 72u) & 1023u) : threadIdx.x;   // de-phase the CUs: 9 lines apart
    float4 stage[WV];
    auto fetch = [&](uint32_t c) {
        const uint32_t lo = ch[c].y;
#pragma unroll
        for (int v = 0; v < WV; v++) stage[v] = *reinterpret_cast<const float4 *>(x + lo + v * 4096u + rot * 4u);
    };
    auto commit = [&](float *dst) {
#pragma unroll
        for (int v = 0; v < WV; v++) *reinterpret_cast<float4 *>(dst + v * 4096u + rot * 4u) = stage[v];
    };
    fetch(0);
    commit(win);
    __syncthreads();
    if (P > 0 && wave == NC) {
        float sink = 0.f;
        float t[P > 0 ? P : 1][WV];
        for (uint32_t c0 = 1; c0 < chunks_per_block; c0 += P) {
#pragma unroll
            for (int q = 0; q < P; q++) {
                const uint32_t c = c0 + q;
                if (c < chunks_per_block) {
                    const uint32_t cp = min(c + (uint32_t)P, chunks_per_block - 1u);
                    const uint32_t lo = ch[cp].y;
#pragma unroll
                    for (int v = 0; v < WV; v++) { sink += t[q][v]; t[q][v] = x[lo + v * 4096u + lane * 32u + (q & 1) * 2048u]; }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (sink == 123.456f) y[0] = sink;
        return;
    }
    uint32_t c = 0, cend = ch[0].x;
    const uint32_t last_c = chunks_per_block - 1u, last_r = rounds_per_block - 1u;
    fetch(min(1u, last_c));
    u32x2 e[D];
#pragma unroll
    for (int d = 0; d < D; d++) e[d] = __builtin_nontemporal_load((const u32x2 *)(st + (size_t)min((uint32_t)d, last_r) * (NC * 64u)));
    for (uint32_t j = 0; j < rounds_per_block; j += D) {   // rounds_per_block is a multiple of D
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint32_t jj = j + d;
            if (jj == cend) {
                c++;
                commit(win + (c & 1u) * WIN);
                __syncthreads();
                cend = ch[c].x;
                fetch(min(c + 1u, last_c));
            }
            const float xv = win[(c & 1u) * WIN + (e[d].x >> 14)];
            const float p = __uint_as_float(e[d].y) * xv;
            if (sizeof(ACC) == 8)
                __hip_atomic_fetch_add((double *)&tile[e[d].x & 0x3FFFu], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                atomicMin((int *)&tile[e[d].x & 0x3FFFu], __float_as_int(p));
            e[d] = __builtin_nontemporal_load((const u32x2 *)(st + (size_t)min(jj + D, last_r) * (NC * 64u)));
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

// K rounds per chunk; a chunk's entries fall uniformly into [lo, lo + K*1024*gap)
template <int ROWS, int WV, int D, typename ACC, int P = 0, int ROT = 0>
static void run(const char *name, int nblocks, uint32_t rounds, int K, double gap, int shift = 0) {
    const size_t xn = 3200000;
    const uint32_t WIN = WV * 4096u;
    rounds = rounds / (D * K) * (D * K);
    const uint32_t nch = rounds / K;
    const uint32_t RE = (P > 0 ? 15 : 16) * 64;   // entries per round
    const size_t n = (size_t)nblocks * rounds * RE;
    std::vector<uint2> h(n), hc((size_t)nblocks * nch);
    std::mt19937 rng(1);
    double span = K * (double)RE * gap;
    if (span > WIN - 32) { printf("%-30s gap %.1f K %d: window too small\n", name, gap, K); return; }
    double step = std::min(span, (double)(xn - WIN) / nch);   // sweep the whole x like a real block
    for (int b = 0; b < nblocks; b++)
        for (uint32_t cc = 0; cc < nch; cc++) {
            const uint32_t lo = ((uint32_t)(cc * step) + (shift ? (b * 1056u) % 16384u : 0u)) & ~31u;
            hc[(size_t)b * nch + cc] = make_uint2((cc + 1) * K, lo);
            for (size_t i = 0; i < (size_t)K * RE; i++) {
                const uint32_t off = (uint32_t)((rng() % 100000) / 100000.0 * span);
                h[((size_t)b * rounds + (size_t)cc * K) * RE + i] = make_uint2((off << 14) | (rng() % ROWS), 0x3f800000u);
            }
        }
    uint2 *s, *chd; float *x, *y;
    CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&chd, hc.size() * 8)); CK(hipMalloc(&x, (xn + 131072) * 4)); CK(hipMalloc(&y, (size_t)nblocks * ROWS * 4));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(chd, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (xn + 65536) * 4));
    size_t lds = (size_t)ROWS * sizeof(ACC) + 2 * WIN * 4;
    CK(hipFuncSetAttribute((const void *)k<ROWS, WV, D, ACC, P, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double t = time_ms([&] { k<ROWS, WV, D, ACC, P, ROT><<<nblocks, THREADS, lds>>>(s, chd, x, y, rounds, nch); }, 5);
    printf("%-30s P%d ROT%d gap %4.1f K %d blocks %4d rounds %4u lds %3zuK: %.3f ms  %.0f GB/s matrix, x staged %.0f GB/s\n", name, P, ROT, gap, K, nblocks,
           rounds, lds / 1024, t, n * 8 / 1e9 / t * 1e3, (double)nblocks * nch * WIN * 4 / 1e9 / t * 1e3);
    CK(hipFree(s)); CK(hipFree(chd)); CK(hipFree(x)); CK(hipFree(y));
}

int main() {
    run<8192, 2, 6, double, 0, 0>("R8K W8K D6 f64", 256, 816, 2, 3.7);
    run<8192, 2, 6, double, 0, 1>("R8K W8K D6 f64", 256, 816, 2, 3.7);
    run<8192, 2, 6, double, 0, 0>("R8K W8K D6 f64 shift", 256, 816, 2, 3.7, 1);
    run<8192, 2, 6, double, 0, 1>("R8K W8K D6 f64 shift", 256, 816, 2, 3.7, 1);
    run<8192, 2, 6, double, 4, 1>("R8K W8K D6 f64 shift", 256, 816, 2, 3.7, 1);
    run<8192, 1, 6, double, 0, 1>("R8K W4K D6 f64", 256, 816, 1, 3.7);
    run<8192, 1, 6, double, 8, 1>("R8K W4K D6 f64", 256, 816, 1, 3.7);
    run<11776, 2, 6, double, 0, 1>("R11.5K W8K D6 f64", 256, 816, 2, 3.7);
    run<11776, 2, 6, double, 4, 1>("R11.5K W8K D6 f64", 256, 816, 2, 3.7);
    run<11776, 2, 6, double, 0, 1>("R11.5K W8K D6 f64", 256, 816, 1, 7.0);
    run<8192, 3, 6, double, 0, 1>("R8K W12K D6 f64", 256, 816, 2, 5.6);
    run<16256, 3, 6, float, 0, 1>("R16K W12K D6 i32min", 256, 816, 3, 3.7);
    return 0;
}
