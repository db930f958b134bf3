// Scratch microbenchmark 8: why does "8-byte stream from HBM" + "sorted gather of x from L2" cost the SUM of
// the two?  Same instruction mix as the SpMV inner loop minus the LDS accumulate; sources are varied.
//   SRC_S: 0 none, 1 stream from HBM (nt), 2 stream from a 2 MB L2-resident buffer, 3 HBM stream without nt
//   SRC_X: 0 none, 1 gather sweeping a 12.8 MB x (L2 hits), 2 gather inside a 16 KB per-CU window (L1 hits)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 1024, U = 4;

template <int SRC_S, int SRC_X>
__global__ __launch_bounds__(THREADS) void k(const uint2 *__restrict__ s, const float *__restrict__ x, const uint32_t *__restrict__ offs,
                                             float *__restrict__ y, uint32_t groups_per_block, uint32_t xn) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = offs[u * 64 + lane];
    const size_t g0 = (size_t)blockIdx.x * groups_per_block;
    const double colstep = (double)(xn - 4096) / groups_per_block;   // the block sweeps all of x
    float acc = 0.f;
    uint32_t accu = 0;
    for (uint32_t g = wave; g < groups_per_block; g += 16 * U) {
        u32x2 e[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t gi = g0 + g + u * 16;
            if (SRC_S == 2) gi &= 4095;   // 4096 groups x 512 B = 2 MB
            if (SRC_S == 1 || SRC_S == 2) e[u] = __builtin_nontemporal_load((const u32x2 *)(s + gi * 64 + lane));
            if (SRC_S == 3) e[u] = *(const u32x2 *)(s + gi * 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t base = (uint32_t)((g + u * 16) * colstep);
            if (SRC_X == 2) base = (blockIdx.x * 8192u) + ((g + u * 16) * 240u & 3071u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (SRC_X) v[u] = x[(size_t)base + o[u]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (SRC_S) accu += e[u].x ^ e[u].y;
            if (SRC_X) acc += v[u];
        }
    }
    if (acc == 123.456f || accu == 0x12345u) y[0] = acc + accu;
}


// 16-byte-per-lane stream: one load instruction carries two groups (lane l gets entry l of group A and of group B)
template <int SRC_X, int NT>
__global__ __launch_bounds__(THREADS) void k4(const uint4 *__restrict__ s, const float *__restrict__ x, const uint32_t *__restrict__ offs,
                                              float *__restrict__ y, uint32_t pairs_per_block, uint32_t xn) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = offs[u * 64 + lane];
    const size_t p0 = (size_t)blockIdx.x * pairs_per_block;
    const double colstep = (double)(xn - 4096) / (pairs_per_block * 2);
    float acc = 0.f;
    uint32_t accu = 0;
    constexpr int UP = U / 2;
    for (uint32_t p = wave; p < pairs_per_block; p += 16 * UP) {
        u32x4 e[UP];
        float v[U];
#pragma unroll
        for (int u = 0; u < UP; u++) {
            const size_t pi = p0 + p + u * 16;
            e[u] = NT ? __builtin_nontemporal_load((const u32x4 *)(s + pi * 64 + lane)) : *(const u32x4 *)(s + pi * 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t base = (uint32_t)(((p + (u / 2) * 16) * 2 + (u & 1)) * colstep);
            base = __builtin_amdgcn_readfirstlane(base);
            if (SRC_X) v[u] = x[(size_t)base + o[u]];
        }
#pragma unroll
        for (int u = 0; u < UP; u++) accu += e[u].x ^ e[u].y ^ e[u].z ^ e[u].w;
#pragma unroll
        for (int u = 0; u < U; u++) if (SRC_X) acc += v[u];
    }
    if (acc == 123.456f || accu == 0x12345u) y[0] = acc + accu;
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main() {
    const uint32_t xn = 3200000, gpb = 12928;          // 256 blocks x 12928 groups x 64 = 211.8 M entries
    const size_t n = (size_t)256 * gpb * 64;
    uint2 *s; float *x, *y; uint32_t *offs;
    CK(hipMalloc(&s, n * 8)); CK(hipMemset(s, 1, n * 8));
    CK(hipMalloc(&x, (size_t)xn * 4 + (8u << 20))); CK(hipMemset(x, 0, (size_t)xn * 4 + (8u << 20))); CK(hipMalloc(&y, 4096)); CK(hipMalloc(&offs, U * 64 * 4));
    std::mt19937 rng(7);
    std::vector<uint32_t> h(U * 64);
    for (int u = 0; u < U; u++) { double c = 0; for (int i = 0; i < 64; i++) { c += 3.7 * (0.25 + 1.5 * (rng() % 1000) / 1000.0); h[u * 64 + i] = (uint32_t)c; } }
    CK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
#define RUN(S, X, name) { double t = time_ms([&] { k<S, X><<<256, THREADS>>>(s, x, offs, y, gpb, xn); }, 5); \
        printf("%-58s %.3f ms  (%.0f GB/s of 8-B entries)\n", name, t, n * 8 / 1e9 / t * 1e3); }
    RUN(1, 0, "stream HBM nt only");
    RUN(3, 0, "stream HBM (no nt) only");
    RUN(2, 0, "stream from 2 MB (L2) only");
    RUN(0, 1, "gather sweeping x (L2) only");
    RUN(0, 2, "gather in 16 KB window (L1) only");
    RUN(1, 1, "stream HBM nt + gather sweeping x (L2)");
    RUN(3, 1, "stream HBM (no nt) + gather sweeping x (L2)");
    RUN(2, 1, "stream L2 + gather sweeping x (L2)");
    RUN(1, 2, "stream HBM nt + gather in 16 KB window (L1)");
    RUN(2, 2, "stream L2 + gather L1");
#define RUN4(X, NT, name) { double t = time_ms([&] { k4<X, NT><<<256, THREADS>>>((const uint4 *)s, x, offs, y, gpb / 2, xn); }, 5); \
        printf("%-58s %.3f ms  (%.0f GB/s of 8-B entries)\n", name, t, n * 8 / 1e9 / t * 1e3); }
    RUN4(0, 1, "16-B/lane stream HBM nt only");
    RUN4(0, 0, "16-B/lane stream HBM only");
    RUN4(1, 1, "16-B/lane stream HBM nt + gather sweeping x (L2)");
    RUN4(1, 0, "16-B/lane stream HBM + gather sweeping x (L2)");
    return 0;
}
