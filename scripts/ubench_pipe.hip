// Scratch microbenchmark 9: the general SpMV inner loop (8-byte stream from HBM -> sorted gather of x -> ds_add_f64)
// as a BATCH loop ("load U groups, gather U, accumulate U": what spmv_rbcs_kernel did in round 1) versus a
// software-pipelined REGISTER RING (stream D groups ahead, gather G groups ahead, static slot indices so every
// s_waitcnt vmcnt(N) is exact).  Same data for both: orkut-like, 256 blocks x 12928 groups, gap 3.7.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int THREADS = 1024, ROWS = 12000;

__device__ __forceinline__ uint32_t load_const(const uint32_t *p) { return *(const __attribute__((address_space(4))) uint32_t *)(p); }

template <int U>
__global__ __launch_bounds__(THREADS) void k_batch(const uint2 *__restrict__ s, const uint32_t *__restrict__ bases, const float *__restrict__ x,
                                                   float *__restrict__ y, uint32_t gpb) {
    extern __shared__ __attribute__((aligned(16))) double tile[];
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = blockIdx.x * gpb;
    for (uint32_t g = wave; g < gpb; g += 16 * U) {
        u32x2 e[U]; uint32_t b[U]; float xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const uint32_t gi = g0 + min(g + u * 16, gpb - 1); e[u] = __builtin_nontemporal_load((const u32x2 *)(s + (size_t)gi * 64 + lane)); b[u] = bases[gi]; }
#pragma unroll
        for (int u = 0; u < U; u++) xv[u] = x[b[u] + (e[u].x >> 14)];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (g + u * 16 < gpb) __hip_atomic_fetch_add(&tile[e[u].x & 0x3fffu], (double)(__uint_as_float(e[u].y) * xv[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS; i += THREADS) y[(size_t)blockIdx.x * ROWS + i] = (float)tile[i];
}

// ring of D stream slots; the gather of a slot is issued G steps before the slot is consumed
template <int D, int G>
__global__ __launch_bounds__(THREADS) void k_ring(const uint2 *__restrict__ s, const uint32_t *__restrict__ bases, const float *__restrict__ x,
                                                  float *__restrict__ y, uint32_t gpb) {
    extern __shared__ __attribute__((aligned(16))) double tile[];
    for (int i = threadIdx.x; i < ROWS; i += THREADS) tile[i] = 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g0 = blockIdx.x * gpb, glast = gpb - 1;
    u32x2 e[D]; uint32_t b[D]; float xv[D];
    // pipeline fill = the steady-state step sequence without the consume part, in the same issue order, so that
    // the loop header sees the same queue (exact vmcnt) from the prologue and from the back edge
#pragma unroll
    for (int t = -D; t < 0; t++) {
        if (t + G >= 0) { const int dg = t + G; xv[dg] = x[b[dg] + (e[dg].x >> 14)]; __builtin_amdgcn_sched_barrier(0); }
        const int d = t + D;
        const uint32_t gi = g0 + min(wave + d * 16u, glast);
        e[d] = __builtin_nontemporal_load((const u32x2 *)(s + (size_t)gi * 64 + lane));
        b[d] = load_const(bases + gi);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t g = wave; g < gpb; g += 16 * D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            constexpr int dummy = 0; (void)dummy;
            const int dg = (d + G) % D;                       // slot whose gather is issued now
            // consume slot d
            const uint32_t gi = g + d * 16u;
            const float p = __uint_as_float(e[d].y) * xv[d];
            if (gi < gpb) __hip_atomic_fetch_add(&tile[e[d].x & 0x3fffu], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // gather for the group G steps ahead (its entries were loaded D - G steps ago)
            xv[dg] = x[b[dg] + (e[dg].x >> 14)];
            __builtin_amdgcn_sched_barrier(0);
            // refill slot d with the group D steps ahead
            const uint32_t gn = g0 + min(gi + 16u * D, glast);
            e[d] = __builtin_nontemporal_load((const u32x2 *)(s + (size_t)gn * 64 + lane));
            b[d] = load_const(bases + gn);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) asm volatile("" : : "v"(e[d].x), "v"(e[d].y), "v"(xv[d]));
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS; i += THREADS) y[(size_t)blockIdx.x * ROWS + i] = (float)tile[i];
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main() {
    const uint32_t xn = 3072512, gpb = 12928, nblocks = 256;
    const size_t ngroups = (size_t)nblocks * gpb, n = ngroups * 64;
    std::vector<uint2> h(n); std::vector<uint32_t> hb(ngroups);
    std::mt19937 rng(1);
    const double gap = (double)(xn - 4096) / ((double)gpb * 64);
    for (uint32_t bk = 0; bk < nblocks; bk++) {
        double col = 0;
        for (uint32_t g = 0; g < gpb; g++) {
            const uint32_t base = (uint32_t)col;
            hb[(size_t)bk * gpb + g] = base;
            for (int j = 0; j < 64; j++) {
                col += gap * (0.25 + 1.5 * (rng() & 1023) / 1024.0);
                h[((size_t)bk * gpb + g) * 64 + j] = make_uint2((((uint32_t)col - base) << 14) | (rng() % ROWS), 0x3f800000u);
            }
        }
    }
    uint2 *s; uint32_t *bases; float *x, *y;
    CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&bases, ngroups * 4)); CK(hipMalloc(&x, (size_t)(xn + 65536) * 4)); CK(hipMalloc(&y, (size_t)nblocks * ROWS * 4));
    CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bases, hb.data(), ngroups * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (size_t)(xn + 65536) * 4));
    const size_t lds = ROWS * 8;
#define RUNB(U) { CK(hipFuncSetAttribute((const void *)k_batch<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        double t = time_ms([&] { k_batch<U><<<nblocks, THREADS, lds>>>(s, bases, x, y, gpb); }, 5); \
        printf("batch U=%d              : %.3f ms  %.0f GB/s\n", U, t, n * 8 / 1e9 / t * 1e3); }
#define RUNR(D, G) { CK(hipFuncSetAttribute((const void *)k_ring<D, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        double t = time_ms([&] { k_ring<D, G><<<nblocks, THREADS, lds>>>(s, bases, x, y, gpb); }, 5); \
        printf("ring D=%d gather-ahead %d: %.3f ms  %.0f GB/s\n", D, G, t, n * 8 / 1e9 / t * 1e3); }
    RUNB(4); RUNB(6); RUNB(8);
    RUNR(6, 2); RUNR(8, 2); RUNR(8, 3); RUNR(8, 4); RUNR(10, 3); RUNR(12, 4); RUNR(12, 6); RUNR(16, 4); RUNR(16, 8);
    return 0;
}
