// Scratch microbenchmark (round 6, not part of the product): what ONE LDS atomic / read instruction of a wavefront costs the CU's
// LDS pipe, as a function of the operation, the number and placement of active lanes and the address pattern.  rocprofv3's
// counters on the pattern-layout SpMV (profiles/r06_pmc_pattern_orkut.txt) say the LDS is 75 % busy and a ds_add_f64 on random
// rows costs ~22 LDS cycles of which ~8 are bank conflicts: which part of that follows the active lanes?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -Wno-atomic-alignment -o build/ubench_lds_atomic scripts/ubench_lds_atomic.hip
// One 1024-thread workgroup per CU (16 wavefronts, like the SpMV kernels), every wavefront issues ITERS x 8 operations back to
// back on a 120 KB tile; reported: LDS clocks per wavefront instruction = wall clocks of the CU x ... / (16 x ITERS x 8).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kTileBytes = 120 * 1024;
constexpr int kIters = 512;

// OP: 0 ds_add_f64, 1 ds_add_u32, 2 ds_min_i32, 3 ds_read_b32, 4 ds_add_u64, 5 ds_add_f32, 6 ds_write_b32, 7 ds_read_b64, 8 ds_add_rtn_f64
// ADDR: 0 random, 1 lane-linear (conflict-free), 2 all lanes one address, 3 random but PAIRS of lanes share an address,
//       4 random within a 2 KB window (many duplicates / same-bank hits)
// LANES: active-lane mask selector: 0 all 64, 1 every 4th (16 spread), 2 lanes 0-15 (16 compact), 3 every 16th (4 spread),
//        4 lanes 0-3 (4 compact), 5 lanes 0-31, 6 every 2nd (32 spread)
template <int OP, int ADDR, int LANES>
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ rnd, unsigned long long *__restrict__ clocks, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < kTileBytes / 8; i += 1024) reinterpret_cast<double *>(lds)[i] = 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    bool active = true;
    if (LANES == 1) active = (lane & 3u) == 0u;
    if (LANES == 2) active = lane < 16u;
    if (LANES == 3) active = (lane & 15u) == 0u;
    if (LANES == 4) active = lane < 4u;
    if (LANES == 5) active = lane < 32u;
    if (LANES == 6) active = (lane & 1u) == 0u;
    constexpr uint32_t esz = (OP == 0 || OP == 4 || OP == 7 || OP == 8) ? 8u : 4u;
    const uint32_t nelem = kTileBytes / esz;
    uint32_t a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t r = rnd[(blockIdx.x * 1024u + threadIdx.x) * 8u + j];
        if (ADDR == 1) r = threadIdx.x + 1024u * j;
        if (ADDR == 2) r = 1000u * (threadIdx.x >> 6) + j;
        if (ADDR == 3) r = rnd[(blockIdx.x * 1024u + (threadIdx.x & ~1u)) * 8u + j];
        if (ADDR == 4) r = (r & 255u) + 256u * ((threadIdx.x >> 6) + 16u * j);
        a[j] = r % nelem;   // element index
    }
    float acc = 0.0f;
    double *td = reinterpret_cast<double *>(__builtin_assume_aligned(lds, 16));
    float *tf = reinterpret_cast<float *>(__builtin_assume_aligned(lds, 16));
    uint32_t *tu = reinterpret_cast<uint32_t *>(__builtin_assume_aligned(lds, 16));
    unsigned long long *tq = reinterpret_cast<unsigned long long *>(__builtin_assume_aligned(lds, 16));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (active) {
        for (int it = 0; it < kIters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t ad = a[j];
                if (OP == 0) __hip_atomic_fetch_add(&td[ad], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OP == 1) __hip_atomic_fetch_add(&tu[ad], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OP == 2) __hip_atomic_fetch_min(reinterpret_cast<int *>(&tu[ad]), (int)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OP == 3) acc += tf[ad];
                if (OP == 4) __hip_atomic_fetch_add(&tq[ad], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OP == 5) __hip_atomic_fetch_add(&tf[ad], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (OP == 6) tf[ad] = (float)it;
                if (OP == 7) acc += (float)td[ad];
                if (OP == 8) acc += (float)__hip_atomic_fetch_add(&td[ad], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            asm volatile("" ::: "memory");   // (loads of loop-invariant addresses stay in the loop)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int OP, int ADDR, int LANES>
static void run(const char *what, const uint32_t *d_rnd, unsigned long long *d_clk, float *d_sink, int cus) {
    CK(hipFuncSetAttribute((const void *)k<OP, ADDR, LANES>, hipFuncAttributeMaxDynamicSharedMemorySize, kTileBytes));
    k<OP, ADDR, LANES><<<cus, 1024, kTileBytes>>>(d_rnd, d_clk, d_sink);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 4; rep++) k<OP, ADDR, LANES><<<cus, 1024, kTileBytes>>>(d_rnd, d_clk, d_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(cus);
    CK(hipMemcpy(c.data(), d_clk, cus * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : c) s += (double)v;
    // s_memtime / readcyclecounter on gfx9 counts at a constant 100 MHz: convert with the measured core clock?  Report both the raw
    // counter per instruction and a ratio against the first line, which is what matters
    printf("%-56s %8.2f s_memtime ticks | %7.2f ns (events, incl. launch + tile clear) per wavefront instruction of the CU\n", what,
           s / cus / (16.0 * kIters * 8), ms / 4 * 1e6 / (16.0 * kIters * 8));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<uint32_t> rnd((size_t)cus * 1024 * 8);
    uint64_t st = 88172645463325252ull;
    for (auto &v : rnd) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        v = (uint32_t)(st >> 16);
    }
    uint32_t *d_rnd;
    unsigned long long *d_clk;
    float *d_sink;
    CK(hipMalloc(&d_rnd, rnd.size() * 4));
    CK(hipMemcpy(d_rnd, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_clk, cus * 8));
    CK(hipMalloc(&d_sink, 4));
    printf("%d CUs, clock %d kHz; the counter ticks at 100 MHz: ticks x clock / 100 MHz = core clocks\n", cus, prop.clockRate);
#define R(OP, ADDR, LANES, TXT) run<OP, ADDR, LANES>(TXT, d_rnd, d_clk, d_sink, cus)
    R(0, 0, 0, "ds_add_f64 random, 64 lanes");
    R(0, 1, 0, "ds_add_f64 lane-linear (conflict-free), 64 lanes");
    R(0, 2, 0, "ds_add_f64 one address per wavefront, 64 lanes");
    R(0, 3, 0, "ds_add_f64 random, lane pairs share an address");
    R(0, 4, 0, "ds_add_f64 random within 2 KB windows");
    R(0, 0, 5, "ds_add_f64 random, lanes 0-31");
    R(0, 0, 6, "ds_add_f64 random, every 2nd lane (32)");
    R(0, 0, 1, "ds_add_f64 random, every 4th lane (16)");
    R(0, 0, 2, "ds_add_f64 random, lanes 0-15");
    R(0, 0, 3, "ds_add_f64 random, every 16th lane (4)");
    R(0, 0, 4, "ds_add_f64 random, lanes 0-3");
    R(8, 0, 0, "ds_add_rtn_f64 random, 64 lanes");
    R(4, 0, 0, "ds_add_u64 random, 64 lanes");
    R(1, 0, 0, "ds_add_u32 random, 64 lanes");
    R(1, 1, 0, "ds_add_u32 lane-linear, 64 lanes");
    R(2, 0, 0, "ds_min_i32 random, 64 lanes");
    R(5, 0, 0, "ds_add_f32 random, 64 lanes");
    R(5, 1, 0, "ds_add_f32 lane-linear, 64 lanes");
    R(3, 0, 0, "ds_read_b32 random, 64 lanes");
    R(3, 1, 0, "ds_read_b32 lane-linear, 64 lanes");
    R(3, 4, 0, "ds_read_b32 random within 1 KB windows");
    R(7, 0, 0, "ds_read_b64 random, 64 lanes");
    R(6, 0, 0, "ds_write_b32 random, 64 lanes");
    return 0;
}
