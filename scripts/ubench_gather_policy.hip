// Scratch microbenchmark 12: cost of a SPARSE 64-lane dword gather (sorted lanes, mean gap 4 .. 26 floats, i.e. 8 .. 52 distinct
// 128-byte lines per instruction, every line used by one instruction only -- the cold groups of a low-degree graph such as the
// pokec stand-in) as a function of the cache-policy bits of the load: does any policy make the L1 fetch less than a whole
// 128-byte line per touched line?   hipcc -O3 --offload-arch=gfx950 scripts/ubench_gather_policy.hip -o /tmp/ubench_gp && /tmp/ubench_gp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int THREADS = 1024;
constexpr int U = 8;   // gathers in flight per wave

#define GLOAD(POL) asm volatile("global_load_dword %0, %1, %2 " POL : "=v"(v[u]) : "v"(o[u] * 4u), "s"(sb))

template <int FORM>
__global__ __launch_bounds__(THREADS) void k(const float *__restrict__ x, const uint32_t *__restrict__ offs, float *__restrict__ y,
                                             uint32_t iters, uint32_t step, uint32_t xn, uint32_t span) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = offs[u * 64 + lane];
    float acc = 0.f;
    // the 16 waves of a workgroup sweep consecutive spans (as the slots of a round do), workgroups start a little apart
    uint32_t base = (blockIdx.x * 977u) % 65536u + wave * span;
    for (uint32_t it = 0; it < iters; it++) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float *sb = x + __builtin_amdgcn_readfirstlane(base + u * step);
            if (FORM == 0) GLOAD("");
            else if (FORM == 1) GLOAD("nt");
            else if (FORM == 2) GLOAD("sc0");
            else if (FORM == 3) GLOAD("sc1");
            else if (FORM == 4) GLOAD("sc0 sc1");
            else if (FORM == 5) GLOAD("sc0 nt");
            else GLOAD("sc0 sc1 nt");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
        base += U * step;
        if (base > xn - 200000u) base -= (xn - 200000u);
    }
    if (acc == 123.456f) y[0] = acc;
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < iters; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main() {
    const uint32_t xn = 3200000;
    float *x, *y; uint32_t *offs;
    CK(hipMalloc(&x, (size_t)xn * 4 + 4194304)); CK(hipMemset(x, 0, (size_t)xn * 4 + 4194304)); CK(hipMalloc(&y, 4096)); CK(hipMalloc(&offs, U * 64 * 4));
    std::mt19937 rng(7);
    const double gaps[] = {1.5, 4.0, 8.0, 13.0, 26.0};
    const char *names[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc0 sc1 nt"};
    for (double gap : gaps) {
        std::vector<uint32_t> h(U * 64);
        double lines = 0;
        for (int u = 0; u < U; u++) {
            double c = 0;
            uint32_t last_line = 0xffffffffu;
            for (int i = 0; i < 64; i++) {
                c += gap * (0.25 + 1.5 * (rng() % 1000) / 1000.0);
                h[u * 64 + i] = (uint32_t)c;
                if ((uint32_t)c / 32u != last_line) { lines++; last_line = (uint32_t)c / 32u; }
            }
        }
        CK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const uint32_t span = (uint32_t)(64 * gap) + 32u;   // columns one gather covers
        const uint32_t iters = 600, step = 16u * span;       // the next gather of this wave continues 16 spans on: no line is touched twice
        printf("gap %5.1f (%4.1f lines / gather):", gap, lines / U);
        double t[7];
        t[0] = time_ms([&] { k<0><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        t[1] = time_ms([&] { k<1><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        t[2] = time_ms([&] { k<2><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        t[3] = time_ms([&] { k<3><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        t[4] = time_ms([&] { k<4><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        t[5] = time_ms([&] { k<5><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        t[6] = time_ms([&] { k<6><<<256, THREADS>>>(x, offs, y, iters, step, xn, span); }, 3);
        // clocks per wave-gather per CU at 2.4 GHz: t * 2.4e6 / (16 waves * iters * U)
        for (int f = 0; f < 7; f++) printf("  %s %5.1f", names[f], t[f] * 2.4e6 / (16.0 * iters * U));
        printf("  clk/gather/CU\n");
    }
    return 0;
}
