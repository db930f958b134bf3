// Scratch microbenchmark 7: cost of a 64-lane dword gather on gfx950 as a function of the address pattern and
// of the instruction form (64-bit vaddr vs saddr + 32-bit voffset vs buffer_load offen).  No matrix stream:
// per-lane offsets live in registers, a moving base sweeps a window of x that stays L2-resident.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int THREADS = 1024;
constexpr int U = 8;   // gathers in flight per wave

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 0));
}

// offs: U per-lane offsets (floats) per wave pattern, [U][64]
template <int FORM>
__global__ __launch_bounds__(THREADS) void k(const float *__restrict__ x, const uint32_t *__restrict__ offs, float *__restrict__ y,
                                             uint32_t iters, uint32_t step, uint32_t xn) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = offs[u * 64 + lane];
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)(xn * 4u), 0x00020000);
    float acc = 0.f;
    uint32_t base = (blockIdx.x * 977u + wave * 4099u) % 65536u;   // every wave elsewhere in the window
    for (uint32_t it = 0; it < iters; it++) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t b = __builtin_amdgcn_readfirstlane(base + u * step);
            if (FORM == 0) v[u] = x[(size_t)b + o[u]];                       // compiler's choice
            else if (FORM == 1) {   // saddr + 32-bit voffset
                const float *sb = x + b;
                asm volatile("global_load_dword %0, %1, %2" : "=v"(v[u]) : "v"(o[u] * 4u), "s"(sb));
            }
            else v[u] = buf_load(rsrc, (b + o[u]) * 4u);
        }
        if (FORM == 1)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
        base += U * step;
        if (base > xn - 70000u) base -= (xn - 70000u);
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
    CK(hipMalloc(&x, (size_t)xn * 4 + 1048576)); CK(hipMemset(x, 0, (size_t)xn * 4 + 1048576)); CK(hipMalloc(&y, 4096)); CK(hipMalloc(&offs, U * 64 * 4));
    std::mt19937 rng(7);
    struct Pat { const char *name; int kind; double gap; };
    const Pat pats[] = {{"coalesced i", 0, 1}, {"permuted within 64", 1, 1}, {"stride 2", 2, 2}, {"stride 4", 2, 4}, {"stride 8", 2, 8},
                        {"sorted gaps mean 1.5", 3, 1.5}, {"sorted gaps mean 3.7", 3, 3.7}, {"sorted gaps mean 6", 3, 6}, {"sorted gaps mean 12", 3, 12},
                        {"pairs same addr, gap 3.7", 4, 3.7}, {"quads same addr, gap 3.7", 5, 3.7}, {"all lanes same addr", 6, 0},
                        {"random in 8K window", 7, 0}, {"sorted 3.7, 16 active lanes", 8, 3.7}, {"sorted 3.7, 32 active lanes", 9, 3.7}};
    for (const Pat &p : pats) {
        std::vector<uint32_t> h(U * 64);
        for (int u = 0; u < U; u++) {
            double c = 0;
            std::vector<uint32_t> perm(64);
            for (int i = 0; i < 64; i++) perm[i] = i;
            for (int i = 63; i > 0; i--) std::swap(perm[i], perm[rng() % (i + 1)]);
            for (int i = 0; i < 64; i++) {
                uint32_t v = 0;
                switch (p.kind) {
                    case 0: v = i; break;
                    case 1: v = perm[i]; break;
                    case 2: v = (uint32_t)(i * p.gap); break;
                    case 3: case 8: case 9: c += p.gap * (0.25 + 1.5 * (rng() % 1000) / 1000.0); v = (uint32_t)c; break;
                    case 4: if ((i & 1) == 0) c += 2 * p.gap * (0.25 + 1.5 * (rng() % 1000) / 1000.0); v = (uint32_t)c; break;
                    case 5: if ((i & 3) == 0) c += 4 * p.gap * (0.25 + 1.5 * (rng() % 1000) / 1000.0); v = (uint32_t)c; break;
                    case 6: v = 5; break;
                    case 7: v = rng() % 8192; break;
                }
                if (p.kind == 8 && i >= 16) v = 0xffffffffu;   // marker: lane disabled through exec? use OOB for buffer form only
                if (p.kind == 9 && i >= 32) v = 0xffffffffu;
                h[u * 64 + i] = v;
            }
        }
        // for the "active lanes" patterns, inactive lanes read the SAME address as lane 0 (cheapest possible extra lanes)
        for (int u = 0; u < U; u++) for (int i = 0; i < 64; i++) if (h[u * 64 + i] == 0xffffffffu) h[u * 64 + i] = h[u * 64];
        CK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const uint32_t iters = 2000, step = 240;
        const double lanes = 256.0 * 16 * 64 * (double)iters * U;
        printf("%-30s", p.name);
        double t0 = time_ms([&] { k<0><<<256, THREADS>>>(x, offs, y, iters, step, xn); }, 3);
        double t1 = time_ms([&] { k<1><<<256, THREADS>>>(x, offs, y, iters, step, xn); }, 3);
        double t2 = time_ms([&] { k<2><<<256, THREADS>>>(x, offs, y, iters, step, xn); }, 3);
        // clocks per wave-gather per CU at 2.4 GHz: t * 2.4e6 / (16 waves * iters * U)
        printf(" vaddr64 %6.1f Glanes/s (%5.1f clk/gather/CU) | saddr %6.1f (%5.1f) | buffer %6.1f (%5.1f)\n", lanes / t0 / 1e6,
               t0 * 2.4e6 / (16.0 * iters * U), lanes / t1 / 1e6, t1 * 2.4e6 / (16.0 * iters * U), lanes / t2 / 1e6, t2 * 2.4e6 / (16.0 * iters * U));
    }
    return 0;
}
