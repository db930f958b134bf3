// Round 4: what does a grid rendezvous through tagged words cost, by the memory operations used on each side?
// G workgroups of 1024 threads; workgroup b publishes word[b] = tag << 32 | b, then lanes 0 .. G-1 of it poll every word.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_rendezvous scripts/ubench_rendezvous.hip && build/ubench_rendezvous
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;

template <int V>
__device__ __forceinline__ void publish(u64 *w, u64 v) {
    if (V == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (V == 1) __hip_atomic_exchange(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (V == 2) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (V == 3) __hip_atomic_exchange(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else { *(volatile u64 *)w = v; __threadfence(); }
}
template <int V>
__device__ __forceinline__ u64 poll(u64 *w) {
    if (V == 0 || V == 1) return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (V == 2) return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (V == 3) return __hip_atomic_fetch_or(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}

template <int V>
__global__ __launch_bounds__(1024) void rendezvous(u64 *words, unsigned tag, u64 *stamps, int sleep) {
    const unsigned G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    const u64 t0 = wall_clock64();
    if (tid == 0) publish<V>(&words[b], ((u64)tag << 32) | b);
    unsigned spins = 0;
    if (tid < G) {
        u64 w;
        while ((unsigned)((w = poll<V>(&words[tid])) >> 32) != tag && ++spins < (1u << 22))
            if (sleep) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (tid == 0) {
        stamps[2 * b] = t0;
        stamps[2 * b + 1] = wall_clock64();
    }
}

template <int V>
void run(const char *name, int G, int sleep) {
    u64 *words, *stamps;
    hipMalloc((void **)&words, 8 * 1024);
    hipMemset(words, 0, 8 * 1024);
    hipMalloc((void **)&stamps, 16 * 1024);
    std::vector<u64> h(2 * 1024);
    std::vector<double> span;
    for (unsigned tag = 1; tag <= 30; tag++) {
        rendezvous<V><<<G, 1024>>>(words, tag, stamps, sleep);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), stamps, 16 * G, hipMemcpyDeviceToHost);
        u64 first = ~0ull, last = 0;
        for (int b = 0; b < G; b++) { first = std::min(first, h[2 * b]); last = std::max(last, h[2 * b + 1]); }
        if (tag > 5) span.push_back((double)(last - first) / 100.0);
    }
    std::sort(span.begin(), span.end());
    printf("%-44s G=%3d sleep=%d: first start -> last release  median %6.2f us  min %6.2f  max %6.2f\n", name, G, sleep, span[span.size() / 2], span.front(), span.back());
    hipFree(words); hipFree(stamps);
}

int main() {
    for (int G : {44, 256}) for (int sleep : {1, 0}) {
        run<0>("store agent / load agent", G, sleep);
        run<1>("exchange agent / load agent", G, sleep);
        run<2>("store system / load system", G, sleep);
        run<3>("exchange agent / fetch_or agent", G, sleep);
        run<4>("plain store + threadfence / acquire load", G, sleep);
    }
    return 0;
}
