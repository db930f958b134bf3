// How fast can kernels store 4-byte values at scattered positions of a page-locked HOST array (zero-copy over PCIe)?
// Motivation: a BFS on orkut spends 0.22 of its 0.55 ms reading the 12 MB distance vector back; if the steps
// wrote each new distance to a host mirror as well (2.1 M scattered stores per run), only the unreached entries would be left.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_pcie_scatter.hip -o build/ubench_pcie_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void scatter_kernel(float *dst, const uint32_t *idx, uint32_t n, float v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[idx[i]] = v;
}
__global__ void linear_kernel(float *dst, uint32_t n, float v) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = v;
}

int main() {
    const uint32_t n = 3072512, m = 2097152;   // array length, scattered stores per launch
    float *host = nullptr, *dev = nullptr;
    CK(hipHostMalloc((void **)&host, (size_t)n * 4, hipHostMallocDefault));
    CK(hipMalloc((void **)&dev, (size_t)n * 4));
    std::vector<uint32_t> h_idx(m);
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = 0; i < m; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h_idx[i] = (uint32_t)(s % n); }
    uint32_t *d_idx = nullptr;
    CK(hipMalloc((void **)&d_idx, (size_t)m * 4));
    CK(hipMemcpy(d_idx, h_idx.data(), (size_t)m * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *host_dev = nullptr;
    CK(hipHostGetDevicePointer((void **)&host_dev, host, 0));
    for (int target = 0; target < 2; target++) {
        float *dst = target ? host_dev : dev;
        for (uint32_t count : {2097152u, 262144u, 32768u}) {
            for (int grid : {256, 2048}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    CK(hipEventRecord(e0));
                    scatter_kernel<<<grid, 256>>>(dst, d_idx, count, (float)rep);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("%s scattered stores: %8u in %8.1f us (%6.1f M/s), grid %d\n", target ? "HOST  " : "device", count, best * 1e3f, count / best / 1e3f, grid);
            }
        }
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            linear_kernel<<<2048, 256>>>(dst, n, (float)rep);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s linear fill of %u floats: %8.1f us (%5.1f GB/s)\n", target ? "HOST  " : "device", n, best * 1e3f, n * 4.0f / best / 1e6f);
    }
    // sanity: the last value written is visible on the host
    CK(hipDeviceSynchronize());
    printf("host[idx[0]] = %g (expect 4)\n", host[h_idx[0]]);
    return 0;
}
