// How long does a blocking call wait?  One empty kernel + (a) hipStreamSynchronize, (b) hipEventSynchronize on a recorded
// event, (c) a host spin on a page-locked word the kernel writes (system-scope store), (d) hipStreamWriteValue32 + spin.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_sync.hip -o build/ubench_sync && build/ubench_sync [spin]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(uint32_t *p) { if (p && threadIdx.x == 0) p[0] += 1u; }
__global__ void k_flag(volatile uint32_t *host_word, uint32_t v) {
    if (threadIdx.x == 0) { __hip_atomic_store((uint32_t *)host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char *what, std::vector<double> &t) {
    std::sort(t.begin(), t.end());
    printf("%-44s median %6.1f us  p10 %6.1f  p90 %6.1f\n", what, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10]);
}
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "spin")) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    if (argc > 1 && !strcmp(argv[1], "yield")) CK(hipSetDeviceFlags(hipDeviceScheduleYield));
    if (argc > 1 && !strcmp(argv[1], "block")) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    uint32_t *h; CK(hipHostMalloc(&h, 64, hipHostMallocMapped)); h[0] = 0;
    uint32_t *hd; CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 2000;
    std::vector<double> t(N);
    for (int i = 0; i < 200; i++) { k_empty<<<1, 64, 0, s>>>(d); } CK(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) { double t0 = now(); k_empty<<<1, 64, 0, s>>>(d); CK(hipStreamSynchronize(s)); t[i] = now() - t0; }
    report("launch + hipStreamSynchronize", t);
    for (int i = 0; i < N; i++) { double t0 = now(); k_empty<<<1, 64, 0, s>>>(d); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); t[i] = now() - t0; }
    report("launch + eventRecord + hipEventSynchronize", t);
    for (int i = 0; i < N; i++) { double t0 = now(); k_empty<<<1, 64, 0, s>>>(d); CK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) {} t[i] = now() - t0; }
    report("launch + eventRecord + hipEventQuery spin", t);
    for (int i = 0; i < N; i++) { double t0 = now(); k_empty<<<1, 64, 0, s>>>(d); while (hipStreamQuery(s) == hipErrorNotReady) {} t[i] = now() - t0; }
    report("launch + hipStreamQuery spin", t);
    for (int i = 0; i < N; i++) {
        double t0 = now(); k_flag<<<1, 64, 0, s>>>(hd, (uint32_t)(i + 1));
        while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != (uint32_t)(i + 1)) {}
        t[i] = now() - t0;
    }
    report("kernel stores a mapped word + host spin", t);
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) {
        double t0 = now(); k_empty<<<1, 64, 0, s>>>(d);
        hipError_t e = hipStreamWriteValue32(s, hd, (uint32_t)(0x1000000 + i), 0);
        if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); break; }
        while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != (uint32_t)(0x1000000 + i)) {}
        t[i] = now() - t0;
    }
    report("launch + hipStreamWriteValue32 + host spin", t);
    CK(hipStreamSynchronize(s));
    // four dependent launches then the wait (the SpMSpV call)
    for (int i = 0; i < N; i++) { double t0 = now(); for (int k = 0; k < 4; k++) k_empty<<<1, 64, 0, s>>>(d); CK(hipStreamSynchronize(s)); t[i] = now() - t0; }
    report("4 launches + hipStreamSynchronize", t);
    for (int i = 0; i < N; i++) {
        double t0 = now(); for (int k = 0; k < 3; k++) k_empty<<<1, 64, 0, s>>>(d); k_flag<<<1, 64, 0, s>>>(hd, (uint32_t)(0x2000000 + i));
        while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != (uint32_t)(0x2000000 + i)) {}
        t[i] = now() - t0;
    }
    report("4 launches, last stores a mapped word + spin", t);
    CK(hipStreamSynchronize(s));
    return 0;
}
