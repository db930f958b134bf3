// Host-side costs around the module API's host vectors (12 MB = one orkut-sized dense vector):
// CPU fill / copy of pageable vs page-locked memory (by hipHostMalloc flag), and the PCIe copies from / to each.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const size_t n = 3072512, bytes = n * 4;
    void *d = nullptr;
    CK(hipMalloc(&d, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Kind { const char *name; unsigned flags; bool pinned; };
    Kind kinds[] = {{"pageable(posix_memalign)", 0, false}, {"hipHostMallocDefault", hipHostMallocDefault, true},
                    {"hipHostMallocNonCoherent", hipHostMallocNonCoherent, true}, {"hipHostMallocCoherent", hipHostMallocCoherent, true},
                    {"hipHostMallocNumaUser", hipHostMallocNumaUser, true}, {"hipHostMallocPortable", hipHostMallocPortable, true},
                    {"malloc+hipHostRegister", 0xffffffffu, true}};
    float *dst = nullptr;
    if (posix_memalign((void **)&dst, 4096, bytes)) return 1;
    memset(dst, 0, bytes);
    for (const Kind &k : kinds) {
        float *h = nullptr;
        double t_alloc = now();
        if (!k.pinned) { if (posix_memalign((void **)&h, 4096, bytes)) return 1; }
        else if (k.flags == 0xffffffffu) { if (posix_memalign((void **)&h, 4096, bytes)) return 1; memset(h, 0, bytes); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); }
        else if (hipHostMalloc((void **)&h, bytes, k.flags) != hipSuccess) { printf("%-28s not available\n", k.name); (void)hipGetLastError(); continue; }
        t_alloc = now() - t_alloc;
        double best[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9};
        for (int rep = 0; rep < 6; rep++) {
            double t = now();
            for (size_t i = 0; i < n; i++) h[i] = 1.0f / (float)n;          // std::vector(n, v) fill
            best[0] = std::min(best[0], now() - t);
            t = now(); memcpy(dst, h, bytes); best[1] = std::min(best[1], now() - t);   // read it back out (return-by-value copy)
            t = now(); memcpy(h, dst, bytes); best[2] = std::min(best[2], now() - t);
            t = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); best[3] = std::min(best[3], now() - t);
            t = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); best[4] = std::min(best[4], now() - t);
            t = now(); { volatile float acc = 0; float a = 0; for (size_t i = 0; i < n; i += 16) a += h[i]; acc = a; (void)acc; } best[5] = std::min(best[5], now() - t);
        }
        printf("%-28s alloc %7.3f ms | fill %6.3f  memcpy-out %6.3f  memcpy-in %6.3f  H2D %6.3f  D2H %6.3f  strided-read %6.3f ms\n", k.name, t_alloc * 1e3,
               best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3, best[4] * 1e3, best[5] * 1e3);
    }
    // device allocation costs
    for (int rep = 0; rep < 3; rep++) {
        void *q; double t = now(); CK(hipMalloc(&q, bytes)); double a = now() - t; t = now(); CK(hipFree(q)); double f = now() - t;
        printf("hipMalloc(12 MB) %.3f ms  hipFree %.3f ms\n", a * 1e3, f * 1e3);
    }
    // large pageable upload: what a plan creation from host CSR pays
    const size_t big = 848u << 20;
    char *hb = (char *)malloc(big); memset(hb, 1, big);
    void *db; CK(hipMalloc(&db, big));
    for (int rep = 0; rep < 2; rep++) { double t = now(); CK(hipMemcpy(db, hb, big, hipMemcpyHostToDevice)); printf("hipMemcpy H2D pageable %zu MB: %.1f ms = %.1f GB/s\n", big >> 20, (now() - t) * 1e3, big / (now() - t) / 1e9); }
    void *hp; CK(hipHostMalloc(&hp, big, hipHostMallocDefault)); memcpy(hp, hb, big);
    for (int rep = 0; rep < 2; rep++) { double t = now(); CK(hipMemcpy(db, hp, big, hipMemcpyHostToDevice)); printf("hipMemcpy H2D pinned   %zu MB: %.1f ms = %.1f GB/s\n", big >> 20, (now() - t) * 1e3, big / (now() - t) / 1e9); }
    { double t = now(); memcpy(hp, hb, big); printf("memcpy pageable->pinned 1 thread: %.1f ms = %.1f GB/s\n", (now() - t) * 1e3, big / (now() - t) / 1e9); }
    return 0;
}
