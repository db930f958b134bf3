// Compile-only translation unit (used in THIS container only, where /root/reference exists): the
// reference's own app drivers, unmodified, against this repo's graphlily/ headers.  Build line:
//   g++ -std=c++11 -I<repo>/include -I/root/reference tests/cpp/ref_apps_compile.cpp -L... -lgraphlily_hip
// "graphlily/module/*.h", "graphlily/global.h", "graphlily/io/*.h" and "graphlily/app/module_collection.h"
// resolve to <repo>/include (first -I); only the three app headers come from the reference tree.
#include "graphlily/app/bfs.h"
#include "graphlily/app/pagerank.h"
#include "graphlily/app/sssp.h"

#include <cstdio>
#include <cstdlib>
#include <cmath>

template <typename A, typename B>
static int compare(const A &ref, const B &got, const char *what, float eps) {
    if (ref.size() != got.size()) { printf("%s: size mismatch\n", what); return 1; }
    for (size_t i = 0; i < ref.size(); i++)
        if (std::fabs(float(got[i]) - ref[i]) > eps * std::fmax(1.0f, std::fabs(ref[i]))) {
            printf("%s: mismatch at %zu: ref %g got %g\n", what, i, ref[i], float(got[i]));
            return 1;
        }
    printf("%s OK\n", what);
    return 0;
}

// -DGRAPHLILY_VAL_UFIXED (the reference's shipped val_t): distances are sums of weights rounded to 2^-24 each, against the
// float sums of compute_reference_results -- a few 1e-7 apart, not equal; BFS levels and the PageRank bound stay as they are.
#if defined(GRAPHLILY_VAL_UFIXED)
static const float kSsspEps = 2e-6f;
#else
static const float kSsspEps = 0.f;
#endif

// With a GPU this runs the reference drivers end to end on the HIP backend (argv[1] = npz path).
int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s graph.npz\n", argv[0]); return 2; }
    int bad = 0;
    {
        graphlily::app::BFS bfs(graphlily::num_hbm_channels, 1024, 512, 256);
        bfs.set_target("hw");
        bfs.set_up_runtime("unused.xclbin");
        bfs.load_and_format_matrix(argv[1], true);
        bfs.send_matrix_host_to_device();
        auto ref = bfs.compute_reference_results(0, 8);
        bad += compare(ref, bfs.pull_push(0, 8, 0.1f), "reference BFS::pull_push", 0.f);
        bad += compare(ref, bfs.pull(0, 8), "reference BFS::pull", 0.f);
        bad += compare(ref, bfs.push(0, 8), "reference BFS::push", 0.f);
    }
    {
        graphlily::app::PageRank pr(graphlily::num_hbm_channels, 1024, 256);
        pr.set_target("hw");
        pr.set_up_runtime("unused.xclbin");
        pr.load_and_format_matrix(argv[1], 0.9f, true);
        pr.send_matrix_host_to_device();
        bad += compare(pr.compute_reference_results(0.9f, 10), pr.pull(0.9f, 10), "reference PageRank::pull", 1e-4f);
    }
    {
        graphlily::app::SSSP sssp(graphlily::num_hbm_channels, 1024, 512, 256);
        sssp.set_target("hw");
        sssp.set_up_runtime("unused.xclbin");
        sssp.load_and_format_matrix(argv[1], true);
        sssp.send_matrix_host_to_device();
        auto ref = sssp.compute_reference_results(0, 8);
        bad += compare(ref, sssp.pull_push(0, 8, 0.1f), "reference SSSP::pull_push", kSsspEps);
        bad += compare(ref, sssp.pull(0, 8), "reference SSSP::pull", kSsspEps);
        bad += compare(ref, sssp.push(0, 8), "reference SSSP::push", kSsspEps);
    }
    return bad ? 1 : 0;
}
