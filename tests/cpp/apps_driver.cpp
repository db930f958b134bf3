// The app drivers of THIS repo (include/graphlily/app/{bfs,sssp,pagerank}.h: the reference's classes over the device-resident
// schedules) end to end: every mode's result is written as raw 32-bit words for tests/test_cpp_layer.py to compare with the
// oracle bit for bit, and checked here against the class's own compute_reference_results.
//   apps_driver graph.npz out_dir iterations [dist]
// `dist`: additionally a world-of-one RCCL communicator (gl_dist_unique_id / gl_dist_init) behind BFS::set_comm -- the sharded
// schedule's code path (row shard plans, one gl_dist_all_gather_bits_tally per slot inside the recorded graph) driven from C++.
//   g++ -std=c++11 -I<repo>/include tests/cpp/apps_driver.cpp -L<repo>/graphlily_amd/lib -lgraphlily_hip
#include "graphlily/app/bfs.h"
#include "graphlily/app/pagerank.h"
#include "graphlily/app/sssp.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

template <typename V>
static void dump(const std::string &dir, const char *name, const V &v) {
    const std::string p = dir + "/" + name + ".bin";
    FILE *f = fopen(p.c_str(), "wb");
    if (!f || fwrite(v.data(), sizeof(v[0]), v.size(), f) != v.size()) {
        printf("cannot write %s\n", p.c_str());
        exit(2);
    }
    fclose(f);
}

template <typename A, typename B>
static int same(const A &ref, const B &got, const char *what, float eps) {
    if (ref.size() != got.size()) { printf("%s: size mismatch\n", what); return 1; }
    for (size_t i = 0; i < ref.size(); i++)
        if (std::fabs(float(got[i]) - ref[i]) > eps * std::fmax(1.0f, std::fabs(ref[i]))) {
            printf("%s: mismatch at %zu: ref %g got %g\n", what, i, ref[i], float(got[i]));
            return 1;
        }
    printf("%s OK\n", what);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) { printf("usage: %s graph.npz out_dir iterations [dist]\n", argv[0]); return 2; }
    const std::string npz = argv[1], out = argv[2];
    const uint32_t iters = (uint32_t)atoi(argv[3]);
    const bool dist = argc > 4 && !strcmp(argv[4], "dist");
    int bad = 0;
    {
        graphlily::app::BFS bfs(graphlily::num_hbm_channels, 1024, 512, 256);
        bfs.set_target("hw");
        bfs.set_up_runtime("unused.xclbin");
        bfs.load_and_format_matrix(npz, true);
        bfs.send_matrix_host_to_device();
        auto ref = bfs.compute_reference_results(0, iters);
        {   // zero iterations as the FIRST call (the schedule's buffers did not exist yet: a null control block went
            // to the library) -- the reference returns the start vector (bfs.h:106-127)
            auto p0 = bfs.pull(0, 0);
            bad += same(bfs.compute_reference_results(0, 0), p0, "BFS::pull with 0 iterations as the first call", 0.f);
            // ... and calls that alternate between the nibble (<= 14 iterations) and the byte read-back keep their own buffers
            for (int rep = 0; rep < 2; rep++) {
                bad += same(bfs.compute_reference_results(0, 3), bfs.pull(0, 3), "BFS::pull, 3 iterations (nibbles)", 0.f);
                bad += same(bfs.compute_reference_results(0, 16), bfs.pull(0, 16), "BFS::pull, 16 iterations (bytes)", 0.f);
            }
        }
        for (int rep = 0; rep < 3; rep++) {   // first call enqueues and records, the others replay the hipGraph
            auto pp = bfs.pull_push(0, iters, 0.001f);
            auto pl = bfs.pull(0, iters);
            if (rep == 2) {
                bad += same(ref, pp, "BFS::pull_push (device-resident schedule, replayed)", 0.f);
                bad += same(ref, pl, "BFS::pull (device-resident schedule, replayed)", 0.f);
                dump(out, "bfs_pull_push", pp);
                dump(out, "bfs_pull", pl);
            }
        }
        auto ph = bfs.push(0, iters);
        bad += same(ref, ph, "BFS::push", 0.f);
        dump(out, "bfs_push", ph);
        auto other = bfs.pull_push(0, iters, 0.5f);      // another threshold: another recorded schedule, same levels
        bad += same(ref, other, "BFS::pull_push threshold 0.5", 0.f);
        auto tb = bfs.pull_push_time_breakdown(0, iters, 0.001f);
        bad += same(ref, tb, "BFS::pull_push_time_breakdown", 0.f);
    }
    if (dist) {
        unsigned char id[128];
        gl_dist comm = nullptr;
        GRAPHLILY_CHECK(gl_init(0));
        GRAPHLILY_CHECK(gl_dist_unique_id(id));
        GRAPHLILY_CHECK(gl_dist_init(&comm, 0, 1, id));
        {
            graphlily::app::BFS bfs(graphlily::num_hbm_channels, 1024, 512, 256);
            bfs.set_target("hw");
            bfs.set_up_runtime("unused.xclbin");
            bfs.set_comm(comm);
            bfs.load_and_format_matrix(npz, true);
            bfs.send_matrix_host_to_device();
            auto ref = bfs.compute_reference_results(0, iters);
            for (int rep = 0; rep < 3; rep++) {
                auto pp = bfs.pull_push(0, iters, 0.001f);
                if (rep == 2) {
                    bad += same(ref, pp, "BFS::pull_push through gl_dist_* (world of one, exchange inside the graph)", 0.f);
                    dump(out, "bfs_pull_push_dist", pp);
                }
            }
        }   // (the schedule's graphs go before the communicator: gl_dist_destroy refuses while one is alive)
        GRAPHLILY_CHECK(gl_dist_destroy(comm));
    }
    {
        graphlily::app::SSSP sssp(graphlily::num_hbm_channels, 1024, 512, 256);
        sssp.set_target("hw");
        sssp.set_up_runtime("unused.xclbin");
        sssp.load_and_format_matrix(npz, true);
        sssp.send_matrix_host_to_device();
        auto ref = sssp.compute_reference_results(0, iters);
        auto pp = sssp.pull_push(0, iters, 0.001f), pl = sssp.pull(0, iters), ph = sssp.push(0, iters);
        bad += same(ref, pp, "SSSP::pull_push", 0.f);
        bad += same(ref, pl, "SSSP::pull", 0.f);
        bad += same(ref, ph, "SSSP::push", 0.f);
        bad += same(ref, sssp.pull_push(0, iters, 0.001f), "SSSP::pull_push again", 0.f);
        dump(out, "sssp_pull_push", pp);
        dump(out, "sssp_pull", pl);
        dump(out, "sssp_push", ph);
    }
    {
        graphlily::app::PageRank pr(graphlily::num_hbm_channels, 1024, 256);
        pr.set_target("hw");
        pr.set_up_runtime("unused.xclbin");
        pr.load_and_format_matrix(npz, 0.9f, true);
        pr.send_matrix_host_to_device();
        auto ref = pr.compute_reference_results(0.9f, 10);
        auto got = pr.pull(0.9f, 10);
        bad += same(ref, got, "PageRank::pull", 1e-4f);
        bad += same(ref, pr.pull_time_breakdown(0.9f, 10), "PageRank::pull_time_breakdown", 1e-4f);
        dump(out, "pagerank", got);
    }
    return bad ? 1 : 0;
}
