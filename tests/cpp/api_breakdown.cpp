// Where the time of a reference-style driver call goes on the HIP backend: the statements of BFS::pull
// (app/bfs.h:106-126) and PageRank::pull (app/pagerank.h:80-90), written against include/graphlily exactly as the
// reference's drivers write them, with a clock after every statement.  Prints one line per call; the first calls show
// the cold costs (fresh host blocks, first launches), the later ones the steady state.
//   api_breakdown <csr_float32.npz> [calls]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "graphlily/app/module_collection.h"
#include "graphlily/io/data_formatter.h"
#include "graphlily/io/data_loader.h"
#include "graphlily/module/add_scalar_vector_dense_module.h"
#include "graphlily/module/assign_vector_dense_module.h"
#include "graphlily/module/spmv_module.h"

using namespace graphlily;
using aligned_dense_vec_t = graphlily::aligned_dense_vec_t;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pull : public app::ModuleCollection {
    module::SpMVModule<val_t, val_t> *SpMV_;
    module::AssignVectorDenseModule<val_t> *DenseAssign_;
    module::eWiseAddModule<val_t> *eWiseAdd_;
    uint32_t n_ = 0;
    Pull(SemiringType sr, MaskType mt) {
        SpMV_ = new module::SpMVModule<val_t, val_t>(16, 1024000, 30720);
        SpMV_->set_semiring(sr);
        SpMV_->set_mask_type(mt);
        DenseAssign_ = new module::AssignVectorDenseModule<val_t>();
        DenseAssign_->set_mask_type(kMaskWriteToOne);
        eWiseAdd_ = new module::eWiseAddModule<val_t>();
        add_module(SpMV_);
        add_module(DenseAssign_);
        add_module(eWiseAdd_);
    }
};

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: api_breakdown <npz> [calls]\n"); return 2; }
    const int calls = argc > 2 ? atoi(argv[2]) : 4;
    io::CSRMatrix<float> m = io::load_csr_matrix_from_float_npz(argv[1]);
    io::util_round_csr_matrix_dim(m, 128, 128);
    for (auto &v : m.adj_data) v = 1;
    {
        Pull bfs(LogicalSemiring, kMaskWriteToZero);
        bfs.set_up_runtime("unused");
        bfs.SpMV_->load_and_format_matrix(m, true);
        bfs.SpMV_->send_matrix_host_to_device();
        const uint32_t n = bfs.SpMV_->get_num_rows(), iters = 6;
        aligned_dense_vec_t keep;
        for (int c = 0; c < calls; c++) {
            double t[8];
            t[0] = now_ms();
            aligned_dense_vec_t input(n, 0), distance(n, 0);
            input[0] = 1; distance[0] = 1;
            t[1] = now_ms();
            bfs.SpMV_->send_vector_host_to_device(input);
            bfs.SpMV_->send_mask_host_to_device(distance);
            t[2] = now_ms();
            bfs.DenseAssign_->bind_mask_buf(bfs.SpMV_->vector_buf);
            bfs.DenseAssign_->bind_inout_buf(bfs.SpMV_->mask_buf);
            bfs.eWiseAdd_->bind_in_buf(bfs.SpMV_->results_buf);
            bfs.eWiseAdd_->bind_out_buf(bfs.SpMV_->vector_buf);
            for (uint32_t it = 1; it <= iters; it++) {
                bfs.SpMV_->run();
                bfs.eWiseAdd_->run(n, 0);
                bfs.DenseAssign_->run(n, it + 1);
            }
            t[3] = now_ms();
            GRAPHLILY_CHECK(gl_sync());
            t[4] = now_ms();
            keep = bfs.SpMV_->send_mask_device_to_host();
            t[5] = now_ms();
            printf("BFS pull call %d: host vectors %.3f  uploads %.3f  enqueue 18 launches %.3f  wait %.3f  download+return %.3f  | total %.3f ms\n",
                   c, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
        }
    }
    {
        io::util_normalize_csr_matrix_by_outdegree(m);
        for (auto &v : m.adj_data) v *= 0.9f;
        Pull pr(ArithmeticSemiring, kNoMask);
        pr.set_up_runtime("unused");
        pr.SpMV_->load_and_format_matrix(m, true);
        pr.SpMV_->send_matrix_host_to_device();
        const uint32_t n = pr.SpMV_->get_num_rows(), iters = 10;
        aligned_dense_vec_t keep;
        for (int c = 0; c < calls; c++) {
            double t[8];
            t[0] = now_ms();
            aligned_dense_vec_t rank(n, 1.0 / n);
            t[1] = now_ms();
            pr.SpMV_->send_vector_host_to_device(rank);
            t[2] = now_ms();
            pr.eWiseAdd_->bind_in_buf(pr.SpMV_->results_buf);
            pr.eWiseAdd_->bind_out_buf(pr.SpMV_->vector_buf);
            for (uint32_t it = 1; it <= iters; it++) {
                pr.SpMV_->run();
                pr.eWiseAdd_->run(n, (1 - 0.9f) / n);
            }
            t[3] = now_ms();
            GRAPHLILY_CHECK(gl_sync());
            t[4] = now_ms();
            keep = pr.SpMV_->send_vector_device_to_host();
            t[5] = now_ms();
            printf("PageRank pull call %d: host vector %.3f  upload %.3f  enqueue 20 launches %.3f  wait %.3f  download+return %.3f  | total %.3f ms (%.4f per iteration)\n",
                   c, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0], (t[5] - t[0]) / iters);
        }
    }
    return 0;
}
