// graphlily/app/pagerank.h -- PageRank over the MI355X backend with the reference's class (graphlily/app/pagerank.h:17-160
// of the reference: same name, constructor, public methods), so that benchmark/bench_pagerank.cpp compiles unmodified with
// -I<this repo>/include in front.  The rank vector starts on the DEVICE (the reference builds n floats on the host and uploads
// them per call, pagerank.h:81-82); an iteration -- SpMV, then eWiseAdd of the teleport term, results -> vector (:84-88) -- runs
// as ONE SpMV whose epilogue adds the teleport term (passed as the semiring's zero: zero + sum is the float eWiseAdd produces)
// and a swap of the two buffers (module/fusion.h, for modules owned by a ModuleCollection).
// -DGRAPHLILY_USE_REFERENCE_APPS: the next graphlily/app/pagerank.h on the include path (the reference checkout's) instead.
#if defined(GRAPHLILY_USE_REFERENCE_APPS)
#include_next "graphlily/app/pagerank.h"
#else
#ifndef GRAPHLILY_HIP_APP_PAGERANK_H_
#define GRAPHLILY_HIP_APP_PAGERANK_H_
#define GRAPHLILY_APP_PAGERANK_H_   // (the reference's guard)

#include "graphlily/app/module_collection.h"
#include "graphlily/module/spmv_module.h"
#include "graphlily/module/add_scalar_vector_dense_module.h"
#include "graphlily/io/data_loader.h"
#include "graphlily/io/data_formatter.h"

#include <chrono>
#include <iostream>

namespace graphlily {
namespace app {

class PageRank : public app::ModuleCollection {
private:
    graphlily::module::SpMVModule<graphlily::val_t, graphlily::val_t> *SpMV_;
    graphlily::module::eWiseAddModule<graphlily::val_t> *eWiseAdd_;
    uint32_t matrix_num_rows_ = 0, matrix_num_cols_ = 0;
    uint32_t num_channels_, spmv_out_buf_len_, vec_buf_len_;
    graphlily::SemiringType semiring_ = graphlily::ArithmeticSemiring;
    using aligned_dense_vec_t = graphlily::aligned_dense_vec_t;
    using aligned_sparse_vec_t = graphlily::aligned_sparse_vec_t;
    using aligned_dense_float_vec_t = graphlily::aligned_dense_float_vec_t;
    typedef graphlily::value_kind<graphlily::val_t> VK;

    // rank = 1 / n everywhere (over the PADDED n, pagerank.h:81), on the device
    void start_() {
        const graphlily::val_t r0 = 1.0 / matrix_num_rows_;
        DeviceBuffer rank(sizeof(graphlily::val_t) * (size_t)matrix_num_rows_);
        GRAPHLILY_CHECK(gl_buf_fill_u32((uint32_t *)rank.ptr(), VK::bits(r0), matrix_num_rows_));
        SpMV_->bind_vector_buf(rank);
        eWiseAdd_->bind_in_buf(SpMV_->results_buf);
        eWiseAdd_->bind_out_buf(SpMV_->vector_buf);
    }

public:
    PageRank(uint32_t num_channels, uint32_t spmv_out_buf_len, uint32_t vec_buf_len)
        : num_channels_(num_channels), spmv_out_buf_len_(spmv_out_buf_len), vec_buf_len_(vec_buf_len) {
        SpMV_ = new graphlily::module::SpMVModule<graphlily::val_t, graphlily::val_t>(num_channels_, spmv_out_buf_len_, vec_buf_len_);
        SpMV_->set_semiring(semiring_);
        SpMV_->set_mask_type(graphlily::kNoMask);
        add_module(SpMV_);
        eWiseAdd_ = new graphlily::module::eWiseAddModule<graphlily::val_t>();
        add_module(eWiseAdd_);
    }

    uint32_t get_nnz() { return SpMV_->get_nnz(); }

    void load_and_format_matrix(std::string csr_float_npz_path, float damping, bool skip_empty_rows) {
        CSRMatrix<float> csr_matrix = graphlily::io::load_csr_matrix_from_float_npz(csr_float_npz_path);
        graphlily::io::util_round_csr_matrix_dim(csr_matrix, num_channels_ * graphlily::pack_size, num_channels_ * graphlily::pack_size);
        graphlily::io::util_normalize_csr_matrix_by_outdegree(csr_matrix);
        for (auto &x : csr_matrix.adj_data) x = x * damping;
        SpMV_->load_and_format_matrix(csr_matrix, skip_empty_rows);
        matrix_num_rows_ = SpMV_->get_num_rows();
        matrix_num_cols_ = SpMV_->get_num_cols();
        assert(matrix_num_rows_ == matrix_num_cols_);
    }

    void send_matrix_host_to_device() { SpMV_->send_matrix_host_to_device(); }

    aligned_dense_vec_t pull(graphlily::val_t damping, uint32_t num_iterations) {
        start_();
        SpMV_->chain(true);   // every result is the next vector: the run's epilogue prepares the next run's packed x
        for (uint32_t iter = 1; iter <= num_iterations; iter++) {
            SpMV_->run();
            eWiseAdd_->run(matrix_num_rows_, (1 - damping) / matrix_num_rows_);   // the teleport term, the reference's expression (:87)
        }
        aligned_dense_vec_t rank = SpMV_->send_vector_device_to_host();
        SpMV_->chain(false);
        return rank;
    }

    // the reference's buckets (pagerank.h:93-147), every call followed by a device synchronisation
    aligned_dense_vec_t pull_time_breakdown(graphlily::val_t damping, uint32_t num_iterations) {
        typedef std::chrono::high_resolution_clock clk;
        float spmv_ms = 0, ewise_ms = 0, transfer_ms = 0;
        auto timed = [](float &bucket, const std::function<void()> &fn) {
            GRAPHLILY_CHECK(gl_sync());
            const auto t0 = clk::now();
            fn();
            GRAPHLILY_CHECK(gl_sync());
            bucket += float(std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count()) / 1000;
        };
        const auto start = clk::now();
        timed(transfer_ms, [&] { start_(); });
        for (uint32_t iter = 1; iter <= num_iterations; iter++) {
            timed(spmv_ms, [&] { SpMV_->run(); GRAPHLILY_CHECK(gl_sync()); });
            timed(ewise_ms, [&] { eWiseAdd_->run(matrix_num_rows_, (1 - damping) / matrix_num_rows_); });
        }
        aligned_dense_vec_t result;
        timed(transfer_ms, [&] { result = SpMV_->send_vector_device_to_host(); });
        const float total_ms = float(std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - start).count()) / 1000;
        std::cout << "total_time_ms per iteration: " << total_ms / num_iterations << std::endl;
        std::cout << "spmv_time_ms per iteration: " << spmv_ms / num_iterations << std::endl;
        std::cout << "ewise_time_ms per iteration: " << ewise_ms / num_iterations << std::endl;
        std::cout << "data_transfer_time_ms per iteration: " << transfer_ms / num_iterations << std::endl;
        return result;
    }

    aligned_dense_float_vec_t compute_reference_results(float damping, uint32_t num_iterations) {
        aligned_dense_float_vec_t rank(matrix_num_rows_, 1.0 / matrix_num_rows_);
        for (uint32_t iter = 1; iter <= num_iterations; iter++) {
            rank = SpMV_->compute_reference_results(rank);
            rank = eWiseAdd_->compute_reference_results(rank, matrix_num_rows_, (1 - damping) / matrix_num_rows_);
        }
        return rank;
    }
};

}  // namespace app
}  // namespace graphlily

#endif  // GRAPHLILY_HIP_APP_PAGERANK_H_
#endif  // GRAPHLILY_USE_REFERENCE_APPS
