// graphlily/app/sssp.h -- SSSP over the MI355X backend with the reference's class (graphlily/app/sssp.h:70-254 of the
// reference: same name, constructor, public methods), so that benchmark/bench_sssp.cpp and tests/test_app.cpp compile
// unmodified with -I<this repo>/include in front.  What differs from the reference's bodies is where the vectors live:
//
//   reference   every call builds its n-element start vectors on the host and uploads them (sssp.h:146-149, :204-212), and
//               pull_push moves the distances device -> host -> device at the push -> pull switch (:224-226): 12 MB each way
//               on orkut, around 2 ms of SpMV;
//   here        the start vectors are FILLED on the device, the switch hands the distance BUFFER from the SpMSpV module to the
//               SpMV module, the push loop's condition reads the SpMSpV's own completion record (no copy, no stream wait:
//               SpMSpVModule::get_results_nnz), and the pull loop's SpMV + eWiseAdd(+0) pairs run as one SpMV and a buffer
//               swap (module/fusion.h).  The only host traffic of a call is the result.
//
// Results are bit-equal to the reference's loops: the same module calls in the same order on the same values.
// -DGRAPHLILY_USE_REFERENCE_APPS: the next graphlily/app/sssp.h on the include path (the reference checkout's) is used instead.
#if defined(GRAPHLILY_USE_REFERENCE_APPS)
#include_next "graphlily/app/sssp.h"
#else
#ifndef GRAPHLILY_HIP_APP_SSSP_H_
#define GRAPHLILY_HIP_APP_SSSP_H_
#define GRAPHLILY_APP_SSSP_H_   // (the reference's guard)

#include "graphlily/app/module_collection.h"
#include "graphlily/module/spmv_module.h"
#include "graphlily/module/spmspv_module.h"
#include "graphlily/module/assign_vector_dense_module.h"
#include "graphlily/module/assign_vector_sparse_module.h"
#include "graphlily/module/add_scalar_vector_dense_module.h"
#include "graphlily/io/data_loader.h"
#include "graphlily/io/data_formatter.h"

#include <iostream>

namespace graphlily {
namespace app {
namespace detail {

// The matrix preparation of the reference's SSSP (sssp.h:16-62): every weight becomes 1 and a weight-0 self edge is added
// per row so that a (min,+) SpMV keeps the previous distance.  The reference edits the CSR arrays in place while it walks the
// rows (O(rows x nnz) vector inserts) and reads every row's END from the not-yet-shifted row pointer: after k insertions only
// the first (length - k) entries of a row are looked at, a row whose window is empty gets its self edge in front, a row with
// a negative window gets none, and an edge inserted "at the end" goes BEFORE the last entry looked at.  Results must equal the
// reference's, so exactly that is reproduced -- in one O(nnz) pass: decide per row where (whether) the edge goes, then write
// the new arrays once.
inline void sssp_preprocess(CSRMatrix<float> &m) {
    const uint32_t n = (uint32_t)m.adj_indptr.size() - 1;
    const std::vector<uint32_t> old_ptr(m.adj_indptr.begin(), m.adj_indptr.end());
    const std::vector<uint32_t> old_idx(m.adj_indices.begin(), m.adj_indices.end());
    std::vector<int64_t> insert_at(n, -1), zero_at(n, -1);   // position inside the row; -1: none
    int64_t k = 0;                                            // self edges inserted so far
    for (uint32_t r = 0; r < n; r++) {
        const int64_t len = (int64_t)old_ptr[r + 1] - old_ptr[r], win = len - k;
        if (win == 0) {
            insert_at[r] = 0;
            k++;
        } else if (win > 0) {
            const uint32_t *row = old_idx.data() + old_ptr[r];
            int64_t hit = -1;                                 // first entry of the window with column >= r
            for (int64_t i = 0; i < win; i++)
                if (row[i] >= r) {
                    hit = i;
                    break;
                }
            if (hit >= 0 && row[hit] == r) {
                zero_at[r] = hit;                             // the diagonal entry exists: its weight becomes 0
            } else {
                insert_at[r] = hit >= 0 ? hit : win - 1;
                k++;
            }
        }
    }
    const size_t total = (size_t)old_ptr[n] + (size_t)k;
    m.adj_indices.assign(total, 0u);
    m.adj_data.assign(total, 1.0f);
    uint32_t out = 0;
    for (uint32_t r = 0; r < n; r++) {
        m.adj_indptr[r] = out;
        const uint32_t len = old_ptr[r + 1] - old_ptr[r];
        const uint32_t *row = old_idx.data() + old_ptr[r];
        for (uint32_t i = 0; i <= len; i++) {
            if ((int64_t)i == insert_at[r]) {
                m.adj_indices[out] = r;
                m.adj_data[out++] = 0.0f;
            }
            if (i == len) break;
            m.adj_indices[out] = row[i];
            m.adj_data[out++] = ((int64_t)i == zero_at[r]) ? 0.0f : 1.0f;
        }
    }
    m.adj_indptr[n] = out;
}

}  // namespace detail

class SSSP : public app::ModuleCollection {
private:
    module::SpMVModule<graphlily::val_t, graphlily::val_t> *SpMV_;
    module::SpMSpVModule<graphlily::val_t, graphlily::val_t, graphlily::idx_val_t> *SpMSpV_;
    module::AssignVectorSparseModule<graphlily::val_t, graphlily::idx_val_t> *SparseAssign_;
    module::eWiseAddModule<graphlily::val_t> *eWiseAdd_;
    uint32_t matrix_num_rows_ = 0, matrix_num_cols_ = 0;
    uint32_t num_channels_, spmv_out_buf_len_, spmspv_out_buf_len_, vec_buf_len_;
    graphlily::SemiringType semiring_ = graphlily::TropicalSemiring;
    using aligned_dense_vec_t = graphlily::aligned_dense_vec_t;
    using aligned_sparse_vec_t = graphlily::aligned_sparse_vec_t;
    using aligned_dense_float_vec_t = graphlily::aligned_dense_float_vec_t;
    typedef graphlily::value_kind<graphlily::val_t> VK;

    // an n-element vector that holds `fill` everywhere and `at_source` at the source, made on the device
    // (a fresh DeviceBuffer per call is a block of the library's device POOL -- gl_buf_alloc hands back the block the previous call
    //  released, GRAPHLILY_POOL_TRACE shows it -- not a hipMalloc / hipFree pair: ADVICE r05 read it as one.  The fusion swap
    //  exchanges storage between the modules' vector and results buffers, so a buffer kept here would have to be re-bound per call
    //  anyway.)
    DeviceBuffer device_dense_(uint32_t source, graphlily::val_t fill, graphlily::val_t at_source) {
        DeviceBuffer b(sizeof(graphlily::val_t) * (size_t)matrix_num_rows_);
        GRAPHLILY_CHECK(gl_buf_fill_u32((uint32_t *)b.ptr(), VK::bits(fill), matrix_num_rows_));
        GRAPHLILY_CHECK(gl_buf_fill_u32((uint32_t *)b.ptr() + source, VK::bits(at_source), 1));
        return b;
    }
    void start_push_(uint32_t source) {   // sssp.h:171-190 / :198-216
        aligned_sparse_vec_t frontier(2);
        idx_val_t head;
        head.index = 1;   // one source vertex
        head.val = 0;
        frontier[0] = head;
        frontier[1] = {source, 0};
        SpMSpV_->send_vector_host_to_device(frontier);
        SpMSpV_->bind_mask_buf(device_dense_(source, semiring_.zero, 0));   // the distances: zero (= infinity) except at the source
        SparseAssign_->bind_mask_buf(SpMSpV_->results_buf);
        SparseAssign_->bind_inout_buf(SpMSpV_->mask_buf);
        SparseAssign_->bind_new_frontier_buf(SpMSpV_->vector_buf);
    }
    void pull_loop_(uint32_t iter, uint32_t num_iterations) {
        eWiseAdd_->bind_in_buf(SpMV_->results_buf);
        eWiseAdd_->bind_out_buf(SpMV_->vector_buf);
        SpMV_->chain(true);   // (as in PageRank::pull; the callers read the vector back and end the chain)
        for (; iter <= num_iterations; iter++) {
            SpMV_->run();
            eWiseAdd_->run(matrix_num_rows_, 0);   // results -> vector (the pair runs as one SpMV + a swap: module/fusion.h)
        }
    }

public:
    SSSP(uint32_t num_channels, uint32_t spmv_out_buf_len, uint32_t spmspv_out_buf_len, uint32_t vec_buf_len)
        : num_channels_(num_channels), spmv_out_buf_len_(spmv_out_buf_len), spmspv_out_buf_len_(spmspv_out_buf_len), vec_buf_len_(vec_buf_len) {
        SpMV_ = new module::SpMVModule<graphlily::val_t, graphlily::val_t>(num_channels_, spmv_out_buf_len_, vec_buf_len_);
        SpMV_->set_semiring(semiring_);
        SpMV_->set_mask_type(graphlily::kNoMask);
        add_module(SpMV_);
        SpMSpV_ = new module::SpMSpVModule<graphlily::val_t, graphlily::val_t, graphlily::idx_val_t>(spmspv_out_buf_len_);
        SpMSpV_->set_semiring(semiring_);
        SpMSpV_->set_mask_type(graphlily::kNoMask);
        add_module(SpMSpV_);
        SparseAssign_ = new module::AssignVectorSparseModule<graphlily::val_t, graphlily::idx_val_t>(true);
        add_module(SparseAssign_);
        eWiseAdd_ = new module::eWiseAddModule<graphlily::val_t>();
        add_module(eWiseAdd_);
    }

    uint32_t get_nnz() { return SpMV_->get_nnz(); }

    void load_and_format_matrix(std::string csr_float_npz_path, bool skip_empty_rows) {
        CSRMatrix<float> csr_matrix = graphlily::io::load_csr_matrix_from_float_npz(csr_float_npz_path);
        detail::sssp_preprocess(csr_matrix);
        graphlily::io::util_round_csr_matrix_dim(csr_matrix, num_channels_ * graphlily::pack_size, num_channels_ * graphlily::pack_size);
        CSCMatrix<float> csc_matrix = graphlily::io::csr2csc(csr_matrix);
        SpMV_->load_and_format_matrix(csr_matrix, skip_empty_rows);
        SpMSpV_->load_and_format_matrix(csc_matrix);
        matrix_num_rows_ = SpMV_->get_num_rows();
        matrix_num_cols_ = SpMV_->get_num_cols();
        assert(matrix_num_rows_ == matrix_num_cols_);
    }

    void send_matrix_host_to_device() {
        SpMV_->send_matrix_host_to_device();
        SpMSpV_->send_matrix_host_to_device();
    }

    aligned_dense_vec_t pull(uint32_t source, uint32_t num_iterations) {
        SpMV_->bind_vector_buf(device_dense_(source, semiring_.zero, 0));
        pull_loop_(1, num_iterations);
        aligned_dense_vec_t distance = SpMV_->send_vector_device_to_host();
        SpMV_->chain(false);
        return distance;
    }

    aligned_dense_vec_t push(uint32_t source, uint32_t num_iterations) {
        start_push_(source);
        for (uint32_t iter = 1; iter <= num_iterations; iter++) {
            SpMSpV_->run();          // candidate distances of the frontier's neighbours
            SparseAssign_->run();    // relax: distance = min, the improved vertices are the next frontier
        }
        return SpMSpV_->send_mask_device_to_host();
    }

    aligned_dense_vec_t pull_push(uint32_t source, uint32_t num_iterations, float threshold = 0.05) {
        start_push_(source);
        uint32_t iter = 1, vector_nnz;
        do {
            SpMSpV_->run();
            SparseAssign_->run();
            vector_nnz = SpMSpV_->get_results_nnz();   // (the run's own completion record: the relax step is already enqueued)
            iter++;
        } while (iter < num_iterations && (float(vector_nnz) / matrix_num_rows_ < threshold));
        std::cout << "SpMSpV runs for " << (iter - 1) << " iterations" << std::endl;
        // push -> pull (sssp.h:224-226 moves the distances through the host): the SpMV takes the distance buffer as its vector
        SpMV_->bind_vector_buf(SpMSpV_->mask_buf);
        pull_loop_(iter, num_iterations);
        aligned_dense_vec_t distance = SpMV_->send_vector_device_to_host();
        SpMV_->chain(false);
        return distance;
    }

    aligned_dense_float_vec_t compute_reference_results(uint32_t source, uint32_t num_iterations) {
        aligned_dense_float_vec_t input(matrix_num_rows_, semiring_.zero);
        input[source] = 0;
        for (uint32_t iter = 1; iter <= num_iterations; iter++) input = SpMV_->compute_reference_results(input);
        return input;
    }
};

}  // namespace app
}  // namespace graphlily

#endif  // GRAPHLILY_HIP_APP_SSSP_H_
#endif  // GRAPHLILY_USE_REFERENCE_APPS
