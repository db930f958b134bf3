// graphlily/module/spmspv_module.h -- SpMSpVModule on MI355X (reference module/spmspv_module.h:27-520).
#ifndef GRAPHLILY_SPMSPV_MODULE_H_
#define GRAPHLILY_SPMSPV_MODULE_H_

#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/io/data_formatter.h"
#include "graphlily/io/data_loader.h"
#include "graphlily/module/base_module.h"

using graphlily::io::CSCMatrix;

namespace graphlily {
namespace module {

template <typename matrix_data_t, typename vector_data_t, typename idx_val_t>
class SpMSpVModule : public BaseModule {
    static_assert(std::is_same<matrix_data_t, vector_data_t>::value && sizeof(vector_data_t) == 4,
                  "matrix and vector share one 32-bit value type (float, unsigned or graphlily::ufixed_32_8)");
    static_assert(sizeof(idx_val_t) == sizeof(gl_idx_val), "sparse element must be {uint32 index; 32-bit value}");
    typedef graphlily::value_kind<vector_data_t> VK;
    static const bool kFloat = VK::kind == GL_VAL_FLOAT;
    using aligned_dense_vec_t = std::vector<vector_data_t, aligned_allocator<vector_data_t>>;
    using aligned_sparse_vec_t = std::vector<idx_val_t, aligned_allocator<idx_val_t>>;

    MaskType mask_type_ = kNoMask;
    SemiringType semiring_ = ArithmeticSemiring;
    uint32_t out_buf_len_;
    uint32_t row_begin_ = 0, row_end_ = 0;
    bool sharded_ = false;
    CSCMatrix<float> csc_matrix_float_;
    gl_spmspv_plan plan_ = nullptr;
    // sparse vectors come back at their full capacity like the reference's mirrors (:53-60), zero beyond the entries
    // the head counts; only the head and those entries are transferred
    static aligned_sparse_vec_t download_sparse_(const DeviceBuffer &buf) {
        const size_t slots = buf.size() / sizeof(idx_val_t);
        graphlily_detail::NoInitScope no_fill;   // (overwritten by the copy below)
        aligned_sparse_vec_t out(slots);
        if (!slots) return out;
        uint32_t nnz = 0;
        GRAPHLILY_CHECK(gl_sparse_nnz((const gl_idx_val *)buf.ptr(), &nnz));
        const size_t used = std::min(slots, (size_t)nnz + 1);
        buf.download(out.data(), sizeof(idx_val_t) * used);
        // beyond the entries the head counts the reference's mirror holds whatever earlier runs left there; here: zeros for
        // vectors of moderate size, nothing defined for the multi-megabyte ones (filling 24 MB costs more than the download)
        if (slots - used <= (1u << 16)) std::fill(out.begin() + used, out.end(), idx_val_t{0, 0});
        return out;
    }

public:
    DeviceBuffer channel_packets_buf, channel_indptr_buf, channel_partptr_buf;  // unused: held by the plan
    DeviceBuffer vector_buf;
    DeviceBuffer mask_buf;
    DeviceBuffer results_buf;
    DeviceBuffer results_nnz_buf;  // unused: gl_sparse_nnz reads the head of results_buf directly

    explicit SpMSpVModule(uint32_t out_buf_len) : BaseModule("overlay"), out_buf_len_(out_buf_len) {}
    ~SpMSpVModule() override {
        detail::fusion().forget(this);
        gl_spmspv_plan_destroy(plan_);
    }

    uint32_t get_num_rows() { return csc_matrix_float_.num_rows; }
    uint32_t get_num_cols() { return csc_matrix_float_.num_cols; }
    uint32_t get_nnz() { return csc_matrix_float_.adj_indptr[csc_matrix_float_.num_cols]; }

    void set_semiring(SemiringType semiring) {
        settle_deferred_();
        semiring_ = semiring;
    }
    void set_mask_type(MaskType mask_type) {
        settle_deferred_();
        mask_type_ = mask_type;
    }
    void set_row_shard(uint32_t row_begin, uint32_t row_end) {
        row_begin_ = row_begin;
        row_end_ = row_end;
        sharded_ = true;
    }
    gl_spmspv_plan plan_handle() const { return plan_; }   // extension: the device-resident schedules of graphlily/app/*.h

    void load_and_format_matrix(CSCMatrix<float> const &csc_matrix_float) { csc_matrix_float_ = csc_matrix_float; }

    void send_matrix_host_to_device() {
        const CSCMatrix<float> &m = csc_matrix_float_;
        detail::fusion().forget(this);
        gl_spmspv_plan_destroy(plan_);
        plan_ = nullptr;
        const float *values = m.adj_data.data();
        std::vector<vector_data_t> words;   // the integer value types: csc_matrix_convert_from_float (io/data_loader.h:86-90)
        if (!kFloat) {
            words.resize(m.adj_data.size());
            for (size_t i = 0; i < words.size(); i++) words[i] = VK::from_float(m.adj_data[i]);
            values = reinterpret_cast<const float *>(words.data());
        }
        GRAPHLILY_CHECK(gl_spmspv_plan_create(&plan_, m.num_rows, m.num_cols, m.adj_indptr.data(), m.adj_indices.data(),
                                              values, sharded_ ? row_begin_ : 0, sharded_ ? row_end_ : m.num_rows));
        if (kFloat && (!sharded_ || (row_begin_ == 0 && row_end_ == m.num_rows)))
            detail::fusion().announce(this, owner_, nullptr, plan_, m.num_rows, m.num_cols, m.adj_indptr[m.num_cols]);
        GRAPHLILY_CHECK(gl_host_pool_reserve(sizeof(idx_val_t) * ((size_t)std::max(m.num_rows, m.num_cols) + 1), 2));
        GRAPHLILY_CHECK(gl_host_pool_reserve(sizeof(float) * (size_t)std::max(m.num_rows, m.num_cols), 6));
        results_buf = DeviceBuffer(sizeof(idx_val_t) * ((size_t)m.num_rows + 1));
        const gl_idx_val head{0u, 0.0f};   // an empty result list until the first run
        results_buf.upload(&head, sizeof(head));
    }

    // the vector may be shorter than num_cols + 1; the device copy always has that many slots
    void send_vector_host_to_device(aligned_sparse_vec_t &vector) {
        barrier_();
        const size_t slots = (size_t)get_num_cols() + 1;   // reference :280,379
        vector_buf = DeviceBuffer(sizeof(idx_val_t) * slots);
        if (vector.empty()) {
            const gl_idx_val head{0u, 0.0f};
            vector_buf.upload(&head, sizeof(head));
            hint_vector_nnz(0);
            return;
        }
        // only the head and the entries it counts are meaningful
        const size_t used = std::min(std::min(vector.size(), slots), (size_t)vector[0].index + 1);
        vector_buf.upload(vector.data(), sizeof(idx_val_t) * used);
        if ((size_t)vector[0].index + 1 > used) {
            // the head claims more entries than the vector holds (the reference would read its zero-initialised mirror):
            // the device block is recycled memory, so the head is clamped to what was uploaded
            idx_val_t head = vector[0];
            head.index = (idx_t)(used - 1);
            vector_buf.upload(&head, sizeof(head));
        }
        hint_vector_nnz((uint32_t)(used - 1));
        // the host holds the CSC: the vector's entry count, the non-zeros of its columns and the longest of them are known
        // here (gl_spmspv_plan_hint_work: a tiny vector runs as one launch instead of four, one below the direction
        // switch's threshold without the decision kernels; never result-relevant).  Valid for the runs that follow while
        // no other module call comes in between (the reference's drivers overwrite vector_buf through
        // copy_buffer_device_to_device, app/bfs.h:149).
        hint_cnt_ = (uint32_t)(used - 1);
        hint_work_ = 0;
        hint_longest_ = 0;
        for (uint32_t k = 1; k <= hint_cnt_; k++) {
            const uint32_t c = vector[k].index;
            if (c >= csc_matrix_float_.num_cols) continue;
            const uint32_t len = csc_matrix_float_.adj_indptr[c + 1] - csc_matrix_float_.adj_indptr[c];
            hint_work_ += len;
            hint_longest_ = std::max(hint_longest_, len);
        }
        hint_stamp_ = hint_cnt_ ? calls_() : 0;
    }

private:
    uint32_t hint_cnt_ = 0, hint_longest_ = 0;
    uint64_t hint_work_ = 0, hint_stamp_ = 0;
    // called right after barrier_(): is this the first module call since the stamp?
    bool hint_fresh_() const { return plan_ && kFloat && hint_stamp_ != 0 && hint_stamp_ + 1 == calls_(); }

public:

    void send_mask_host_to_device(aligned_dense_vec_t &mask) {
        barrier_();
        mask_buf = DeviceBuffer(sizeof(float) * mask.size());
        mask_buf.upload(mask.data(), sizeof(float) * mask.size());
    }

    void bind_mask_buf(DeviceBuffer src_buf) {                            // extension
        settle_deferred_();
        mask_buf = src_buf;
    }
    void bind_vector_buf(DeviceBuffer src_buf) {                          // extension
        settle_deferred_();
        vector_buf = src_buf;
        hint_stamp_ = 0;
    }

    // extensions (gl_spmspv_plan_attach_pull / gl_spmspv_plan_hint): a driver that also holds the matrix as a
    // (||,&&) SpMVModule lets heavy frontiers run row-wise; pass SpMVModule::plan_handle() after BOTH modules
    // have sent their matrices, and again if the SpMV module re-formats (semiring change).  nullptr detaches.
    void attach_pull_plan(gl_spmv_plan spmv_plan) { GRAPHLILY_CHECK(gl_spmspv_plan_attach_pull(plan_, spmv_plan)); }
    void hint_vector_nnz(uint32_t nnz) {
        if (plan_) GRAPHLILY_CHECK(gl_spmspv_plan_hint(plan_, nnz));
    }

    void run() {
        barrier_();
        const bool fresh = hint_fresh_();
        if (fresh) GRAPHLILY_CHECK(gl_spmspv_plan_hint_work(plan_, hint_cnt_, hint_work_, hint_longest_));
        hint_stamp_ = fresh ? calls_() : 0;      // (a run reads the vector: the knowledge stays good)
        GRAPHLILY_CHECK(gl_spmspv_run_typed(plan_, vector_buf.ptr(), mask_type_ == kNoMask ? nullptr : mask_buf.ptr(), results_buf.ptr(),
                                            (int)semiring_.op, VK::bits(semiring_.zero), (int)mask_type_, VK::kind));
        finish_run_();
    }

private:
    // The result count of the last run, if that run reported it (gl_spmspv_wait: the operator's last workgroup stores it to
    // page-locked memory) and nothing has been enqueued by any module since.
    uint32_t known_nnz_ = 0;
    uint64_t known_stamp_ = 0;
    // a blocking run waits for the operator's own completion record instead of the whole stream
    // results_buf's touch stamp right after the last run(): while it stands, the run's completion record describes the buffer
    uint64_t run_touch_ = 0;
    const void *run_results_ = nullptr;
    void finish_run_() {
        known_stamp_ = 0;
        run_touch_ = results_buf.touched();
        run_results_ = results_buf.id();
        if (!blocking_) return;
        uint32_t nnz = 0xffffffffu;
        GRAPHLILY_CHECK(gl_spmspv_wait(plan_, &nnz));
        if (nnz != 0xffffffffu) {
            known_nnz_ = nnz;
            known_stamp_ = calls_();
        }
    }

public:
    // extension (gl_spmspv_run_assign): run() + AssignVectorSparseModule::run(val) with the results as its mask and
    // `inout` as its inout (the push iteration of app/bfs.h:146-148) in one call
    void run_assign(DeviceBuffer inout, vector_data_t val) {
        barrier_();
        if (!kFloat) {   // the fused call exists for float; the integer value types run the two steps
            run();
            GRAPHLILY_CHECK(gl_assign_sparse_typed(results_buf.ptr(), inout.ptr(), VK::bits(val), (uint32_t)(results_buf.size() / sizeof(gl_idx_val) - 1)));
            finish_();
            return;
        }
        GRAPHLILY_CHECK(gl_spmspv_run_assign(plan_, (const gl_idx_val *)vector_buf.ptr(),
                                             mask_type_ == kNoMask ? nullptr : (const float *)mask_buf.ptr(),
                                             (gl_idx_val *)results_buf.ptr(), (int)semiring_.op, (float)semiring_.zero,
                                             (int)mask_type_, (float *)inout.ptr(), (float)val));
        finish_run_();
    }

    aligned_sparse_vec_t send_vector_device_to_host() { return download_sparse_(vector_buf); }
    aligned_dense_vec_t send_mask_device_to_host() {
        barrier_();
        graphlily_detail::NoInitScope no_fill;
        aligned_dense_vec_t out(mask_buf.size() / sizeof(float));
        mask_buf.download(out.data(), sizeof(float) * out.size());
        return out;
    }
    aligned_sparse_vec_t send_results_device_to_host() { return download_sparse_(results_buf); }

    // the per-iteration device->host control read of the push loops (reference :239-242)
    uint32_t get_results_nnz() {
        barrier_();
        if (hint_fresh_()) hint_stamp_ = calls_();   // (reads only)
        if (known_stamp_ != 0 && known_stamp_ + 1 == calls_()) {   // the run itself reported the count: no copy
            known_stamp_ = calls_();
            return known_nnz_;
        }
        uint32_t nnz = 0xffffffffu;
        // the run's own record, if it kept one -- trusted while nobody has accessed results_buf for writing since that run
        // (SSSP's relax step, enqueued behind the run, only reads it as its mask: app/sssp.h:218-221); handed out once
        if (!blocking_ && plan_ && run_touch_ != 0 && run_results_ == results_buf.id() && results_buf.touched() == run_touch_)
            GRAPHLILY_CHECK(gl_spmspv_wait(plan_, &nnz));
        run_touch_ = 0;
        if (nnz == 0xffffffffu) GRAPHLILY_CHECK(gl_sparse_nnz((const gl_idx_val *)results_buf.ptr(), &nnz));
        return nnz;
    }

    // CPU reference (reference :445-520): dense result, (min,+) product saturating at FLOAT_INF, mask
    // compared with semiring.zero and masked-off rows set to semiring.zero.
    graphlily::aligned_dense_float_vec_t compute_reference_results(graphlily::aligned_sparse_float_vec_t &vector,
                                                                    graphlily::aligned_dense_float_vec_t &mask) {
        const CSCMatrix<float> &m = csc_matrix_float_;
        const float zero = (float)semiring_.zero;
        graphlily::aligned_dense_float_vec_t y(m.num_rows, zero);
        const uint32_t active = vector[0].index;
        for (uint32_t k = 1; k <= active; k++) {
            const float xv = vector[k].val;
            const uint32_t col = vector[k].index;
            for (uint32_t e = m.adj_indptr[col]; e < m.adj_indptr[col + 1]; e++) {
                float &acc = y[m.adj_indices[e]];
                const float a = m.adj_data[e];
                if (semiring_.op == kMulAdd) {
                    acc += a * xv;
                } else if (semiring_.op == kLogicalAndOr) {
                    acc = acc || (a && xv);
                } else if (semiring_.op == kAddMin) {
                    const float finf = (float)FLOAT_INF;
                    float t = (a > finf || xv > finf) ? finf : a + xv;
                    if (t > finf) t = finf;
                    acc = (acc < t) ? acc : t;
                } else {
                    std::cerr << "ERROR: [Module SpMSpV] Invalid semiring" << std::endl;
                }
            }
        }
        for (uint32_t i = 0; i < m.num_rows; i++) {
            bool off = true;
            if (mask_type_ == kNoMask) off = false;
            else if (mask_type_ == kMaskWriteToOne) off = (mask[i] == zero);
            else if (mask_type_ == kMaskWriteToZero) off = (mask[i] != zero);
            if (off) y[i] = zero;
        }
        return y;
    }
};

}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_SPMSPV_MODULE_H_
