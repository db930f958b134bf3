// graphlily/module/spmv_module.h -- SpMVModule on MI355X (reference module/spmv_module.h:37-532).
// Same public surface; load_and_format_matrix + send_matrix_host_to_device hand the CSR to
// gl_spmv_plan_create (which builds the CDNA4 layout) and run() is gl_spmv_run.
#ifndef GRAPHLILY_SPMV_MODULE_H_
#define GRAPHLILY_SPMV_MODULE_H_

#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/io/data_formatter.h"
#include "graphlily/io/data_loader.h"
#include "graphlily/module/base_module.h"

using graphlily::io::CSRMatrix;

namespace graphlily {
namespace module {

template <typename matrix_data_t, typename vector_data_t>
class SpMVModule : public BaseModule {
    // one of the reference's three value types (global.h:62-64) for matrix and vector alike: float, unsigned, or the 32-bit
    // fixed point (graphlily::ufixed_32_8); the host matrix always arrives as float and is converted like
    // csr_matrix_convert_from_float<matrix_data_t> does (io/data_loader.h:75-84)
    static_assert(std::is_same<matrix_data_t, vector_data_t>::value && sizeof(vector_data_t) == 4,
                  "matrix and vector share one 32-bit value type");
    typedef graphlily::value_kind<vector_data_t> VK;
    static const bool kFloat = VK::kind == GL_VAL_FLOAT;
    using aligned_dense_vec_t = std::vector<vector_data_t, aligned_allocator<vector_data_t>>;

    MaskType mask_type_ = kNoMask;
    SemiringType semiring_ = ArithmeticSemiring;
    uint32_t num_channels_, out_buf_len_, vec_buf_len_;  // FPGA geometry, accepted as hints
    uint32_t row_begin_ = 0, row_end_ = 0;
    bool sharded_ = false;
    CSRMatrix<float> csr_matrix_float_;
    gl_spmv_plan plan_ = nullptr;
    uint32_t plan_flags_ = 0;
    // The reference keeps host mirrors of its three buffers (:67-70).  Their contents are only ever observable through
    // send_*_device_to_host, i.e. after a download, so uploads go straight from the caller's vector and downloads
    // straight into the vector that is returned.

    // upload `n` floats taken from `src` (zero-padded if it is shorter) into a fresh buffer
    static DeviceBuffer upload_dense_(aligned_dense_vec_t &src, size_t n) {
        DeviceBuffer buf(sizeof(float) * n);
        const size_t have = std::min(src.size(), n);
        if (have) buf.upload(src.data(), sizeof(float) * have);   // (every value type is one 32-bit word; 0 is all bits clear)
        if (have < n) GRAPHLILY_CHECK(gl_buf_fill_f32((float *)buf.ptr() + have, 0.0f, n - have));
        return buf;
    }
    static aligned_dense_vec_t download_dense_(const DeviceBuffer &buf, size_t n) {
        graphlily_detail::NoInitScope no_fill;   // (overwritten by the copy below)
        aligned_dense_vec_t out(n);
        buf.download(out.data(), sizeof(float) * n);
        return out;
    }

    // The semiring known at upload time picks the layout: pattern-only entries and a bit vector for (||,&&),
    // a larger hot-column table for (min,+), 8-byte accumulators for (+,x).
    // GRAPHLILY_SPMV_ORDER=reference (float only; read when the plan is made): the DIAGNOSTIC layout that evaluates the
    // reference's own loop -- CSR order, fp32 multiply and add rounded separately (spmv_module.h:478-510 of the reference) -- and
    // is bit-equal to compute_reference_results for all three semirings.  The default layouts accumulate (+,x) in f64, i.e. they
    // return the correctly rounded row sum, which differs from the reference's sequential fp32 sum by that sum's own rounding
    // error ("float (+,x) vs the reference loop").
    static bool reference_order_() {
        const char *e = getenv("GRAPHLILY_SPMV_ORDER");
        return kFloat && e && std::string(e) == "reference";
    }
    static uint32_t flags_for_(OperationType op) {
        if (reference_order_()) return GL_PLAN_REFERENCE_ORDER;
        if (op == kLogicalAndOr && kFloat) return GL_PLAN_BOOLEAN | GL_PLAN_NO_MULADD;   // (the bit layout serves float only)
        return (op != kMulAdd) ? GL_PLAN_NO_MULADD : 0u;
    }
    bool plan_serves_(OperationType op) const {
        if ((plan_flags_ & GL_PLAN_BOOLEAN) && op != kLogicalAndOr) return false;
        return !((plan_flags_ & GL_PLAN_NO_MULADD) && op == kMulAdd);
    }
    void make_plan_() {
        const CSRMatrix<float> &m = csr_matrix_float_;
        detail::fusion().forget(this);      // (first: a debt that re-runs this module's SpMV needs the plan it ran on)
        gl_spmv_plan_destroy(plan_);
        plan_ = nullptr;
        plan_flags_ = flags_for_(semiring_.op);
        const float *values = m.adj_data.data();
        std::vector<vector_data_t> words;   // the integer value types: the matrix as value words (same size, passed as bits)
        if (!kFloat) {
            words.resize(m.adj_data.size());
            for (size_t i = 0; i < words.size(); i++) words[i] = VK::from_float(m.adj_data[i]);
            values = reinterpret_cast<const float *>(words.data());
        }
        GRAPHLILY_CHECK(gl_spmv_plan_create_ex(&plan_, m.num_rows, m.num_cols, m.adj_indptr.data(), m.adj_indices.data(),
                                               values, sharded_ ? row_begin_ : 0,
                                               sharded_ ? row_end_ : m.num_rows, plan_flags_));
        gl_spmv_plan_desc desc;
        GRAPHLILY_CHECK(gl_spmv_plan_describe(plan_, &desc));
        const int layout = desc.layout;
        const uint32_t segments = desc.segments;
        const bool whole = !sharded_ || (row_begin_ == 0 && row_end_ == m.num_rows);
        fusable_plan_ = kFloat && whole && layout == GL_LAYOUT_BOOLEAN && segments == 1 && m.num_rows == m.num_cols;
        if (whole && kFloat) detail::fusion().announce(this, owner_, plan_, nullptr, m.num_rows, m.num_cols, m.adj_indptr[m.num_rows]);
    }
    bool fusable_plan_ = false;
    void run_now_() {
        GRAPHLILY_CHECK(gl_spmv_run_typed(plan_, vector_buf.rptr(), mask_type_ == kNoMask ? nullptr : mask_buf.rptr(), results_buf.ptr(),
                                          (int)semiring_.op, VK::bits(semiring_.zero), (int)mask_type_, VK::kind));
    }

public:
    // device buffers (the reference's public cl::Buffer members, :83-86)
    std::vector<DeviceBuffer> channel_packets_buf;  // unused: the formatted matrix lives in the plan
    DeviceBuffer vector_buf;
    DeviceBuffer mask_buf;
    DeviceBuffer results_buf;

    SpMVModule(uint32_t num_channels, uint32_t out_buf_len, uint32_t vec_buf_len)
        : BaseModule("overlay"), num_channels_(num_channels), out_buf_len_(out_buf_len), vec_buf_len_(vec_buf_len) {}
    ~SpMVModule() override {
        detail::fusion().forget(this);
        gl_spmv_plan_destroy(plan_);
    }

    void set_semiring(SemiringType semiring) {
        settle_deferred_();     // (a deferred run() reads the semiring when it finally runs: module/fusion.h)
        semiring_ = semiring;
    }
    void set_mask_type(MaskType mask_type) {
        settle_deferred_();
        mask_type_ = mask_type;
    }
    // extension (multi-GPU): this device owns rows [row_begin, row_end)
    void set_row_shard(uint32_t row_begin, uint32_t row_end) {
        row_begin_ = row_begin;
        row_end_ = row_end;
        sharded_ = true;
    }

    gl_spmv_plan plan_handle() const { return plan_; }   // extension: for SpMSpVModule::attach_pull_plan

    // extensions for (||,&&) loops that keep the frontier as a bit vector (gl_spmv_plan_bits_words, gl_pack_bits,
    // gl_spmv_run_bits, gl_bfs_pull_step): bits_words() is 0 unless the plan holds the boolean layout
    uint64_t bits_words() {
        uint64_t w = 0;
        if (plan_ && plan_serves_(semiring_.op)) GRAPHLILY_CHECK(gl_spmv_plan_bits_words(plan_, &w));
        return w;
    }
    static void pack_bits(DeviceBuffer x, uint32_t n, DeviceBuffer bits) {
        GRAPHLILY_CHECK(gl_pack_bits((const float *)x.ptr(), n, (uint32_t *)bits.ptr()));
    }
    void run_bits(DeviceBuffer bits) {
        barrier_();
        GRAPHLILY_CHECK(gl_spmv_run_bits(plan_, (const uint32_t *)bits.ptr(),
                                         mask_type_ == kNoMask ? nullptr : (const float *)mask_buf.ptr(),
                                         (float *)results_buf.ptr(), (float)semiring_.zero, (int)mask_type_));
        finish_();
    }
    // one BFS pull iteration (app/bfs.h:118-123) in one launch; false if the plan is split: use the three calls
    bool bfs_pull_step(DeviceBuffer bits_in, DeviceBuffer bits_out, DeviceBuffer distance, float level) {
        barrier_();
        const int rc = gl_bfs_pull_step(plan_, (const uint32_t *)bits_in.ptr(), (uint32_t *)bits_out.ptr(),
                                        (float *)distance.ptr(), level);
        if (rc == GL_ERR_UNSUPPORTED) return false;
        GRAPHLILY_CHECK(rc);
        finish_();
        return true;
    }
    uint32_t get_num_rows() { return csr_matrix_float_.num_rows; }
    uint32_t get_num_cols() { return csr_matrix_float_.num_cols; }
    uint32_t get_nnz() { return csr_matrix_float_.adj_indptr[csr_matrix_float_.num_rows]; }

    void load_and_format_matrix(CSRMatrix<float> const &csr_matrix_float, bool /*skip_empty_rows*/) {
        csr_matrix_float_ = csr_matrix_float;
    }

    void send_matrix_host_to_device() {
        const CSRMatrix<float> &m = csr_matrix_float_;
        barrier_();
        make_plan_();
        // host blocks for the n-element vectors a driver call builds and returns (two inputs, the result, the previous
        // call's result and the caller's reference result still alive): parked and paged in now, not inside the first timed call
        GRAPHLILY_CHECK(gl_host_pool_reserve(sizeof(float) * (size_t)std::max(m.num_rows, m.num_cols), 6));
        results_buf = DeviceBuffer(sizeof(float) * m.num_rows);
        GRAPHLILY_CHECK(gl_buf_fill_f32((float *)results_buf.ptr(), 0.0f, m.num_rows));
        finish_();
    }

    void send_vector_host_to_device(aligned_dense_vec_t &vector) {
        barrier_();
        vector_buf = upload_dense_(vector, get_num_cols());
    }

    void send_mask_host_to_device(aligned_dense_vec_t &mask) {
        barrier_();
        mask_buf = upload_dense_(mask, get_num_rows());
    }

    void bind_mask_buf(DeviceBuffer src_buf) {
        barrier_();
        mask_buf = src_buf;
    }
    void bind_vector_buf(DeviceBuffer src_buf) {    // extension
        barrier_();
        vector_buf = src_buf;
    }
    void bind_results_buf(DeviceBuffer src_buf) {   // extension
        barrier_();
        results_buf = src_buf;
    }

    // Extension (gl_spmv_plan_chain): the caller feeds every result straight back as the next vector and writes neither in between
    // (PageRank::pull, SSSP::pull: SpMV + eWiseAdd run as one SpMV and a swap, module/fusion.h) -- a run's epilogue then leaves y
    // in the next run's packed form and the next run skips its helper launch.  GRAPHLILY_SPMV_CHAIN=0: never.
    void chain(bool on) {
        static const bool allowed = !(getenv("GRAPHLILY_SPMV_CHAIN") && atoi(getenv("GRAPHLILY_SPMV_CHAIN")) == 0);
        if (plan_ && (allowed || !on)) GRAPHLILY_CHECK(gl_spmv_plan_chain(plan_, on ? 1 : 0, nullptr));
    }

    void run() {
        barrier_();
        if (!plan_serves_(semiring_.op)) {   // semiring switched after upload: re-format
            GRAPHLILY_CHECK(gl_sync());
            make_plan_();
        }
        // the first call of a BFS pull iteration (app/bfs.h:118-123)?  Then it waits for the two that follow (module/fusion.h)
        detail::PullFusion &F = detail::fusion();
        if (!blocking_ && fusable_plan_ && F.enabled() && semiring_.op == kLogicalAndOr && VK::bits(semiring_.zero) == 0u &&
            mask_type_ == kMaskWriteToZero && vector_buf.valid() && mask_buf.valid() && results_buf.valid()) {
            if (gl_spmspv_plan csc = F.partner_of(this)) {
                F.defer_spmv(this, plan_, csc, get_num_rows(), vector_buf, mask_buf, results_buf, [this] { run_now_(); });
                return;
            }
        }
        // the first call of an SSSP / PageRank pull iteration (SpMV, then eWiseAdd(n, val) results -> vector)?  It waits for the
        // second one (module/fusion.h 3.)
        // (not on the reference-order layout: folding eWiseAdd's value into the SpMV starts the sequential float sum from it,
        //  which rounds differently from sum-then-add -- that layout exists to be bit-equal to the reference)
        if (!blocking_ && kFloat && F.enabled() && mask_type_ == kNoMask && !sharded_ && get_num_rows() == get_num_cols() &&
            !(plan_flags_ & (GL_PLAN_BOOLEAN | GL_PLAN_REFERENCE_ORDER)) && vector_buf.valid() && results_buf.valid() && vector_buf.id() != results_buf.id()) {
            F.defer_copy_spmv(this, get_num_rows(), vector_buf, results_buf, VK::bits(semiring_.zero) == 0u, semiring_.op == kMulAdd,
                              [this](float extra, bool fold) { run_now_plus_(extra, fold); },
                              [this](const DeviceBuffer &x, const DeviceBuffer &y) {
                                  GRAPHLILY_CHECK(gl_spmv_run_typed(plan_, x.raw(), nullptr, y.raw(), (int)semiring_.op, VK::bits(semiring_.zero),
                                                                    (int)kNoMask, VK::kind));
                              });
            return;
        }
        run_now_();
        finish_();
    }
    // the SpMV with `extra` as the semiring's zero (only when the module's own zero is 0: zero + sum + extra == extra + sum)
    void run_now_plus_(float extra, bool fold) {
        if (!fold) {
            run_now_();
            return;
        }
        uint32_t zb;
        memcpy(&zb, &extra, 4);
        GRAPHLILY_CHECK(gl_spmv_run_typed(plan_, vector_buf.rptr(), nullptr, results_buf.ptr(), (int)semiring_.op, zb, (int)kNoMask, VK::kind));
    }

    // (a download of a buffer that a deferred / fused call still owes settles the debt first: DeviceBuffer::ptr)
    aligned_dense_vec_t send_vector_device_to_host() { return download_dense_(vector_buf, get_num_cols()); }
    aligned_dense_vec_t send_mask_device_to_host() {
        barrier_();
        return download_dense_(mask_buf, get_num_rows());
    }
    aligned_dense_vec_t send_results_device_to_host() { return download_dense_(results_buf, get_num_rows()); }

    // CPU reference the callers verify against (part of the reference's public API, :478-532).
    // Sequential row loop with a float accumulator, like the reference.
    graphlily::aligned_dense_float_vec_t compute_reference_results(graphlily::aligned_dense_float_vec_t &vector) {
        const CSRMatrix<float> &m = csr_matrix_float_;
        graphlily::aligned_dense_float_vec_t y(m.num_rows, (float)semiring_.zero);
        for (uint32_t r = 0; r < m.num_rows; r++) {
            float acc = (float)semiring_.zero;
            for (uint32_t i = m.adj_indptr[r]; i < m.adj_indptr[r + 1]; i++) {
                const float a = m.adj_data[i], b = vector[m.adj_indices[i]];
                switch (semiring_.op) {
                    case kMulAdd: acc += a * b; break;
                    case kLogicalAndOr: acc = acc || (a && b); break;
                    case kAddMin: acc = std::min(acc, a + b); break;
                    default: std::cerr << "Invalid semiring" << std::endl; break;
                }
            }
            y[r] = acc;
        }
        return y;
    }

    graphlily::aligned_dense_float_vec_t compute_reference_results(graphlily::aligned_dense_float_vec_t &vector,
                                                                    graphlily::aligned_dense_float_vec_t &mask) {
        graphlily::aligned_dense_float_vec_t y = compute_reference_results(vector);
        const bool keep_zero = (mask_type_ == kMaskWriteToZero);   // every other type behaves as WriteToOne
        for (size_t i = 0; i < y.size(); i++)
            if ((mask[i] == 0) != keep_zero) y[i] = 0;
        return y;
    }
};

}  // namespace module
}  // namespace graphlily

#endif  // GRAPHLILY_SPMV_MODULE_H_
