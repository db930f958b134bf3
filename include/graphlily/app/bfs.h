// graphlily/app/bfs.h -- BFS over the MI355X backend, with the reference's class (graphlily/app/bfs.h:20-361 of the
// reference: same name, constructor, public methods and return types), so that a driver written against the reference --
// benchmark/bench_bfs.cpp, tests/test_app.cpp -- compiles unmodified with -I<this repo>/include in front and runs the
// DEVICE-RESIDENT schedule instead of the reference's host-driven loop:
//
//   reference   every push iteration reads the result count back to decide the direction (bfs.h:180-190), the switch converts
//               the frontier on the host (:195-205): two uploads, one download and a device round trip per iteration;
//   here        pull() / pull_push() enqueue the WHOLE run up front -- gl_bfs_bits_begin, one gl_bfs_bits_shard_step per
//               iteration slot, gl_bfs_bits_shard_finish (include/graphlily_hip.h) -- with the frontier as bits only and the
//               reference's loop condition replayed on the device from exact counts (same float comparison); the sequence
//               is recorded as a hipGraph on the first call and replayed afterwards; the levels come back packed (nibbles /
//               bytes) when that pays.  Results are bit-equal to the reference's loops (levels do not depend on direction).
//
// -DGRAPHLILY_USE_REFERENCE_APPS: this header steps aside and the next graphlily/app/bfs.h on the include path (the reference
// checkout's) is used over the module layer, as in rounds 1-4 (oracle/Makefile ref_apps).
// GRAPHLILY_BFS_HOST_LOOP=1 (environment) or a value type other than float: the reference's module-call sequence, written out
// below (push(), and the fallback of pull() / pull_push()).
//
// Extension for row-sharded runs: set_comm(gl_dist) before load_and_format_matrix -- the rank then owns an
// nnz-balanced range of rows cut on multiples of 64, every slot exchanges its bit vector + tallies with ONE
// gl_dist_all_gather_bits_tally (recorded inside the hipGraph), and the result is all-gathered at the end.
#if defined(GRAPHLILY_USE_REFERENCE_APPS)
#include_next "graphlily/app/bfs.h"
#else
#ifndef GRAPHLILY_HIP_APP_BFS_H_
#define GRAPHLILY_HIP_APP_BFS_H_
#define GRAPHLILY_APP_BFS_H_   // (the reference's guard: a later include of its header is a no-op)

#include "graphlily/app/module_collection.h"
#include "graphlily/app/row_shards.h"
#include "graphlily/module/spmv_module.h"
#include "graphlily/module/spmspv_module.h"
#include "graphlily/module/assign_vector_dense_module.h"
#include "graphlily/module/assign_vector_sparse_module.h"
#include "graphlily/module/add_scalar_vector_dense_module.h"
#include "graphlily/io/data_loader.h"
#include "graphlily/io/data_formatter.h"

#include <chrono>
#include <iostream>
#include <map>
#include <tuple>

namespace graphlily {
namespace app {

class BFS : public app::ModuleCollection {
private:
    module::SpMVModule<graphlily::val_t, graphlily::val_t> *SpMV_;
    module::AssignVectorDenseModule<graphlily::val_t> *DenseAssign_;
    module::SpMSpVModule<graphlily::val_t, graphlily::val_t, graphlily::idx_val_t> *SpMSpV_;
    module::AssignVectorSparseModule<graphlily::val_t, graphlily::idx_val_t> *SparseAssign_;
    module::eWiseAddModule<graphlily::val_t> *eWiseAdd_;
    uint32_t matrix_num_rows_ = 0, matrix_num_cols_ = 0;
    uint32_t num_channels_, spmv_out_buf_len_, spmspv_out_buf_len_, vec_buf_len_;
    graphlily::SemiringType semiring_ = graphlily::LogicalSemiring;
    using aligned_dense_vec_t = graphlily::aligned_dense_vec_t;
    using aligned_sparse_vec_t = graphlily::aligned_sparse_vec_t;
    using aligned_dense_float_vec_t = graphlily::aligned_dense_float_vec_t;
    static constexpr bool kFloat = std::is_same<graphlily::val_t, float>::value;

    // ---- the device-resident schedule's state (per matrix; rebuilt when a longer run asks for more slots)
    struct Schedule {
        uint32_t slots = 0, words = 0, nvec = 0, nvec_all = 0, ctl_words = 0;
        DeviceBuffer both;      // n distances, then the control words: one read-back fetches both
        DeviceBuffer vecs;      // nvec bit vectors of `words` words, then the ranks' tallies
        // the levels as nibbles ([0]: up to 14 iterations) / bytes ([1]) + the control words (gl_levels_pack): one pair of buffers
        // per WIDTH, so that calls alternating between N <= 14 and N >= 15 keep their recorded graphs (one pair,
        // re-allocated on every change of width, destroyed every graph each time)
        DeviceBuffer packed[2];
        void *h_packed[2] = {nullptr, nullptr};
        size_t h_packed_bytes[2] = {0, 0};
        std::map<std::tuple<uint32_t, uint32_t, int, int>, gl_graph> graphs;   // (slots, threshold bits, pull only, packed)
        ~Schedule() { clear(); }
        void clear() {
            for (auto &g : graphs) gl_graph_destroy(g.second);
            graphs.clear();
            for (int w = 0; w < 2; w++) {
                if (h_packed[w]) gl_host_free(h_packed[w]);
                h_packed[w] = nullptr;
                h_packed_bytes[w] = 0;
            }
            slots = 0;
        }
    } sched_;
    DeviceBuffer col_len_;       // column lengths of the WHOLE matrix (the slots' decisions weigh frontiers by them)
    std::vector<uint32_t> col_len_host_;
    uint64_t nnz_global_ = 0;
    RowShards shards_;           // world of one unless set_comm() was called
    uint32_t push_iterations_ = 0;

    bool schedule_ok_() {
        const char *e = getenv("GRAPHLILY_BFS_HOST_LOOP");
        if (!kFloat || (e && atoi(e) != 0)) return false;
        if (!SpMV_->plan_handle() || !SpMSpV_->plan_handle() || (float)semiring_.zero != 0.0f) return false;
        gl_spmv_plan_desc d;
        GRAPHLILY_CHECK(gl_spmv_plan_describe(SpMV_->plan_handle(), &d));
        return SpMV_->bits_words() > 0 && d.layout == GL_LAYOUT_BOOLEAN && d.segments == 1;
    }

    // the whole run as launches on the library's stream (nothing here waits or copies to the host)
    void enqueue_schedule_(uint32_t N, float threshold, bool pull_only) {
        Schedule &s = sched_;
        const uint32_t n = matrix_num_rows_;
        float *distance = (float *)s.both.raw();
        uint32_t *ctl = (uint32_t *)s.both.raw() + n, *vecs = (uint32_t *)s.vecs.raw();
        uint32_t *tally = vecs + (size_t)s.nvec * s.words;
        const float back = pull_only ? 0.0f : 1.0f;   // after the reference's rule has switched to pulling, the direction follows the work
        auto may_of = [N](uint32_t it) { return (int)((it + 1 < N ? 1 : 0) | (it + 1 <= N ? 2 : 0)); };
        GRAPHLILY_CHECK(gl_bfs_bits_begin(ctl, s.ctl_words, distance, n, vecs, s.words, s.nvec_all, pull_only ? 0u : 0xffffffffu));
        for (uint32_t it = 1; it <= N; it++) {
            GRAPHLILY_CHECK(gl_bfs_bits_shard_step(SpMSpV_->plan_handle(), SpMV_->plan_handle(), vecs + (size_t)it * s.words,
                                                   vecs + (size_t)(it + 1) * s.words, s.words, distance, (float)(it + 1), ctl, tally, nullptr, it,
                                                   shards_.rank, shards_.world, (const uint32_t *)col_len_.raw(), nnz_global_, threshold,
                                                   it > 1 ? may_of(it - 1) : 0, back));
            if (shards_.comm)   // every rank's rows of the new frontier + every rank's tallies of the slot: one grouped exchange
                GRAPHLILY_CHECK(gl_dist_all_gather_bits_tally(shards_.comm, vecs + (size_t)(it + 1) * s.words, shards_.bounds.data(),
                                                              tally + GL_BFS_TALLY_HEAD_WORDS + (size_t)(it - 1) * shards_.world * GL_BFS_TALLY_RANK_WORDS,
                                                              4u * GL_BFS_TALLY_RANK_WORDS));
        }
        GRAPHLILY_CHECK(gl_bfs_bits_shard_finish(SpMSpV_->plan_handle(), SpMV_->plan_handle(), ctl, tally, nullptr, N, shards_.rank, shards_.world,
                                                 nnz_global_, threshold, may_of(N), back));
        if (shards_.comm) GRAPHLILY_CHECK(gl_dist_all_gather_f32(shards_.comm, distance, shards_.bounds.data()));
    }

    aligned_dense_vec_t run_schedule_(uint32_t source, uint32_t N, float threshold, bool pull_only) {
        Schedule &s = sched_;
        const uint32_t n = matrix_num_rows_;
        if (!s.both.valid() || s.slots < N) {   // (N == 0 as the first call: the buffers exist all the same -- ADVICE r05)
            s.clear();
            s.slots = std::max<uint32_t>(N, 1u);
            s.words = ((uint32_t)SpMV_->bits_words() + 3u) & ~3u;
            s.nvec = s.slots + 2;
            s.ctl_words = (18u + 2u * s.slots + 15u) & ~15u;
            const size_t tally_words = GL_BFS_TALLY_WORDS(s.slots, shards_.world);
            s.nvec_all = s.nvec + (uint32_t)((tally_words + s.words - 1) / s.words);
            s.both = DeviceBuffer(sizeof(float) * ((size_t)n + s.ctl_words));
            s.vecs = DeviceBuffer(sizeof(uint32_t) * (size_t)s.nvec_all * s.words);
        }
        // Levels are small integers: up to 14 iterations they fit a nibble, up to 254 a byte -- 1.5 / 3 MB over PCIe instead of
        // 12 MB on orkut -- and a few host threads expand them (gl_sync_levels_unpack).  Small vectors and hosts with few
        // threads copy the floats (GRAPHLILY_BFS_U8=0 pins that).
        const int bits = N + 1 <= 15 ? 4 : 8;
        const char *pin = getenv("GRAPHLILY_BFS_U8");
        const bool packed = N + 1 <= 255 && n % 8 == 0 && n >= (1u << 19) && !(pin && atoi(pin) == 0) && gl_host_unpack_threads() >= 4;
        uint32_t *ctl = (uint32_t *)s.both.raw() + n;
        size_t packed_words = 0;
        const int pw = bits == 4 ? 0 : 1;
        // Round 6: the pack kernel stores into the page-locked block itself, a flag behind every chunk, and the host threads expand
        // chunk k while chunk k + 1 crosses PCIe (gl_levels_pack_stream; GRAPHLILY_BFS_STREAM=0: pack -> copy -> wait -> expand)
        static const bool streamed = !(getenv("GRAPHLILY_BFS_STREAM") && atoi(getenv("GRAPHLILY_BFS_STREAM")) == 0);
        if (packed) {
            packed_words = ((size_t)n * bits / 8 + 15) / 16 * 4;     // levels, padded to 16 bytes; the control words follow
            size_t bytes = 4 * (packed_words + s.ctl_words);
            if (streamed) GRAPHLILY_CHECK(gl_levels_stream_bytes(n, bits, s.ctl_words, &bytes));
            if (s.h_packed_bytes[pw] != bytes) {   // (first use of this width since the schedule's buffers were made: no graph holds it yet)
                if (s.h_packed[pw]) gl_host_free(s.h_packed[pw]);
                GRAPHLILY_CHECK(gl_host_alloc(&s.h_packed[pw], bytes));
                s.h_packed_bytes[pw] = bytes;
                if (!streamed) s.packed[pw] = DeviceBuffer(bytes);
            }
            if (streamed) GRAPHLILY_CHECK(gl_levels_stream_arm(s.h_packed[pw], n, bits, s.ctl_words));   // (before this run's launches)
        }
        auto read_back = [&]() -> int {
            if (streamed) return gl_levels_pack_stream((const float *)s.both.raw(), n, bits, ctl, s.ctl_words, s.h_packed[pw]);
            const int rc = gl_levels_pack((const float *)s.both.raw(), n, bits, ctl, s.ctl_words, s.packed[pw].raw());
            return rc != GL_OK ? rc : gl_buf_d2h_async(s.h_packed[pw], s.packed[pw].raw(), s.h_packed_bytes[pw]);
        };
        auto everything = [&] {
            enqueue_schedule_(N, threshold, pull_only);
            if (packed) GRAPHLILY_CHECK(read_back());
        };
        GRAPHLILY_CHECK(gl_buf_fill_u32(ctl + 2, source, 1));          // ctl[2] = source: the recorded sequence serves any source
        uint32_t tbits;
        memcpy(&tbits, &threshold, 4);
        const auto key = std::make_tuple(N, tbits, (int)pull_only, (int)packed);
        auto g = s.graphs.find(key);
        if (g != s.graphs.end() && g->second) {
            GRAPHLILY_CHECK(gl_graph_launch(g->second));
        } else {
            everything();
            if (g == s.graphs.end()) {
                // first call with these arguments: it ran launch by launch (kernels take their one-time attributes then); the
                // same sequence is now RECORDED, without running, so that the next call is one graph launch
                gl_graph rec = nullptr;
                if (gl_graph_begin_capture() == GL_OK) {
                    enqueue_schedule_(N, threshold, pull_only);
                    if (packed) (void)read_back();
                    if (gl_graph_end_capture(&rec) != GL_OK) rec = nullptr;
                }
                s.graphs[key] = rec;      // (nullptr: capture is not possible here -- keep enqueueing)
            }
        }
        aligned_dense_vec_t result;
        {
            graphlily_detail::NoInitScope no_fill;   // (every element is written below)
            result.resize(n);
        }
        std::vector<uint32_t> c(s.ctl_words);
        if (packed) {
            if (streamed) {
                GRAPHLILY_CHECK(gl_sync_levels_unpack_stream((float *)result.data(), s.h_packed[pw], n, bits, c.data(), s.ctl_words));
            } else {
                GRAPHLILY_CHECK(gl_sync_levels_unpack((float *)result.data(), s.h_packed[pw], n, bits));
                memcpy(c.data(), (const uint32_t *)s.h_packed[pw] + packed_words, 4u * s.ctl_words);
            }
        } else {
            GRAPHLILY_CHECK(gl_buf_d2h(result.data(), s.both.raw(), sizeof(float) * (size_t)n));
            GRAPHLILY_CHECK(gl_buf_d2h(c.data(), ctl, 4u * s.ctl_words));
        }
        push_iterations_ = c[1];
        return result;
    }

    // ---- the reference's module-call sequences (bfs.h:106-219), for push() and as the fallback of the other two
    aligned_dense_vec_t dense_(uint32_t source, graphlily::val_t fill, graphlily::val_t at_source) {
        aligned_dense_vec_t v(matrix_num_rows_, fill);
        v[source] = at_source;
        return v;
    }
    void bind_pull_() {
        DenseAssign_->bind_mask_buf(SpMV_->vector_buf);
        DenseAssign_->bind_inout_buf(SpMV_->mask_buf);
        eWiseAdd_->bind_in_buf(SpMV_->results_buf);
        eWiseAdd_->bind_out_buf(SpMV_->vector_buf);
    }
    void pull_iteration_(uint32_t iter) {   // masked SpMV, results -> vector (eWiseAdd + 0), distance[new] = level
        SpMV_->run();
        eWiseAdd_->run(matrix_num_rows_, 0);
        DenseAssign_->run(matrix_num_rows_, iter + 1);
    }
    void start_push_(uint32_t source) {
        aligned_sparse_vec_t frontier(2);
        idx_val_t head;
        head.index = 1;   // one source vertex
        head.val = 0;
        frontier[0] = head;
        frontier[1] = {source, 1};
        aligned_dense_vec_t distance = dense_(source, 0, 1);
        SpMSpV_->send_vector_host_to_device(frontier);
        SpMSpV_->send_mask_host_to_device(distance);
        SparseAssign_->bind_mask_buf(SpMSpV_->vector_buf);
        SparseAssign_->bind_inout_buf(SpMSpV_->mask_buf);
    }
    uint32_t push_iteration_(uint32_t iter) {   // SpMSpV, results -> vector, distance[new] = level; returns the new frontier's size
        SpMSpV_->run();
        const uint32_t nnz = SpMSpV_->get_results_nnz();
        SpMSpV_->copy_buffer_device_to_device(SpMSpV_->results_buf, SpMSpV_->vector_buf, sizeof(graphlily::idx_val_t) * (1 + (size_t)nnz));
        SparseAssign_->run(iter + 1);
        return nnz;
    }
    void switch_to_pull_() {   // bfs.h:195-205: the sparse frontier becomes the SpMV's dense input, through the host
        SpMV_->bind_mask_buf(SpMSpV_->mask_buf);
        aligned_sparse_vec_t sparse = SpMSpV_->send_results_device_to_host();
        aligned_dense_vec_t dense = graphlily::convert_sparse_vec_to_dense_vec<aligned_sparse_vec_t, aligned_dense_vec_t, graphlily::val_t>(
            sparse, matrix_num_rows_, graphlily::LogicalSemiring.zero);
        SpMV_->send_vector_host_to_device(dense);
        bind_pull_();
    }

public:
    BFS(uint32_t num_channels, uint32_t spmv_out_buf_len, uint32_t spmspv_out_buf_len, uint32_t vec_buf_len)
        : num_channels_(num_channels), spmv_out_buf_len_(spmv_out_buf_len), spmspv_out_buf_len_(spmspv_out_buf_len), vec_buf_len_(vec_buf_len) {
        SpMV_ = new module::SpMVModule<graphlily::val_t, graphlily::val_t>(num_channels_, spmv_out_buf_len_, vec_buf_len_);
        SpMV_->set_semiring(semiring_);
        SpMV_->set_mask_type(graphlily::kMaskWriteToZero);
        add_module(SpMV_);
        DenseAssign_ = new module::AssignVectorDenseModule<graphlily::val_t>();
        DenseAssign_->set_mask_type(graphlily::kMaskWriteToOne);
        add_module(DenseAssign_);
        SpMSpV_ = new module::SpMSpVModule<graphlily::val_t, graphlily::val_t, graphlily::idx_val_t>(spmspv_out_buf_len_);
        SpMSpV_->set_semiring(semiring_);
        SpMSpV_->set_mask_type(graphlily::kMaskWriteToZero);
        add_module(SpMSpV_);
        SparseAssign_ = new module::AssignVectorSparseModule<graphlily::val_t, graphlily::idx_val_t>(false);
        add_module(SparseAssign_);
        eWiseAdd_ = new module::eWiseAddModule<graphlily::val_t>();
        add_module(eWiseAdd_);
    }

    uint32_t get_nnz() { return SpMV_->get_nnz(); }

    // extension: this process is one rank of a row-sharded run; call before load_and_format_matrix
    void set_comm(gl_dist comm) { shards_.set_comm(comm); }
    // extension: how many iterations of the last pull_push() pushed before the rule switched to pulling (the reference prints it)
    uint32_t push_iterations() const { return push_iterations_; }

    void load_and_format_matrix(std::string csr_float_npz_path, bool skip_empty_rows) {
        CSRMatrix<float> csr_matrix = graphlily::io::load_csr_matrix_from_float_npz(csr_float_npz_path);
        graphlily::io::util_round_csr_matrix_dim(csr_matrix, num_channels_ * graphlily::pack_size, num_channels_ * graphlily::pack_size);
        for (auto &x : csr_matrix.adj_data) x = 1;
        CSCMatrix<float> csc_matrix = graphlily::io::csr2csc(csr_matrix);
        shards_.cut(csr_matrix.adj_indptr);
        if (shards_.world > 1 || shards_.comm) {
            SpMV_->set_row_shard(shards_.row_begin(), shards_.row_end());
            SpMSpV_->set_row_shard(shards_.row_begin(), shards_.row_end());
        }
        SpMV_->load_and_format_matrix(csr_matrix, skip_empty_rows);
        SpMSpV_->load_and_format_matrix(csc_matrix);
        matrix_num_rows_ = SpMV_->get_num_rows();
        matrix_num_cols_ = SpMV_->get_num_cols();
        assert(matrix_num_rows_ == matrix_num_cols_);
        col_len_host_.resize(csc_matrix.num_cols);
        for (uint32_t c = 0; c < csc_matrix.num_cols; c++) col_len_host_[c] = csc_matrix.adj_indptr[c + 1] - csc_matrix.adj_indptr[c];
        nnz_global_ = csr_matrix.adj_indptr[csr_matrix.num_rows];
    }

    void send_matrix_host_to_device() {
        SpMV_->send_matrix_host_to_device();
        SpMSpV_->send_matrix_host_to_device();
        sched_.clear();      // buffers sized for the old matrix, graphs with the old plans' pointers baked in
        col_len_ = DeviceBuffer(sizeof(uint32_t) * std::max<size_t>(col_len_host_.size(), 1));
        if (!col_len_host_.empty()) col_len_.upload(col_len_host_.data(), sizeof(uint32_t) * col_len_host_.size());
    }

    aligned_dense_vec_t pull(uint32_t source, uint32_t num_iterations) {
        if (num_iterations > 0 && schedule_ok_()) return run_schedule_(source, num_iterations, -1.0f, true);   // (0 iterations: the host loop returns the start vector)
        aligned_dense_vec_t input = dense_(source, semiring_.zero, 1), distance = dense_(source, 0, 1);
        SpMV_->send_vector_host_to_device(input);
        SpMV_->send_mask_host_to_device(distance);
        bind_pull_();
        for (uint32_t iter = 1; iter <= num_iterations; iter++) pull_iteration_(iter);
        return SpMV_->send_mask_device_to_host();
    }

    aligned_dense_vec_t push(uint32_t source, uint32_t num_iterations) {
        start_push_(source);
        for (uint32_t iter = 1; iter <= num_iterations; iter++) push_iteration_(iter);
        return SpMSpV_->send_mask_device_to_host();
    }

    aligned_dense_vec_t pull_push(uint32_t source, uint32_t num_iterations, float threshold = 0.05) {
        if (num_iterations > 0 && schedule_ok_()) {
            aligned_dense_vec_t r = run_schedule_(source, num_iterations, threshold, false);
            std::cout << "SpMSpV runs for " << push_iterations_ << " iterations" << std::endl;
            return r;
        }
        start_push_(source);
        uint32_t iter = 1, vector_nnz;
        do {
            vector_nnz = push_iteration_(iter);
            iter++;
        } while (iter < num_iterations && (float(vector_nnz) / matrix_num_rows_ < threshold));
        push_iterations_ = iter - 1;
        std::cout << "SpMSpV runs for " << (iter - 1) << " iterations" << std::endl;
        switch_to_pull_();
        for (; iter <= num_iterations; iter++) pull_iteration_(iter);
        return SpMSpV_->send_mask_device_to_host();   // the mask of SpMV on the host is not valid
    }

    // the reference's four wall-clock buckets (bfs.h:222-347) around the module-call sequence, every call followed by a device
    // synchronisation (the reference's module calls are blocking)
    aligned_dense_vec_t pull_push_time_breakdown(uint32_t source, uint32_t num_iterations, float threshold = 0.05) {
        typedef std::chrono::high_resolution_clock clk;
        float spmv_spmspv_ms = 0, assign_ms = 0, transfer_ms = 0;
        auto timed = [](float &bucket, const std::function<void()> &fn) {
            GRAPHLILY_CHECK(gl_sync());
            const auto t0 = clk::now();
            fn();
            GRAPHLILY_CHECK(gl_sync());
            bucket += float(std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count()) / 1000;
        };
        const auto start = clk::now();
        timed(transfer_ms, [&] { start_push_(source); });
        uint32_t iter = 1, vector_nnz = 0;
        do {
            timed(spmv_spmspv_ms, [&] { SpMSpV_->run(); });
            vector_nnz = SpMSpV_->get_results_nnz();
            timed(transfer_ms, [&] {
                SpMSpV_->copy_buffer_device_to_device(SpMSpV_->results_buf, SpMSpV_->vector_buf, sizeof(graphlily::idx_val_t) * (1 + (size_t)vector_nnz));
            });
            timed(assign_ms, [&] { SparseAssign_->run(iter + 1); });
            iter++;
        } while (iter < num_iterations && (float(vector_nnz) / matrix_num_rows_ < threshold));
        std::cout << "SpMSpV runs for " << (iter - 1) << " iterations" << std::endl;
        switch_to_pull_();
        for (; iter <= num_iterations; iter++) {
            timed(spmv_spmspv_ms, [&] { SpMV_->run(); });
            timed(transfer_ms, [&] { eWiseAdd_->run(matrix_num_rows_, 0); });
            timed(assign_ms, [&] { DenseAssign_->run(matrix_num_rows_, iter + 1); });
        }
        aligned_dense_vec_t result;
        timed(transfer_ms, [&] { result = SpMSpV_->send_mask_device_to_host(); });
        const float total_ms = float(std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - start).count()) / 1000;
        std::cout << "total_time_ms: " << total_ms << std::endl;
        std::cout << "spmv_spmspv_time_ms: " << spmv_spmspv_ms << std::endl;
        std::cout << "assign_time_ms: " << assign_ms << std::endl;
        std::cout << "data_transfer_time_ms: " << transfer_ms << std::endl;
        std::cout << "overhead_time_ms: " << total_ms - spmv_spmspv_ms - assign_ms - transfer_ms << std::endl;
        return result;
    }

    // the CPU result callers verify against (bfs.h:350-360): masked (||,&&) SpMV + dense assign of the level, per iteration
    aligned_dense_float_vec_t compute_reference_results(uint32_t source, uint32_t num_iterations) {
        aligned_dense_float_vec_t input(matrix_num_rows_, semiring_.zero), distance(matrix_num_rows_, 0);
        input[source] = 1;
        distance[source] = 1;
        for (uint32_t iter = 1; iter <= num_iterations; iter++) {
            input = SpMV_->compute_reference_results(input, distance);
            DenseAssign_->compute_reference_results(input, distance, matrix_num_rows_, iter + 1);
        }
        return distance;
    }
};

}  // namespace app
}  // namespace graphlily

#endif  // GRAPHLILY_HIP_APP_BFS_H_
#endif  // GRAPHLILY_USE_REFERENCE_APPS
