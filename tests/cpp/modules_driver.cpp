// C++ parity driver for the header-only module layer (include/graphlily/module/*.h): every module is
// run on the GPU through the C ABI and checked against its own compute_reference_results, following
// the reference's tests/test_module_spmv_spmspv.cpp and tests/test_module_apply.cpp with seeded inputs.
//   g++ -std=c++11 -I include tests/cpp/modules_driver.cpp -L graphlily_amd/lib -lgraphlily_hip
#include <cmath>
#include <cstdio>
#include <random>

#include "graphlily/app/module_collection.h"
#include "graphlily/module/add_scalar_vector_dense_module.h"
#include "graphlily/module/assign_vector_dense_module.h"
#include "graphlily/module/assign_vector_sparse_module.h"
#include "graphlily/module/spmspv_module.h"
#include "graphlily/module/spmv_module.h"

using namespace graphlily;
typedef aligned_dense_float_vec_t fvec;

static int failures = 0;

static void verify(const fvec &ref, const fvec &got, const char *what, bool exact) {
    bool ok = ref.size() == got.size();
    for (size_t i = 0; ok && i < ref.size(); i++) {
        if (exact) ok = ref[i] == got[i];
        else ok = std::fabs(got[i] - ref[i]) <= 1e-4f * std::fmax(1.0f, std::fabs(ref[i]));   // reference eps, relative
        if (!ok) printf("  %s: row %zu ref %g got %g\n", what, i, ref[i], got[i]);
    }
    printf("%-58s %s\n", what, ok ? "OK" : "FAIL");
    failures += !ok;
}

static CSRMatrix<float> uniform_csr(uint32_t n, uint32_t deg, uint32_t seed) {
    std::mt19937 rng(seed);
    CSRMatrix<float> m;
    m.num_rows = m.num_cols = n;
    m.adj_indptr.push_back(0);
    for (uint32_t r = 0; r < n; r++) {
        std::vector<uint32_t> cols;
        while (cols.size() < deg) {
            uint32_t c = rng() % n;
            if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
        }
        std::sort(cols.begin(), cols.end());
        for (uint32_t c : cols) { m.adj_indices.push_back(c); m.adj_data.push_back(1.0f / n); }
        m.adj_indptr.push_back((uint32_t)m.adj_indices.size());
    }
    return m;
}

int main() {
    std::mt19937 rng(42);
    CSRMatrix<float> csr = uniform_csr(10000, 10, 7);
    io::util_round_csr_matrix_dim(csr, num_hbm_channels * pack_size, pack_size);
    const SemiringType tropical255 = {kAddMin, 0, UFIXED_INF};
    const SemiringType sems[] = {ArithmeticSemiring, LogicalSemiring, tropical255, TropicalSemiring};
    const char *sem_names[] = {"Arithmetic", "Logical", "Tropical(255)", "Tropical(FLOAT_INF)"};
    const MaskType masks[] = {kNoMask, kMaskWriteToZero, kMaskWriteToOne};
    const char *mask_names[] = {"NoMask", "WriteToZero", "WriteToOne"};
    char name[128];

    {   // SpMV
        module::SpMVModule<val_t, val_t> spmv(num_hbm_channels, 1024, 256);
        spmv.set_target("hw");
        spmv.set_up_runtime("unused.xclbin");
        spmv.load_and_format_matrix(csr, true);
        spmv.send_matrix_host_to_device();
        fvec x(csr.num_cols), mask(csr.num_rows);
        for (auto &v : x) v = float(rng() % 2);
        for (auto &v : mask) v = float(rng() % 2);
        spmv.send_vector_host_to_device(x);
        spmv.send_mask_host_to_device(mask);
        for (int s = 0; s < 4; s++)
            for (int k = 0; k < 3; k++) {
                spmv.set_semiring(sems[s]);
                spmv.set_mask_type(masks[k]);
                spmv.run();
                fvec ref = (masks[k] == kNoMask) ? spmv.compute_reference_results(x) : spmv.compute_reference_results(x, mask);
                snprintf(name, sizeof name, "SpMV %s %s", sem_names[s], mask_names[k]);
                verify(ref, spmv.send_results_device_to_host(), name, s != 0);
            }
    }
    {   // extensions: frontier as bits, fused BFS pull step == SpMV(WriteToZero by distance) + eWiseAdd(0) + assign(level)
        module::SpMVModule<val_t, val_t> spmv(num_hbm_channels, 1024, 256);
        spmv.set_semiring(LogicalSemiring);
        spmv.set_mask_type(kMaskWriteToZero);
        spmv.set_up_runtime("unused.xclbin");
        spmv.load_and_format_matrix(csr, true);
        spmv.send_matrix_host_to_device();
        const uint32_t n = csr.num_rows;
        fvec x(csr.num_cols), dist(n);
        for (auto &v : x) v = float(rng() % 8 == 0);
        for (auto &v : dist) v = (rng() % 3 == 0) ? 2.0f : 0.0f;
        spmv.send_vector_host_to_device(x);
        spmv.send_mask_host_to_device(dist);
        const uint64_t words = spmv.bits_words();
        if (words == 0) { printf("boolean layout expected\n"); failures++; }
        DeviceBuffer bits_in(4 * words), bits_out(4 * words);
        std::vector<uint32_t> zero_words(words, 0u);
        bits_in.upload(zero_words.data(), 4 * words);
        bits_out.upload(zero_words.data(), 4 * words);
        module::SpMVModule<val_t, val_t>::pack_bits(spmv.vector_buf, csr.num_cols, bits_in);
        spmv.run_bits(bits_in);
        fvec y = spmv.send_results_device_to_host();
        verify(spmv.compute_reference_results(x, dist), y, "SpMV run_bits Logical WriteToZero", true);
        fvec want = dist;
        for (uint32_t r = 0; r < n; r++) if (y[r] != 0) want[r] = 5.0f;
        if (!spmv.bfs_pull_step(bits_in, bits_out, spmv.mask_buf, 5.0f)) { printf("bfs_pull_step unsupported\n"); failures++; }
        verify(want, spmv.send_mask_device_to_host(), "fused BFS pull step: distance", true);
        std::vector<uint32_t> got_bits(words);
        bits_out.download(got_bits.data(), 4 * words);
        fvec front(n);
        for (uint32_t r = 0; r < n; r++) front[r] = float((got_bits[r >> 5] >> (r & 31)) & 1u);
        verify(y, front, "fused BFS pull step: next frontier bits", true);
    }
    {   // SpMSpV
        CSCMatrix<float> csc = io::csr2csc(csr);
        module::SpMSpVModule<val_t, val_t, idx_val_t> sp(512);
        sp.set_target("hw");
        sp.set_up_runtime("unused.xclbin");
        sp.load_and_format_matrix(csc);
        sp.send_matrix_host_to_device();
        const uint32_t nnz = csc.num_cols / 100;
        aligned_sparse_float_vec_t vf(nnz + 1);
        aligned_sparse_vec_t v(nnz + 1);
        vf[0] = idx_float_t{nnz, 0};
        for (uint32_t i = 0; i < nnz; i++) vf[i + 1] = idx_float_t{i * 100, float(rng() % 10) / 10};
        for (uint32_t i = 0; i <= nnz; i++) v[i] = idx_val_t{vf[i].index, vf[i].val};
        fvec mask(csc.num_rows);
        for (auto &m : mask) m = float(rng() % 2);
        sp.send_mask_host_to_device(mask);
        sp.send_vector_host_to_device(v);
        for (int s = 0; s < 3; s++)
            for (int k = 0; k < 3; k++) {
                sp.set_semiring(sems[s]);
                sp.set_mask_type(masks[k]);
                sp.run();
                aligned_sparse_vec_t res = sp.send_results_device_to_host();
                if (res[0].index != sp.get_results_nnz()) { printf("get_results_nnz mismatch\n"); failures++; }
                fvec got = convert_sparse_vec_to_dense_vec<aligned_sparse_vec_t, fvec, float>(res, csc.num_rows, sems[s].zero);
                snprintf(name, sizeof name, "SpMSpV %s %s", sem_names[s], mask_names[k]);
                verify(sp.compute_reference_results(vf, mask), got, name, s != 0);
            }
        {   // extension: run_assign == run + AssignVectorSparse::run(val), with the mask as the assigned vector (BFS)
            sp.set_semiring(LogicalSemiring);
            sp.set_mask_type(kMaskWriteToZero);
            sp.send_mask_host_to_device(mask);
            sp.run_assign(sp.mask_buf, 9.0f);
            fvec y = sp.compute_reference_results(vf, mask), want = mask;
            for (size_t i = 0; i < want.size(); i++)
                if (y[i] != 0) want[i] = 9.0f;
            verify(want, sp.send_mask_device_to_host(), "SpMSpV run_assign: assigned vector", true);
            aligned_sparse_vec_t res = sp.send_results_device_to_host();
            fvec got = convert_sparse_vec_to_dense_vec<aligned_sparse_vec_t, fvec, float>(res, csc.num_rows, 0.0f);
            verify(y, got, "SpMSpV run_assign: results", true);
        }
    }
    {   // apply modules
        const uint32_t len = 128 * 100;
        fvec in(len), mask(len), inout(len);
        for (auto &v : in) v = float(rng() % 10) / 100;
        for (auto &v : mask) v = float(rng() % 2);
        for (auto &v : inout) v = float(rng() % 2);
        module::eWiseAddModule<val_t> add;
        add.set_up_runtime("unused.xclbin");
        add.send_in_host_to_device(in);
        add.allocate_out_buf(len);
        add.run(len, 1);
        verify(add.compute_reference_results(in, len, 1), add.send_out_device_to_host(), "eWiseAdd", true);

        module::AssignVectorDenseModule<val_t> dense;
        dense.set_up_runtime("unused.xclbin");
        dense.set_mask_type(kMaskWriteToOne);
        dense.send_mask_host_to_device(mask);
        dense.send_inout_host_to_device(inout);
        dense.run(len, 23);
        fvec ref = inout;
        dense.compute_reference_results(mask, ref, len, 23);
        verify(ref, dense.send_inout_device_to_host(), "AssignVectorDense WriteToOne", true);

        const uint32_t n = 8192, cnt = 819;
        aligned_sparse_float_vec_t mf(cnt + 1);
        aligned_sparse_vec_t ms(cnt + 1);
        mf[0] = idx_float_t{cnt, 0};
        for (uint32_t i = 0; i < cnt; i++) mf[i + 1] = idx_float_t{i * 10, float(rng() % 10)};
        for (uint32_t i = 0; i <= cnt; i++) ms[i] = idx_val_t{mf[i].index, mf[i].val};
        fvec io1(n), io2(n);
        for (auto &v : io1) v = float(rng() % 10);
        for (auto &v : io2) v = (rng() % 10 > 5) ? 5.0f : FLOAT_INF;
        module::AssignVectorSparseModule<val_t, idx_val_t> bfs_mode(false);
        bfs_mode.set_up_runtime("unused.xclbin");
        bfs_mode.send_mask_host_to_device(ms);
        bfs_mode.send_inout_host_to_device(io1);
        bfs_mode.run(3);
        ref = io1;
        bfs_mode.compute_reference_results(mf, ref, 3.0f);
        verify(ref, bfs_mode.send_inout_device_to_host(), "AssignVectorSparse (no new frontier)", true);

        module::AssignVectorSparseModule<val_t, idx_val_t> sssp_mode(true);
        sssp_mode.set_up_runtime("unused.xclbin");
        sssp_mode.send_mask_host_to_device(ms);
        sssp_mode.send_inout_host_to_device(io2);
        sssp_mode.run();
        ref = io2;
        aligned_sparse_float_vec_t nf_ref;
        sssp_mode.compute_reference_results(mf, ref, nf_ref);
        verify(ref, sssp_mode.send_inout_device_to_host(), "AssignVectorSparse (new frontier) inout", true);
        aligned_sparse_vec_t nf = sssp_mode.send_new_frontier_device_to_host();
        fvec d_ref = convert_sparse_vec_to_dense_vec<aligned_sparse_float_vec_t, fvec, float>(nf_ref, n, 0);
        fvec d_got = convert_sparse_vec_to_dense_vec<aligned_sparse_vec_t, fvec, float>(nf, n, 0);
        verify(d_ref, d_got, "AssignVectorSparse (new frontier) frontier", true);

        // copy_buffer_device_to_device + bind_* (reference TEST(CopyBufferBindBuffer, Basic))
        dense.send_mask_host_to_device(mask);
        dense.send_inout_host_to_device(inout);
        dense.copy_buffer_device_to_device(dense.mask_buf, dense.inout_buf, sizeof(val_t) * len);
        verify(mask, dense.send_inout_device_to_host(), "copy_buffer_device_to_device", true);
    }
    // ---- the BFS pull iteration of the reference's driver (app/bfs.h:106-126) through a ModuleCollection: the three module
    //      calls run fused (module/fusion.h).  After EVERY iteration the three buffers -- read through the module API -- must
    //      hold what the unfused sequence leaves (GRAPHLILY_MODULE_FUSION=0), and reading them must not disturb the next one.
    {
        struct Bfs : public app::ModuleCollection {
            module::SpMVModule<val_t, val_t> *SpMV;
            module::SpMSpVModule<val_t, val_t, idx_val_t> *SpMSpV;
            module::eWiseAddModule<val_t> *eWise;
            module::AssignVectorDenseModule<val_t> *Assign;
            Bfs() {
                SpMV = new module::SpMVModule<val_t, val_t>(16, 1024, 256);
                SpMV->set_semiring(LogicalSemiring);
                SpMV->set_mask_type(kMaskWriteToZero);
                SpMSpV = new module::SpMSpVModule<val_t, val_t, idx_val_t>(1024);
                SpMSpV->set_semiring(LogicalSemiring);
                SpMSpV->set_mask_type(kMaskWriteToZero);
                eWise = new module::eWiseAddModule<val_t>();
                Assign = new module::AssignVectorDenseModule<val_t>();
                Assign->set_mask_type(kMaskWriteToOne);
                add_module(SpMV);
                add_module(SpMSpV);
                add_module(eWise);
                add_module(Assign);
            }
        };
        CSRMatrix<float> g = uniform_csr(20000, 6, 11);
        io::util_round_csr_matrix_dim(g, 128, 128);
        for (auto &x : g.adj_data) x = 1;
        CSCMatrix<float> gc = io::csr2csc(g);
        const uint32_t n = g.num_rows, iters = 7;
        std::vector<fvec> seen[2];
        for (int fused = 0; fused < 2; fused++) {
            setenv("GRAPHLILY_MODULE_FUSION", fused ? "1" : "0", 1);
            Bfs bfs;
            bfs.set_up_runtime("unused.xclbin");
            bfs.SpMV->load_and_format_matrix(g, true);
            bfs.SpMSpV->load_and_format_matrix(gc);
            bfs.SpMV->send_matrix_host_to_device();
            bfs.SpMSpV->send_matrix_host_to_device();
            aligned_dense_vec_t input(n, 0), distance(n, 0);
            input[3] = 1;
            distance[3] = 1;
            bfs.SpMV->send_vector_host_to_device(input);
            bfs.SpMV->send_mask_host_to_device(distance);
            bfs.Assign->bind_mask_buf(bfs.SpMV->vector_buf);
            bfs.Assign->bind_inout_buf(bfs.SpMV->mask_buf);
            bfs.eWise->bind_in_buf(bfs.SpMV->results_buf);
            bfs.eWise->bind_out_buf(bfs.SpMV->vector_buf);
            for (uint32_t it = 1; it <= iters; it++) {
                bfs.SpMV->run();
                bfs.eWise->run(n, 0);
                bfs.Assign->run(n, float(it + 1));
                // iterations 1, 2 and 5 are inspected (a reader in the middle of a run), 3, 4, 6, 7 chain fused step to fused step
                if (it <= 2 || it == 5) {
                    seen[fused].push_back(bfs.SpMV->send_vector_device_to_host());
                    seen[fused].push_back(bfs.SpMV->send_results_device_to_host());
                    seen[fused].push_back(bfs.SpMV->send_mask_device_to_host());
                }
            }
            seen[fused].push_back(bfs.SpMV->send_mask_device_to_host());
            seen[fused].push_back(bfs.SpMV->send_vector_device_to_host());
            // a lone SpMV run() (no eWiseAdd / assign behind it) still delivers its results
            bfs.SpMV->run();
            seen[fused].push_back(bfs.SpMV->send_results_device_to_host());
            // ... and so does SpMV + eWiseAdd followed by something else
            bfs.SpMV->run();
            bfs.eWise->run(n, 0);
            seen[fused].push_back(bfs.SpMV->send_vector_device_to_host());
        }
        unsetenv("GRAPHLILY_MODULE_FUSION");
        uint32_t reached = 0;
        for (float v : seen[0][seen[0].size() - 4]) reached += v != 0;
        printf("fused BFS pull: %u of %u vertices reached in %u iterations\n", reached, n, iters);
        failures += reached < n / 2;
        for (size_t k = 0; k < seen[0].size(); k++) {
            snprintf(name, sizeof(name), "fused BFS pull iteration == the three calls (read %zu)", k);
            verify(seen[0][k], seen[1][k], name, true);
        }
    }
    // ---- the pull loops of the reference's SSSP and PageRank drivers (app/sssp.h:152-166, app/pagerank.h:80-90) through a
    //      ModuleCollection: SpMV->run(); eWiseAdd->run(n, val) run as one SpMV whose result block SWAPS places with the vector's
    //      (module/fusion.h 3.).  Vector and results -- read through the module API at every iteration or only at the end -- must
    //      be what the two calls leave with the fusion off; (min,+) bit for bit, (+,x) to the f64-accumulated sum's rounding.
    for (int which = 0; which < 2; which++) {
        struct Pull : public app::ModuleCollection {
            module::SpMVModule<val_t, val_t> *SpMV;
            module::eWiseAddModule<val_t> *eWise;
            Pull(SemiringType sem) {
                SpMV = new module::SpMVModule<val_t, val_t>(16, 1024, 256);
                SpMV->set_semiring(sem);
                SpMV->set_mask_type(kNoMask);
                eWise = new module::eWiseAddModule<val_t>();
                add_module(SpMV);
                add_module(eWise);
            }
        };
        CSRMatrix<float> g = uniform_csr(20000, 6, 23);
        io::util_round_csr_matrix_dim(g, 128, 128);
        const uint32_t n = g.num_rows, iters = 6;
        const bool sssp = which == 0;
        const float val = sssp ? 0.0f : 0.1f / n;
        for (size_t i = 0; i < g.adj_data.size(); i++) g.adj_data[i] = sssp ? float(1 + i % 3) : 0.9f / 6.0f;
        std::vector<fvec> seen[2];
        for (int fused = 0; fused < 2; fused++) {
            setenv("GRAPHLILY_MODULE_FUSION", fused ? "1" : "0", 1);
            Pull p(sssp ? TropicalSemiring : ArithmeticSemiring);
            p.set_up_runtime("unused.xclbin");
            p.SpMV->load_and_format_matrix(g, true);
            p.SpMV->send_matrix_host_to_device();
            aligned_dense_vec_t x(n, sssp ? (float)TropicalSemiring.zero : 1.0f / n);
            if (sssp) x[3] = 0;
            p.SpMV->send_vector_host_to_device(x);
            p.eWise->bind_in_buf(p.SpMV->results_buf);
            p.eWise->bind_out_buf(p.SpMV->vector_buf);
            for (uint32_t it = 1; it <= iters; it++) {
                p.SpMV->run();
                p.eWise->run(n, val);
                if (it == 2) seen[fused].push_back(p.SpMV->send_results_device_to_host());   // a reader of the owed results, mid-run
                if (it == 3) seen[fused].push_back(p.SpMV->send_vector_device_to_host());
            }
            seen[fused].push_back(p.SpMV->send_vector_device_to_host());
            seen[fused].push_back(p.SpMV->send_results_device_to_host());
            // a lone run() afterwards, and a run() + an eWiseAdd that does NOT match the pattern (another length)
            p.SpMV->run();
            seen[fused].push_back(p.SpMV->send_results_device_to_host());
            p.SpMV->run();
            p.eWise->run(n / 2, val);
            seen[fused].push_back(p.SpMV->send_results_device_to_host());
            seen[fused].push_back(p.SpMV->send_vector_device_to_host());
            // the vector overwritten IN PLACE while the swap's results are still owed ("a copy of the vector, taken when somebody
            // reads it"): the results must be what the SpMV left, not what was uploaded since
            p.SpMV->run();
            p.eWise->run(n, val);
            aligned_dense_vec_t x2(n, 0.25f);
            p.SpMV->vector_buf.upload(x2.data(), (size_t)n * sizeof(val_t));
            seen[fused].push_back(p.SpMV->send_results_device_to_host());
            seen[fused].push_back(p.SpMV->send_vector_device_to_host());
            // a module destroyed while its swap's results are still owed, with ANOTHER module's SpMV deferred in between: the
            // buffer (kept alive by this handle) must hold the results, and settling it must not need the dead module
            {
                Pull *a = new Pull(sssp ? TropicalSemiring : ArithmeticSemiring);
                a->set_up_runtime("unused.xclbin");
                a->SpMV->load_and_format_matrix(g, true);
                a->SpMV->send_matrix_host_to_device();
                a->SpMV->send_vector_host_to_device(x);
                a->eWise->bind_in_buf(a->SpMV->results_buf);
                a->eWise->bind_out_buf(a->SpMV->vector_buf);
                a->SpMV->run();
                a->eWise->run(n, val);
                auto keep = a->SpMV->results_buf;
                Pull b(sssp ? TropicalSemiring : ArithmeticSemiring);
                b.set_up_runtime("unused.xclbin");
                b.SpMV->load_and_format_matrix(g, true);
                b.SpMV->send_matrix_host_to_device();
                b.SpMV->send_vector_host_to_device(x2);
                b.SpMV->run();                                   // (deferred when the fusion is on)
                delete a;
                aligned_dense_vec_t got(n);
                keep.download(got.data(), (size_t)n * sizeof(val_t));
                seen[fused].push_back(got);
                seen[fused].push_back(b.SpMV->send_results_device_to_host());
            }
        }
        unsetenv("GRAPHLILY_MODULE_FUSION");
        for (size_t k = 0; k < seen[0].size(); k++) {
            snprintf(name, sizeof(name), "%s pull loop: SpMV + eWiseAdd as a swap == the two calls (read %zu)", sssp ? "SSSP" : "PageRank", k);
            verify(seen[0][k], seen[1][k], name, sssp);
        }
    }
    printf("%s\n", failures ? "SOME CHECKS FAILED" : "ALL CHECKS PASSED");
    return failures ? 1 : 0;
}
