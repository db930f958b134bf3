// The module layer instantiated with the reference's OTHER value types (global.h:62-64): built with -DGRAPHLILY_VAL_UFIXED
// (ap_ufixed<32, 8, AP_RND, AP_SAT>, the shipped default) or -DGRAPHLILY_VAL_UNSIGNED.  The driver holds no arithmetic of its
// own: it runs every module on the GPU and writes its inputs and outputs as 32-bit words into argv[1]; the test
// (tests/test_cpp_layer.py) replays the same inputs through the oracle's integer restatement and compares word for word.
//   g++ -std=c++11 -DGRAPHLILY_VAL_UFIXED -I include tests/cpp/typed_modules_driver.cpp -L graphlily_amd/lib -lgraphlily_hip
#include <cstdio>
#include <random>
#include <string>

#include "graphlily/module/add_scalar_vector_dense_module.h"
#include "graphlily/module/assign_vector_dense_module.h"
#include "graphlily/module/assign_vector_sparse_module.h"
#include "graphlily/module/spmspv_module.h"
#include "graphlily/module/spmv_module.h"

using namespace graphlily;
typedef value_kind<val_t> VK;

static std::string out_dir;

static void dump(const std::string &name, const void *p, size_t words) {
    const std::string path = out_dir + "/" + name + ".u32";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 4, words, f) != words) {
        printf("cannot write %s\n", path.c_str());
        exit(2);
    }
    fclose(f);
}
template <typename V>
static void dump_vec(const std::string &name, const V &v) {
    static_assert(sizeof(typename V::value_type) % 4 == 0, "whole words");
    dump(name, v.data(), v.size() * sizeof(typename V::value_type) / 4);
}

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s out_dir\n", argv[0]); return 2; }
    out_dir = argv[1];
    std::mt19937 rng(1234);
    const bool fixed = VK::kind == GL_VAL_UFIXED_32_8;

    // a uniform graph with float weights in steps of 1/8 (exact in every value type once converted)
    const uint32_t n = 4096, deg = 12;
    CSRMatrix<float> csr;
    csr.num_rows = csr.num_cols = n;
    csr.adj_indptr.push_back(0);
    for (uint32_t r = 0; r < n; r++) {
        std::vector<uint32_t> cols;
        const uint32_t d = (r == 17) ? 3000 : deg;       // one hub row: its (+,x) sum saturates the fixed point
        while (cols.size() < d) {
            const uint32_t c = rng() % n;
            if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
        }
        std::sort(cols.begin(), cols.end());
        for (uint32_t c : cols) {
            csr.adj_indices.push_back(c);
            csr.adj_data.push_back(fixed ? float(rng() % 40) / 8 + (rng() % 3 == 0 ? 1.0f / 3 : 0.0f) : float(rng() % 6));
        }
        csr.adj_indptr.push_back((uint32_t)csr.adj_indices.size());
    }
    io::util_round_csr_matrix_dim(csr, num_hbm_channels * pack_size, pack_size);
    dump_vec("csr_indptr", csr.adj_indptr);
    dump_vec("csr_indices", csr.adj_indices);
    dump_vec("csr_data_float", csr.adj_data);

    const SemiringType sems[] = {ArithmeticSemiring, LogicalSemiring, TropicalSemiring};
    const MaskType masks[] = {kNoMask, kMaskWriteToZero, kMaskWriteToOne};
    uint32_t zero_words[3];
    for (int s = 0; s < 3; s++) zero_words[s] = VK::bits(sems[s].zero);
    dump("zero_words", zero_words, 3);

    aligned_dense_vec_t x(csr.num_cols), mask(csr.num_rows);
    std::vector<float> x_float(csr.num_cols);
    for (uint32_t i = 0; i < csr.num_cols; i++) {
        x_float[i] = (rng() % 4 == 0) ? 0.0f : (fixed ? float(rng() % 1000) / 64 + 0.3f : float(rng() % 50));
        x[i] = val_t(x_float[i]);
    }
    for (auto &v : mask) v = val_t(rng() % 2);
    dump_vec("x_float", x_float);
    dump_vec("x", x);
    dump_vec("mask", mask);

    {   // SpMV: 3 semirings x 3 masks
        module::SpMVModule<val_t, val_t> spmv(num_hbm_channels, 1024, 256);
        spmv.set_target("hw");
        spmv.set_up_runtime("unused.xclbin");
        spmv.load_and_format_matrix(csr, true);
        spmv.send_matrix_host_to_device();
        spmv.send_vector_host_to_device(x);
        spmv.send_mask_host_to_device(mask);
        for (int s = 0; s < 3; s++)
            for (int k = 0; k < 3; k++) {
                spmv.set_semiring(sems[s]);
                spmv.set_mask_type(masks[k]);
                spmv.run();
                dump_vec("spmv_" + std::to_string(s) + "_" + std::to_string(k), spmv.send_results_device_to_host());
            }
    }
    {   // SpMSpV
        CSCMatrix<float> csc = io::csr2csc(csr);
        dump_vec("csc_indptr", csc.adj_indptr);
        dump_vec("csc_indices", csc.adj_indices);
        dump_vec("csc_data_float", csc.adj_data);
        module::SpMSpVModule<val_t, val_t, idx_val_t> sp(512);
        sp.set_target("hw");
        sp.set_up_runtime("unused.xclbin");
        sp.load_and_format_matrix(csc);
        sp.send_matrix_host_to_device();
        const uint32_t cnt = csc.num_cols / 16;
        aligned_sparse_vec_t v(cnt + 1);
        v[0].index = cnt;
        v[0].val = val_t(0);
        for (uint32_t i = 0; i < cnt; i++) {
            v[i + 1].index = i * 16 + rng() % 16;
            v[i + 1].val = val_t(fixed ? float(rng() % 500) / 32 : float(rng() % 30));
        }
        dump_vec("sv", v);
        sp.send_mask_host_to_device(mask);
        sp.send_vector_host_to_device(v);
        for (int s = 0; s < 3; s++)
            for (int k = 0; k < 3; k++) {
                sp.set_semiring(sems[s]);
                sp.set_mask_type(masks[k]);
                sp.run();
                aligned_sparse_vec_t res = sp.send_results_device_to_host();
                if (res[0].index != sp.get_results_nnz()) { printf("get_results_nnz mismatch\n"); return 1; }
                dump_vec("spmspv_" + std::to_string(s) + "_" + std::to_string(k), res);
            }
        // run_assign (extension): SpMSpV (||,&&) masked WriteToZero, then the results assigned into the mask vector
        sp.set_semiring(LogicalSemiring);
        sp.set_mask_type(kMaskWriteToZero);
        sp.send_mask_host_to_device(mask);
        sp.run_assign(sp.mask_buf, val_t(9));
        dump_vec("run_assign_inout", sp.send_mask_device_to_host());
        dump_vec("run_assign_results", sp.send_results_device_to_host());
    }
    {   // apply modules
        const uint32_t len = 128 * 100;
        aligned_dense_vec_t in(len), m(len), inout(len);
        for (auto &v : in) v = fixed ? val_t(float(rng() % 4096) / 16) : val_t(rng() % 2 ? 0xfffffff0u + rng() % 16 : rng() % 1000);
        for (auto &v : m) v = val_t(rng() % 2);
        for (auto &v : inout) v = val_t(rng() % 7);
        dump_vec("ewise_in", in);
        module::eWiseAddModule<val_t> add;
        add.set_up_runtime("unused.xclbin");
        add.send_in_host_to_device(in);
        add.allocate_out_buf(len);
        const val_t addend = fixed ? val_t(3.5) : val_t(11);
        const uint32_t addend_word = VK::bits(addend);
        dump("ewise_val", &addend_word, 1);
        add.run(len, addend);
        dump_vec("ewise_out", add.send_out_device_to_host());

        dump_vec("dense_mask", m);
        dump_vec("dense_inout_before", inout);
        for (int k = 1; k < 3; k++) {
            module::AssignVectorDenseModule<val_t> dense;
            dense.set_up_runtime("unused.xclbin");
            dense.set_mask_type(masks[k]);
            dense.send_mask_host_to_device(m);
            dense.send_inout_host_to_device(inout);
            dense.run(len, val_t(23));
            dump_vec("dense_inout_after_" + std::to_string(k), dense.send_inout_device_to_host());
        }
        const uint32_t w23 = VK::bits(val_t(23));
        dump("dense_val", &w23, 1);

        const uint32_t sn = 8192, cnt = 819;
        aligned_sparse_vec_t ms(cnt + 1);
        ms[0].index = cnt;
        ms[0].val = val_t(0);
        for (uint32_t i = 0; i < cnt; i++) {
            ms[i + 1].index = i * 10;
            ms[i + 1].val = val_t(rng() % 10);
        }
        aligned_dense_vec_t io1(sn), io2(sn);
        for (auto &v : io1) v = val_t(rng() % 10);
        for (auto &v : io2) v = (rng() % 10 > 5) ? val_t(5) : UFIXED_INF;
        dump_vec("sparse_mask", ms);
        dump_vec("sparse_io1_before", io1);
        dump_vec("sparse_io2_before", io2);
        module::AssignVectorSparseModule<val_t, idx_val_t> bfs_mode(false);
        bfs_mode.set_up_runtime("unused.xclbin");
        bfs_mode.send_mask_host_to_device(ms);
        bfs_mode.send_inout_host_to_device(io1);
        bfs_mode.run(val_t(3));
        const uint32_t w3 = VK::bits(val_t(3));
        dump("sparse_val", &w3, 1);
        dump_vec("sparse_io1_after", bfs_mode.send_inout_device_to_host());

        module::AssignVectorSparseModule<val_t, idx_val_t> sssp_mode(true);
        sssp_mode.set_up_runtime("unused.xclbin");
        sssp_mode.send_mask_host_to_device(ms);
        sssp_mode.send_inout_host_to_device(io2);
        sssp_mode.run();
        dump_vec("sparse_io2_after", sssp_mode.send_inout_device_to_host());
        dump_vec("sparse_new_frontier", sssp_mode.send_new_frontier_device_to_host());
    }
    printf("TYPED DRIVER DONE kind=%d\n", VK::kind);
    return 0;
}
