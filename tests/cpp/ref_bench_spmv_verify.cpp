// The reference's benchmark/bench_spmv.cpp leaves its own `verify` template (:15-33) uncalled (:91-92 are
// commented out).  This translation unit -- built in THIS container only, where /root/reference exists --
// includes that file UNMODIFIED (its `main` renamed by the preprocessor), so that the reference's `verify`
// and its set-up protocol (:50-64: adj_data = 1/num_rows, rows padded to num_channels*pack_size, x and mask
// rand()%2) are the ones compiled, and calls `verify<val_t>` on the HIP backend's results.
//   g++ -std=c++11 -I<repo>/include -I/root/reference tests/cpp/ref_bench_spmv_verify.cpp -lgraphlily_hip
#define main reference_bench_spmv_main
#include "benchmark/bench_spmv.cpp"
#undef main

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s graph.npz\n", argv[0]); return 2; }
    uint32_t num_channels = 16;
    graphlily::module::SpMVModule<graphlily::val_t, graphlily::val_t> spmv(num_channels, 1024000, 30720);
    spmv.set_target("hw");
    spmv.set_mask_type(graphlily::kNoMask);
    spmv.set_semiring(graphlily::ArithmeticSemiring);
    spmv.set_up_runtime("unused.xclbin");
    CSRMatrix<float> csr_matrix = graphlily::io::load_csr_matrix_from_float_npz(argv[1]);
    for (auto &x : csr_matrix.adj_data) x = 1.0 / csr_matrix.num_rows;
    graphlily::io::util_round_csr_matrix_dim(csr_matrix, num_channels * graphlily::pack_size, graphlily::pack_size);
    std::vector<float, aligned_allocator<float>> vector_float(csr_matrix.num_cols);
    std::generate(vector_float.begin(), vector_float.end(), [&] { return float(rand() % 2); });
    std::vector<graphlily::val_t, aligned_allocator<graphlily::val_t>> vector(vector_float.begin(), vector_float.end());
    spmv.load_and_format_matrix(csr_matrix, true);
    spmv.send_matrix_host_to_device();
    spmv.send_vector_host_to_device(vector);
    spmv.run();
    auto kernel_results = spmv.send_results_device_to_host();
    auto reference_results = spmv.compute_reference_results(vector_float);
    verify<graphlily::val_t>(reference_results, kernel_results);    // exits on a mismatch (bench_spmv.cpp:23-31)
    std::cout << "SpMV passed" << std::endl;                        // the line bench_spmv.cpp:92 would print
    return 0;
}
