// graphlily/io/data_loader.h -- CSR/CSC containers and the scipy-npz loader (reference
// io/data_loader.h:18-157).  The npz parsing is done by libgraphlily_hip.so (gl_npz_csr_*), which
// replaces the un-vendored cnpy dependency.
#ifndef GRAPHLILY_IO_DATA_LOADER_H_
#define GRAPHLILY_IO_DATA_LOADER_H_

#include <cassert>
#include <cstdint>
#include <iterator>
#include <string>
#include <vector>

#include "graphlily/global.h"

namespace graphlily {
namespace io {

template <typename data_type>
struct CSRMatrix {
    uint32_t num_rows;
    uint32_t num_cols;
    std::vector<data_type> adj_data;
    std::vector<uint32_t> adj_indices;
    std::vector<uint32_t> adj_indptr;
};

template <typename data_type>
CSRMatrix<data_type> create_csr_matrix(uint32_t num_rows, uint32_t num_cols,
                                       std::vector<data_type> const &adj_data,
                                       std::vector<uint32_t> const &adj_indices,
                                       std::vector<uint32_t> const &adj_indptr) {
    return CSRMatrix<data_type>{num_rows, num_cols, adj_data, adj_indices, adj_indptr};
}

// scipy.sparse.save_npz file with float32 data -> CSR (reference :51-70)
inline CSRMatrix<float> load_csr_matrix_from_float_npz(std::string csr_float_npz_path) {
    CSRMatrix<float> m;
    gl_npz_csr h = nullptr;
    uint64_t nnz = 0;
    GRAPHLILY_CHECK(gl_npz_csr_open(csr_float_npz_path.c_str(), &h, &m.num_rows, &m.num_cols, &nnz));
    m.adj_data.resize(nnz);
    m.adj_indices.resize(nnz);
    m.adj_indptr.resize((size_t)m.num_rows + 1);
    GRAPHLILY_CHECK(gl_npz_csr_read(h, m.adj_data.data(), m.adj_indices.data(), m.adj_indptr.data()));
    return m;
}

template <typename data_type>
CSRMatrix<data_type> csr_matrix_convert_from_float(CSRMatrix<float> const &in) {
    CSRMatrix<data_type> out;
    out.num_rows = in.num_rows;
    out.num_cols = in.num_cols;
    out.adj_data.assign(in.adj_data.begin(), in.adj_data.end());
    out.adj_indices = in.adj_indices;
    out.adj_indptr = in.adj_indptr;
    return out;
}

template <typename data_type>
struct CSCMatrix {
    uint32_t num_rows;
    uint32_t num_cols;
    std::vector<data_type> adj_data;
    std::vector<uint32_t> adj_indices;  // row ids
    std::vector<uint32_t> adj_indptr;   // num_cols + 1
};

// Transpose; rows inside a column stay ascending (reference :108-144).
template <typename data_type>
CSCMatrix<data_type> csr2csc(CSRMatrix<data_type> const &csr) {
    CSCMatrix<data_type> csc;
    csc.num_rows = csr.num_rows;
    csc.num_cols = csr.num_cols;
    const size_t nnz = csr.adj_indptr[csr.num_rows];
    csc.adj_data.resize(nnz);
    csc.adj_indices.resize(nnz);
    csc.adj_indptr.assign((size_t)csr.num_cols + 1, 0);
    for (size_t i = 0; i < nnz; i++) csc.adj_indptr[csr.adj_indices[i] + 1]++;
    for (size_t c = 0; c < csr.num_cols; c++) csc.adj_indptr[c + 1] += csc.adj_indptr[c];
    std::vector<uint32_t> cursor(csc.adj_indptr.begin(), csc.adj_indptr.end() - 1);
    for (uint32_t r = 0; r < csr.num_rows; r++) {
        for (size_t i = csr.adj_indptr[r]; i < csr.adj_indptr[r + 1]; i++) {
            const uint32_t dst = cursor[csr.adj_indices[i]]++;
            csc.adj_indices[dst] = r;
            csc.adj_data[dst] = csr.adj_data[i];
        }
    }
    return csc;
}

template <typename data_type>
CSCMatrix<data_type> csc_matrix_convert_from_float(CSCMatrix<float> const &in) {
    CSCMatrix<data_type> out;
    out.num_rows = in.num_rows;
    out.num_cols = in.num_cols;
    out.adj_data.assign(in.adj_data.begin(), in.adj_data.end());
    out.adj_indices = in.adj_indices;
    out.adj_indptr = in.adj_indptr;
    return out;
}

}  // namespace io
}  // namespace graphlily

#endif  // GRAPHLILY_IO_DATA_LOADER_H_
