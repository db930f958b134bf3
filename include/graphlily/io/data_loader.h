// graphlily/io/data_loader.h -- sparse-matrix containers, the scipy-npz loader and the CSR->CSC
// transpose of graphlily::io (reference io/data_loader.h:18-157), MI355X build.
//
// CSRMatrix<T> and CSCMatrix<T> are two instantiations of one compressed-storage aggregate; field names
// and order are the reference's, so brace / designated initialisation in caller code keeps working.
// npz parsing and the float transpose are done natively by libgraphlily_hip.so (gl_npz_csr_*,
// gl_csr2csc), replacing the un-vendored cnpy dependency and the single-threaded counting sort.
#ifndef GRAPHLILY_IO_DATA_LOADER_H_
#define GRAPHLILY_IO_DATA_LOADER_H_

#include <cstdint>
#include <string>
#include <type_traits>
#include <vector>

#include "graphlily/global.h"

namespace graphlily {
namespace io {

namespace detail {
struct by_row {};
struct by_col {};

// Compressed sparse storage.  For by_row: adj_indices are column ids, adj_indptr has num_rows + 1
// entries.  For by_col: adj_indices are row ids, adj_indptr has num_cols + 1 entries.
template <typename data_type, typename major>
struct Compressed {
    uint32_t num_rows;
    uint32_t num_cols;
    std::vector<data_type> adj_data;
    std::vector<uint32_t> adj_indices;
    std::vector<uint32_t> adj_indptr;
};

template <typename To, typename major, typename From>
Compressed<To, major> retype(Compressed<From, major> const &src) {
    Compressed<To, major> dst;
    dst.num_rows = src.num_rows;
    dst.num_cols = src.num_cols;
    dst.adj_data.reserve(src.adj_data.size());
    for (From v : src.adj_data) dst.adj_data.push_back(static_cast<To>(v));
    dst.adj_indices = src.adj_indices;
    dst.adj_indptr = src.adj_indptr;
    return dst;
}
}  // namespace detail

template <typename data_type>
using CSRMatrix = detail::Compressed<data_type, detail::by_row>;
template <typename data_type>
using CSCMatrix = detail::Compressed<data_type, detail::by_col>;

template <typename data_type>
CSRMatrix<data_type> create_csr_matrix(uint32_t num_rows, uint32_t num_cols,
                                       std::vector<data_type> const &adj_data,
                                       std::vector<uint32_t> const &adj_indices,
                                       std::vector<uint32_t> const &adj_indptr) {
    return {num_rows, num_cols, adj_data, adj_indices, adj_indptr};
}

// scipy.sparse.save_npz file (float data, 32- or 64-bit indices) -> CSR; reference :51-70
inline CSRMatrix<float> load_csr_matrix_from_float_npz(std::string csr_float_npz_path) {
    gl_npz_csr file = nullptr;
    uint32_t rows = 0, cols = 0;
    uint64_t nnz = 0;
    GRAPHLILY_CHECK(gl_npz_csr_open(csr_float_npz_path.c_str(), &file, &rows, &cols, &nnz));
    CSRMatrix<float> m{rows, cols, std::vector<float>(nnz), std::vector<uint32_t>(nnz),
                       std::vector<uint32_t>((size_t)rows + 1)};
    GRAPHLILY_CHECK(gl_npz_csr_read(file, m.adj_data.data(), m.adj_indices.data(), m.adj_indptr.data()));
    return m;
}

template <typename data_type>
CSRMatrix<data_type> csr_matrix_convert_from_float(CSRMatrix<float> const &in) {
    return detail::retype<data_type>(in);
}

template <typename data_type>
CSCMatrix<data_type> csc_matrix_convert_from_float(CSCMatrix<float> const &in) {
    return detail::retype<data_type>(in);
}

// Transpose.  Entries of one column come out in ascending row order (what the reference's row-by-row
// counting sort produces, :131-139).
template <typename data_type>
CSCMatrix<data_type> csr2csc(CSRMatrix<data_type> const &a) {
    const size_t nnz = a.adj_indptr[a.num_rows];
    CSCMatrix<data_type> t{a.num_rows, a.num_cols, std::vector<data_type>(nnz), std::vector<uint32_t>(nnz),
                           std::vector<uint32_t>((size_t)a.num_cols + 1, 0u)};
    if (std::is_same<data_type, float>::value) {
        GRAPHLILY_CHECK(gl_csr2csc(a.num_rows, a.num_cols, a.adj_indptr.data(), a.adj_indices.data(),
                                        reinterpret_cast<const float *>(a.adj_data.data()), t.adj_indptr.data(),
                                        t.adj_indices.data(), reinterpret_cast<float *>(t.adj_data.data())));
        return t;
    }
    // other value types: histogram of the column ids, prefix sum, then a scatter pass over the rows
    std::vector<uint32_t> &colptr = t.adj_indptr;
    for (size_t k = 0; k < nnz; k++) colptr[a.adj_indices[k] + 1]++;
    for (uint32_t c = 0; c < a.num_cols; c++) colptr[c + 1] += colptr[c];
    std::vector<uint32_t> next(colptr.begin(), colptr.end() - 1);
    for (uint32_t r = 0; r < a.num_rows; r++)
        for (uint32_t k = a.adj_indptr[r]; k < a.adj_indptr[r + 1]; k++) {
            const uint32_t at = next[a.adj_indices[k]]++;
            t.adj_indices[at] = r;
            t.adj_data[at] = a.adj_data[k];
        }
    return t;
}

}  // namespace io
}  // namespace graphlily

#endif  // GRAPHLILY_IO_DATA_LOADER_H_
