// graphlily/io/data_formatter.h -- the two host formatters the graph drivers use (reference
// io/data_formatter.h:18-51).  The FPGA layouts of the reference header (CPSR streams, tiled CSC
// packets, :54-721) are produced for CDNA4 inside gl_spmv_plan_create / gl_spmspv_plan_create instead.
#ifndef GRAPHLILY_IO_DATA_FORMATTER_H_
#define GRAPHLILY_IO_DATA_FORMATTER_H_

#include <cstdint>
#include <type_traits>
#include <vector>

#include "graphlily/global.h"
#include "graphlily/io/data_loader.h"

namespace graphlily {
namespace io {

// Pad rows (empty) and columns up to the next multiple of the divisors, in place.
template <typename data_type>
void util_round_csr_matrix_dim(CSRMatrix<data_type> &m, uint32_t row_divisor, uint32_t col_divisor) {
    if (m.num_rows % row_divisor != 0) {
        const uint32_t pad = row_divisor - m.num_rows % row_divisor;
        m.adj_indptr.insert(m.adj_indptr.end(), pad, m.adj_indptr[m.num_rows]);
        m.num_rows += pad;
    }
    if (m.num_cols % col_divisor != 0) m.num_cols += col_divisor - m.num_cols % col_divisor;
}

// adj_data[i] = 1.0 / (non-zeros in the column of i): double divide, stored as data_type.
template <typename data_type>
void util_normalize_csr_matrix_by_outdegree(CSRMatrix<data_type> &m) {
    if (std::is_same<data_type, float>::value) {   // natively (GPU when the runtime is up): same double divide, float store
        GRAPHLILY_CHECK(gl_csr_normalize_by_outdegree(m.num_rows, m.num_cols, m.adj_indptr.data(), m.adj_indices.data(),
                                                      reinterpret_cast<float *>(m.adj_data.data())));
        return;
    }
    std::vector<uint32_t> per_col(m.num_cols, 0);
    for (uint32_t c : m.adj_indices) per_col[c]++;
    const size_t nnz = m.adj_indptr[m.num_rows];
    for (size_t i = 0; i < nnz; i++) m.adj_data[i] = 1.0 / per_col[m.adj_indices[i]];
}

}  // namespace io
}  // namespace graphlily

#endif  // GRAPHLILY_IO_DATA_FORMATTER_H_
