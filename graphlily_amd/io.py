"""Host-side matrix containers and formatters of the graphlily::io namespace.

Mirrors (names, argument meaning, in-place behaviour):
  CSRMatrix / CSCMatrix / create_csr_matrix      io/data_loader.h:18-47, 92-104
  load_csr_matrix_from_float_npz                 io/data_loader.h:51-70   (npz parsing done natively by
                                                 gl_npz_csr_* in libgraphlily_hip.so, replacing cnpy)
  csr2csc                                        io/data_loader.h:108-144
  util_round_csr_matrix_dim                      io/data_formatter.h:18-33
  util_normalize_csr_matrix_by_outdegree         io/data_formatter.h:36-51
The FPGA-only formatters (csr2cpsr, formatCSC: io/data_formatter.h:54-721) have no counterpart here:
the device layout is produced inside gl_spmv_plan_create / gl_spmspv_plan_create.
"""
import numpy as np

from . import capi


class CSRMatrix:
    """num_rows, num_cols, adj_data[nnz] f32, adj_indices[nnz] u32, adj_indptr[num_rows+1] u32."""

    def __init__(self, num_rows=0, num_cols=0, adj_data=(), adj_indices=(), adj_indptr=(0,)):
        self.num_rows = int(num_rows)
        self.num_cols = int(num_cols)
        self.adj_data = np.ascontiguousarray(adj_data, dtype=np.float32)
        self.adj_indices = np.ascontiguousarray(adj_indices, dtype=np.uint32)
        self.adj_indptr = np.ascontiguousarray(adj_indptr, dtype=np.uint32)

    @property
    def nnz(self):
        return int(self.adj_indptr[-1])

    def copy(self):
        return type(self)(self.num_rows, self.num_cols, self.adj_data.copy(), self.adj_indices.copy(),
                          self.adj_indptr.copy())


class CSCMatrix(CSRMatrix):
    """Same fields; adj_indices are row ids and adj_indptr has num_cols+1 entries."""


def create_csr_matrix(num_rows, num_cols, adj_data, adj_indices, adj_indptr):
    return CSRMatrix(num_rows, num_cols, adj_data, adj_indices, adj_indptr)


def load_csr_matrix_from_float_npz(csr_float_npz_path):
    nr, nc, data, indices, indptr = capi.npz_load_csr(csr_float_npz_path)
    return CSRMatrix(nr, nc, data, indices, indptr)


def csr2csc(csr_matrix):
    """Transpose (io/data_loader.h:108-144); rows inside a column stay ascending.  Done natively
    (gl_csr2csc: a radix sort on the GPU for large matrices, a parallel counting sort on the host otherwise): the numpy formulation took 18 s on 212 M non-zeros."""
    indptr, indices, data = capi.host_csr2csc(csr_matrix.num_rows, csr_matrix.num_cols, csr_matrix.adj_indptr,
                                              csr_matrix.adj_indices, csr_matrix.adj_data)
    return CSCMatrix(csr_matrix.num_rows, csr_matrix.num_cols, data, indices, indptr)


def util_round_csr_matrix_dim(csr_matrix, row_divisor, col_divisor):
    """Pads in place: extra rows are empty (indptr repeats its last value), extra columns just widen."""
    rem = csr_matrix.num_rows % row_divisor
    if rem:
        pad = row_divisor - rem
        last = csr_matrix.adj_indptr[csr_matrix.num_rows]
        csr_matrix.adj_indptr = np.concatenate([csr_matrix.adj_indptr, np.full(pad, last, dtype=np.uint32)])
        csr_matrix.num_rows += pad
    rem = csr_matrix.num_cols % col_divisor
    if rem:
        csr_matrix.num_cols += col_divisor - rem


def util_normalize_csr_matrix_by_outdegree(csr_matrix):
    """adj_data[i] = 1.0 / (#non-zeros in column of i): double divide, float store."""
    csr_matrix.adj_data = capi.csr_normalize_by_outdegree(csr_matrix.num_rows, csr_matrix.num_cols, csr_matrix.adj_indptr,
                                                          csr_matrix.adj_indices)


def sssp_add_self_edges(csr_matrix):
    """The matrix preparation of the reference's SSSP app (app/sssp.h:16-62): all weights become 1 and
    weight-0 self edges are added so that a (min,+) SpMV keeps the previous distance.

    The reference edits the CSR arrays in place while walking the rows and reads the end of each
    row from the not-yet-shifted indptr, so after k insertions only the first (len - k) entries of a
    row are examined and rows no longer than k receive no self edge at all.  Results must match the
    reference, so that behaviour is reproduced here rather than "fixed".
    """
    n = csr_matrix.adj_indptr.shape[0] - 1
    indptr = csr_matrix.adj_indptr.astype(np.int64)
    indices = csr_matrix.adj_indices
    lens = np.diff(indptr)
    # rows at or after `r` can only be touched while k <= their length
    suffix_max = np.maximum.accumulate(lens[::-1])[::-1] if n else lens
    insert_at = np.full(n, -1, dtype=np.int64)   # position inside the row where the self edge goes
    zero_at = np.full(n, -1, dtype=np.int64)     # position of an existing diagonal entry to zero
    k = 0
    for r in range(n):
        if k > suffix_max[r]:
            break
        win = int(lens[r]) - k
        if win == 0:
            insert_at[r] = 0
            k += 1
        elif win > 0:
            row = indices[indptr[r]:indptr[r] + win]
            hit = np.nonzero(row >= r)[0]
            if hit.size and row[hit[0]] == r:
                zero_at[r] = hit[0]
            else:
                insert_at[r] = hit[0] if hit.size else win - 1
                k += 1
    has_ins = insert_at >= 0
    new_lens = lens + has_ins
    new_indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(new_lens, out=new_indptr[1:])
    total = int(new_indptr[n])
    out_idx = np.empty(total, dtype=np.uint32)
    out_val = np.ones(total, dtype=np.float32)
    # destination of every original entry: shifted by one inside rows after the insertion point
    row_of = np.repeat(np.arange(n, dtype=np.int64), lens)
    pos_in_row = np.arange(int(indptr[n]), dtype=np.int64) - indptr[row_of]
    shift = (has_ins[row_of] & (pos_in_row >= insert_at[row_of])).astype(np.int64)
    dest = new_indptr[row_of] + pos_in_row + shift
    out_idx[dest] = indices[:int(indptr[n])]
    rows_ins = np.nonzero(has_ins)[0]
    self_pos = new_indptr[rows_ins] + insert_at[rows_ins]
    out_idx[self_pos] = rows_ins.astype(np.uint32)
    out_val[self_pos] = 0.0
    rows_zero = np.nonzero(zero_at >= 0)[0]
    out_val[new_indptr[rows_zero] + zero_at[rows_zero]] = 0.0
    csr_matrix.adj_indptr = new_indptr.astype(np.uint32)
    csr_matrix.adj_indices = out_idx
    csr_matrix.adj_data = out_val
