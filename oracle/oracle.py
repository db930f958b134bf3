"""ctypes/numpy front-end of the CPU oracle (oracle/graphlily_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Nothing under graphlily_amd/ may import it.

Function names follow the reference members they restate; see the C file for
the reference file:line of each.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgraphlily_oracle.so")

MULADD, ANDOR, ADDMIN = 0, 1, 2            # graphlily/global.h:83-87
NOMASK, WRITETOZERO, WRITETOONE = 0, 1, 2  # graphlily/global.h:103-107
FLOAT_INF = np.float32(999999999)          # graphlily/global.h:80
UFIXED_INF = np.float32(255)               # graphlily/global.h:79 (active TropicalSemiring zero, :99)

# graphlily/global.h:69 idx_val_t {idx_t index; val_t val;}
IDX_VAL = np.dtype([("index", np.uint32), ("val", np.float32)])

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_ivp = np.ctypeslib.ndpointer(dtype=IDX_VAL, flags="C_CONTIGUOUS")
# sparse element of the integer value types: {uint32 index; uint32 value word}
IDX_WORD = np.dtype([("index", np.uint32), ("val", np.uint32)])
_iwp = np.ctypeslib.ndpointer(dtype=IDX_WORD, flags="C_CONTIGUOUS")
VAL_FLOAT, VAL_UNSIGNED, VAL_UFIXED_32_8 = 0, 1, 2   # graphlily/global.h:62-64
_u32 = ctypes.c_uint32
_f32 = ctypes.c_float
_int = ctypes.c_int


def build():
    """Compile the C restatement (gcc, a second or two)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libgraphlily_oracle.so"])


def _load():
    if not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "graphlily_oracle.c"))):
        build()
    lib = ctypes.CDLL(_LIB_PATH)
    sig = {
        "orc_csr2csc": [_u32, _u32, _u32p, _u32p, _f32p, _u32p, _u32p, _f32p],
        "orc_normalize_by_outdegree": [_u32, _u32, _u32p, _u32p, _f32p],
        "orc_sssp_preprocess": [_u32, _u32p, _u32p, _u32p, _u32p, _f32p],
        "orc_sparse_to_dense": [_ivp, _u32, _f32, _f32p],
        "orc_spmv": [_int, _f32, _u32, _u32p, _u32p, _f32p, _f32p, _f32p],
        "orc_spmv_omp": [_int, _f32, _u32, _u32p, _u32p, _f32p, _f32p, _f32p],
        "orc_spmv_masked": [_int, _f32, _int, _u32, _u32p, _u32p, _f32p, _f32p, _f32p, _f32p],
        "orc_spmspv": [_int, _f32, _int, _u32, _u32p, _u32p, _f32p, _ivp, _f32p, _f32p],
        "orc_ewise_add": [_f32p, _u32, _f32, _f32p],
        "orc_assign_dense": [_int, _f32p, _f32p, _u32, _f32],
        "orc_assign_sparse": [_ivp, _f32p, _f32],
        "orc_assign_sparse_new_frontier": [_ivp, _f32p, _ivp],
        "orc_spmv_words": [_int, _int, _u32, _int, _u32, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p],
        "orc_spmspv_words": [_int, _int, _u32, _int, _u32, _u32p, _u32p, _u32p, _iwp, _u32p, _u32p],
        "orc_ewise_add_words": [_int, _u32p, _u32, _u32, _u32p],
        "orc_assign_dense_words": [_int, _u32p, _u32p, _u32, _u32],
        "orc_assign_sparse_new_frontier_words": [_iwp, _u32p, _iwp],
        "orc_bfs": [_u32, _u32p, _u32p, _f32p, _u32, _u32, _f32p],
        "orc_pagerank": [_u32, _u32p, _u32p, _f32p, _f32, _u32, _f32p],
        "orc_sssp": [_u32, _u32p, _u32p, _f32p, _f32, _u32, _u32, _f32p],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = None
    lib.orc_word_from_float.argtypes = [_int, _f32]
    lib.orc_word_from_float.restype = _u32
    lib.orc_word_to_float.argtypes = [_int, _u32]
    lib.orc_word_to_float.restype = _f32
    lib.orc_sssp_preprocess.restype = _u32
    lib.orc_assign_dense.restype = _int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


class CSR:
    """graphlily::io::CSRMatrix<float> (io/data_loader.h:18-30)."""

    def __init__(self, num_rows, num_cols, adj_data, adj_indices, adj_indptr):
        self.num_rows = int(num_rows)
        self.num_cols = int(num_cols)
        self.adj_data = np.ascontiguousarray(adj_data, dtype=np.float32)
        self.adj_indices = np.ascontiguousarray(adj_indices, dtype=np.uint32)
        self.adj_indptr = np.ascontiguousarray(adj_indptr, dtype=np.uint32)
        assert self.adj_indptr.shape[0] == self.num_rows + 1

    @property
    def nnz(self):
        return int(self.adj_indptr[self.num_rows])

    def copy(self):
        return CSR(self.num_rows, self.num_cols, self.adj_data.copy(),
                   self.adj_indices.copy(), self.adj_indptr.copy())


class CSC(CSR):
    """graphlily::io::CSCMatrix<float> (io/data_loader.h:92-104); adj_indptr has num_cols+1 entries."""

    def __init__(self, num_rows, num_cols, adj_data, adj_indices, adj_indptr):
        self.num_rows = int(num_rows)
        self.num_cols = int(num_cols)
        self.adj_data = np.ascontiguousarray(adj_data, dtype=np.float32)
        self.adj_indices = np.ascontiguousarray(adj_indices, dtype=np.uint32)
        self.adj_indptr = np.ascontiguousarray(adj_indptr, dtype=np.uint32)
        assert self.adj_indptr.shape[0] == self.num_cols + 1

    @property
    def nnz(self):
        return int(self.adj_indptr[self.num_cols])


def load_csr_matrix_from_float_npz(path):
    """io/data_loader.h:51-70.  The reference reads `shape` as uint32 words [0] and [2] of an
    int64 pair and indices/indptr as 32-bit words; numpy does the same decoding here."""
    z = np.load(path)
    shape = z["shape"].astype(np.int64).view(np.uint32)
    num_rows, num_cols = int(shape[0]), int(shape[2])
    data = z["data"].astype(np.float32)
    nnz = data.shape[0]
    indices = z["indices"].view(np.uint32)[:nnz]
    indptr = z["indptr"].view(np.uint32)[:num_rows + 1]
    return CSR(num_rows, num_cols, data, indices, indptr)


def csr2csc(csr):
    nnz = csr.nnz
    indptr = np.zeros(csr.num_cols + 1, dtype=np.uint32)
    indices = np.zeros(nnz, dtype=np.uint32)
    data = np.zeros(nnz, dtype=np.float32)
    lib().orc_csr2csc(csr.num_rows, csr.num_cols, csr.adj_indptr, csr.adj_indices, csr.adj_data,
                      indptr, indices, data)
    return CSC(csr.num_rows, csr.num_cols, data, indices, indptr)


def util_round_csr_matrix_dim(csr, row_divisor, col_divisor):
    """io/data_formatter.h:18-33 (in place, like the reference)."""
    if csr.num_rows % row_divisor != 0:
        pad = row_divisor - csr.num_rows % row_divisor
        csr.adj_indptr = np.concatenate(
            [csr.adj_indptr, np.full(pad, csr.adj_indptr[csr.num_rows], dtype=np.uint32)])
        csr.num_rows += pad
    if csr.num_cols % col_divisor != 0:
        csr.num_cols += col_divisor - csr.num_cols % col_divisor


def util_normalize_csr_matrix_by_outdegree(csr):
    lib().orc_normalize_by_outdegree(csr.num_rows, csr.num_cols, csr.adj_indptr, csr.adj_indices,
                                     csr.adj_data)


def sssp_preprocess(csr):
    """app/sssp.h:16-62 (in place)."""
    n = csr.adj_indptr.shape[0] - 1
    nnz = int(csr.adj_indptr[n])
    out_indptr = np.zeros(n + 1, dtype=np.uint32)
    out_indices = np.zeros(nnz + n, dtype=np.uint32)
    out_data = np.zeros(nnz + n, dtype=np.float32)
    new_nnz = lib().orc_sssp_preprocess(n, csr.adj_indptr, csr.adj_indices,
                                        out_indptr, out_indices, out_data)
    csr.adj_indptr = out_indptr
    csr.adj_indices = np.ascontiguousarray(out_indices[:new_nnz])
    csr.adj_data = np.ascontiguousarray(out_data[:new_nnz])


def make_sparse_vec(indices, vals, head_val=0.0):
    """[0] = {nnz, head_val}, payload in [1..nnz] (module/spmspv_module.h:53-60)."""
    indices = np.asarray(indices)
    v = np.zeros(len(indices) + 1, dtype=IDX_VAL)
    v["index"][0] = len(indices)
    v["val"][0] = head_val
    v["index"][1:] = indices
    v["val"][1:] = vals
    return v


def convert_sparse_vec_to_dense_vec(sv, rng, zero):
    out = np.zeros(rng, dtype=np.float32)
    lib().orc_sparse_to_dense(np.ascontiguousarray(sv, dtype=IDX_VAL), rng, float(zero), out)
    return out


def _f32a(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def spmv(csr, x, op, zero, mask=None, mask_type=NOMASK, omp=False):
    """SpMVModule::compute_reference_results (module/spmv_module.h:478-532).
    With mask=None this is the unmasked overload; with a mask the masked overload (which treats
    every non-WriteToZero mask_type as WriteToOne, :524-530)."""
    y = np.zeros(csr.num_rows, dtype=np.float32)
    x = _f32a(x)
    assert x.shape[0] >= csr.num_cols
    if mask is None:
        fn = lib().orc_spmv_omp if omp else lib().orc_spmv
        fn(op, float(zero), csr.num_rows, csr.adj_indptr, csr.adj_indices, csr.adj_data, x, y)
    else:
        lib().orc_spmv_masked(op, float(zero), mask_type, csr.num_rows, csr.adj_indptr,
                              csr.adj_indices, csr.adj_data, x, _f32a(mask), y)
    return y


def spmspv(csc, v, op, zero, mask=None, mask_type=NOMASK):
    """SpMSpVModule::compute_reference_results (module/spmspv_module.h:445-520); dense result."""
    y = np.zeros(csc.num_rows, dtype=np.float32)
    if mask is None:
        mask = np.zeros(csc.num_rows, dtype=np.float32)
    lib().orc_spmspv(op, float(zero), mask_type, csc.num_rows, csc.adj_indptr, csc.adj_indices,
                     csc.adj_data, np.ascontiguousarray(v, dtype=IDX_VAL), _f32a(mask), y)
    return y


def ewise_add(inp, length, val):
    out = np.zeros(length, dtype=np.float32)
    lib().orc_ewise_add(_f32a(inp), length, float(val), out)
    return out


def assign_dense(mask_type, mask, inout, length, val):
    """In place on `inout` (float32, contiguous); raises on kNoMask like the reference exits."""
    assert inout.dtype == np.float32 and inout.flags["C_CONTIGUOUS"]
    rc = lib().orc_assign_dense(mask_type, _f32a(mask), inout, length, float(val))
    if rc != 0:
        raise ValueError("Invalid mask type")


def assign_sparse(mask_sv, inout, val):
    assert inout.dtype == np.float32 and inout.flags["C_CONTIGUOUS"]
    lib().orc_assign_sparse(np.ascontiguousarray(mask_sv, dtype=IDX_VAL), inout, float(val))


def assign_sparse_new_frontier(mask_sv, inout):
    assert inout.dtype == np.float32 and inout.flags["C_CONTIGUOUS"]
    mask_sv = np.ascontiguousarray(mask_sv, dtype=IDX_VAL)
    nf = np.zeros(int(mask_sv["index"][0]) + 1, dtype=IDX_VAL)
    lib().orc_assign_sparse_new_frontier(mask_sv, inout, nf)
    return nf[:int(nf["index"][0]) + 1]


def bfs(csr, source, num_iterations):
    d = np.zeros(csr.num_rows, dtype=np.float32)
    lib().orc_bfs(csr.num_rows, csr.adj_indptr, csr.adj_indices, csr.adj_data, source,
                  num_iterations, d)
    return d


def pagerank(csr, damping, num_iterations):
    r = np.zeros(csr.num_rows, dtype=np.float32)
    lib().orc_pagerank(csr.num_rows, csr.adj_indptr, csr.adj_indices, csr.adj_data,
                       float(damping), num_iterations, r)
    return r


def sssp(csr, source, num_iterations, zero=UFIXED_INF):
    d = np.zeros(csr.num_rows, dtype=np.float32)
    lib().orc_sssp(csr.num_rows, csr.adj_indptr, csr.adj_indices, csr.adj_data, float(zero),
                   source, num_iterations, d)
    return d


# ------------------------------------------------------------------ the reference's integer value types (words)
def _u32a(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def words_from_float(vt, values):
    """float -> value words, the conversion of csr_matrix_convert_from_float<val_t> (io/data_loader.h:75-84)."""
    f = lib().orc_word_from_float
    return np.array([f(vt, float(v)) for v in np.asarray(values, dtype=np.float32).ravel()], dtype=np.uint32)


def spmv_words(csr_indptr, csr_indices, data_words, x_words, op, vt, zero_word, mask=None, mask_type=NOMASK):
    n = len(csr_indptr) - 1
    y = np.zeros(n, dtype=np.uint32)
    m = _u32a(mask) if mask is not None else np.zeros(n, dtype=np.uint32)
    lib().orc_spmv_words(op, vt, int(zero_word), mask_type if mask is not None else NOMASK, n, _u32a(csr_indptr),
                         _u32a(csr_indices), _u32a(data_words), _u32a(x_words), m, y)
    return y


def spmspv_words(csc_indptr, csc_indices, data_words, v, num_rows, op, vt, zero_word, mask=None, mask_type=NOMASK):
    y = np.zeros(num_rows, dtype=np.uint32)
    m = _u32a(mask) if mask is not None else np.zeros(num_rows, dtype=np.uint32)
    lib().orc_spmspv_words(op, vt, int(zero_word), mask_type if mask is not None else NOMASK, num_rows, _u32a(csc_indptr),
                           _u32a(csc_indices), _u32a(data_words), np.ascontiguousarray(v, dtype=IDX_WORD), m, y)
    return y


def ewise_add_words(vt, inp, val_word):
    inp = _u32a(inp)
    out = np.zeros_like(inp)
    lib().orc_ewise_add_words(vt, inp, inp.shape[0], int(val_word), out)
    return out


def assign_dense_words(mask_type, mask, inout, val_word):
    assert inout.dtype == np.uint32 and inout.flags["C_CONTIGUOUS"]
    lib().orc_assign_dense_words(mask_type, _u32a(mask), inout, inout.shape[0], int(val_word))


def assign_sparse_new_frontier_words(mask_sv, inout):
    assert inout.dtype == np.uint32 and inout.flags["C_CONTIGUOUS"]
    mask_sv = np.ascontiguousarray(mask_sv, dtype=IDX_WORD)
    nf = np.zeros(int(mask_sv["index"][0]) + 1, dtype=IDX_WORD)
    lib().orc_assign_sparse_new_frontier_words(mask_sv, inout, nf)
    return nf[:int(nf["index"][0]) + 1]
