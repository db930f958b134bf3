"""CPU stand-in for graphlily_amd.app.HipBackend, used ONLY by the gloo tests: the same module API,
every run() evaluated by the oracle on the owned row shard, buffers as CPU torch tensors.  It lets the
world_size-2 tests exercise the drivers' distributed control flow (sharding, slice ops, all-gathers)
without a GPU.  It is test infrastructure like the oracle itself."""
import numpy as np
import torch

from graphlily_amd import module as M
from graphlily_amd.capi import IDX_VAL
from oracle import oracle as O


class CpuBuf:
    def __init__(self, tensor):
        self.tensor = tensor
        self.nbytes = tensor.numel() * tensor.element_size()

    def np(self, dtype=None):
        a = self.tensor.numpy()
        return a.view(IDX_VAL) if a.dtype == np.int64 else a


class _Mod:
    blocking = False

    def set_row_shard(self, r0, r1):
        self.r0, self.r1 = r0, r1

    def set_semiring(self, s):
        self.semiring_ = s

    def set_mask_type(self, m):
        self.mask_type_ = m


def _sub(csr, r0, r1, cls=O.CSR):
    ip = csr.adj_indptr.astype(np.int64)
    return cls(r1 - r0, csr.num_cols, csr.adj_data[ip[r0]:ip[r1]], csr.adj_indices[ip[r0]:ip[r1]],
               (ip[r0:r1 + 1] - ip[r0]).astype(np.uint32))


class SpMVModule(_Mod):
    def __init__(self, *a):
        self.mask_type_, self.r0, self.r1 = M.kNoMask, 0, None

    def load_and_format_matrix(self, csr, skip):
        self.csr = csr

    def send_matrix_host_to_device(self):
        self.r1 = self.csr.num_rows if self.r1 is None else self.r1
        self.sub = _sub(self.csr, self.r0, self.r1)

    def get_num_rows(self):
        return self.csr.num_rows

    def get_num_cols(self):
        return self.csr.num_cols

    def get_nnz(self):
        return self.csr.nnz

    def bind_vector_buf(self, b):
        self.vector_buf = b

    def bind_mask_buf(self, b):
        self.mask_buf = b

    def bind_results_buf(self, b):
        self.results_buf = b

    def run(self):
        s = self.semiring_
        x = self.vector_buf.np()
        # (+,x) with a non-zero `zero` (PageRank passes its teleport term there): the device adds it to the finished
        # row sum -- zero + sum, one float add, like the reference's eWiseAdd after the SpMV -- not in front of it
        late = s.op == M.kMulAdd and s.zero != 0.0
        zero = 0.0 if late else s.zero
        if self.mask_type_ == M.kNoMask:
            y = O.spmv(self.sub, x, s.op, zero)
            allowed = np.ones(y.shape[0], bool)
        else:
            mask = self.mask_buf.np()[self.r0:self.r1].copy()
            y = O.spmv(self.sub, x, s.op, zero, mask, self.mask_type_)
            allowed = (mask == 0) if self.mask_type_ == M.kMaskWriteToZero else (mask != 0)
        if late:
            y = np.where(allowed, np.float32(s.zero) + y, np.float32(0)).astype(np.float32)
        self.results_buf.np()[self.r0:self.r1] = y


class SpMSpVModule(_Mod):
    def __init__(self, *a):
        self.mask_type_, self.r0, self.r1 = M.kNoMask, 0, None

    def load_and_format_matrix(self, csc):
        self.csc = csc

    def send_matrix_host_to_device(self):
        self.r1 = self.csc.num_rows if self.r1 is None else self.r1
        self.ocsc = O.CSC(self.csc.num_rows, self.csc.num_cols, self.csc.adj_data, self.csc.adj_indices,
                          self.csc.adj_indptr)
        self.results_buf = None

    def get_num_rows(self):
        return self.csc.num_rows

    def bind_vector_buf(self, b):
        self.vector_buf = b

    def bind_mask_buf(self, b):
        self.mask_buf = b

    def run(self):
        s = self.semiring_
        v = self.vector_buf.np()
        dense = O.spmspv(self.ocsc, v[:int(v["index"][0]) + 1].copy(), s.op, s.zero, self.mask_buf.np().copy(),
                         self.mask_type_)
        idx = np.nonzero(dense[self.r0:self.r1] != np.float32(s.zero))[0] + self.r0
        out = self.results_buf.np()
        out["index"][0], out["val"][0] = idx.size, s.zero
        out["index"][1:idx.size + 1] = idx
        out["val"][1:idx.size + 1] = dense[idx]

    def get_results_nnz(self):
        return int(self.results_buf.np()["index"][0])


class eWiseAddModule(_Mod):
    def bind_in_buf(self, b):
        self.in_buf = b

    def bind_out_buf(self, b):
        self.out_buf = b

    def run(self, length, val):
        self.out_buf.np()[:length] = O.ewise_add(self.in_buf.np()[:length].copy(), length, val)


class AssignVectorDenseModule(_Mod):
    def bind_mask_buf(self, b):
        self.mask_buf = b

    def bind_inout_buf(self, b):
        self.inout_buf = b

    def run(self, length, val):
        io_ = self.inout_buf.np()[:length].copy()
        O.assign_dense(self.mask_type_, self.mask_buf.np()[:length].copy(), io_, length, val)
        self.inout_buf.np()[:length] = io_


class AssignVectorSparseModule(_Mod):
    def __init__(self, generate_new_frontier):
        self.gen = generate_new_frontier

    def bind_mask_buf(self, b):
        self.mask_buf = b

    def bind_inout_buf(self, b):
        self.inout_buf = b

    def bind_new_frontier_buf(self, b):
        self.new_frontier_buf = b

    def run(self, val=None):
        m = self.mask_buf.np()
        m = m[:int(m["index"][0]) + 1].copy()
        io_ = self.inout_buf.np().copy()
        if val is None:
            nf = O.assign_sparse_new_frontier(m, io_)
            self.new_frontier_buf.np()[:nf.shape[0]] = nf
        else:
            O.assign_sparse(m, io_, val)
        self.inout_buf.np()[:] = io_


class CpuBackend:
    SpMVModule, SpMSpVModule, eWiseAddModule = SpMVModule, SpMSpVModule, eWiseAddModule
    AssignVectorDenseModule, AssignVectorSparseModule = AssignVectorDenseModule, AssignVectorSparseModule

    def init(self):
        pass

    def alloc(self, count, dtype):
        return CpuBuf(torch.zeros(count, dtype=torch.float32 if np.dtype(dtype).itemsize == 4 else torch.int64))

    def view(self, buf, first, count, itemsize):
        return CpuBuf(buf.tensor[first:first + count])

    def upload(self, buf, arr):
        arr = np.ascontiguousarray(arr)
        src = arr.view(np.int64) if arr.dtype == IDX_VAL else arr
        buf.tensor[:src.shape[0]].copy_(torch.from_numpy(src.copy()))

    def download(self, buf, dtype, count):
        return buf.np()[:count].copy()

    def download_result(self, buf, count):
        return self.download(buf, np.float32, count)

    def copy(self, dst, src, nbytes):
        k = nbytes // dst.tensor.element_size()
        dst.tensor[:k].copy_(src.tensor[:k])

    def fill(self, buf, value, count):
        buf.tensor[:count] = float(value)

    def sparse_to_dense(self, sparse, dense, rng, zero, max_entries):
        dense.np()[:rng] = O.convert_sparse_vec_to_dense_vec(sparse.np().copy(), rng, zero)

    def sync(self):
        pass
