"""Python mirror of the graphlily::module operator API on top of the HIP C ABI.

Same class names, method names, argument meaning and blocking behaviour as the reference's
header-only modules, so parity tests read like the reference's tests/test_module_*.cpp:

  BaseModule                  module/base_module.h:10-133
  SpMVModule                  module/spmv_module.h:37-532
  SpMSpVModule                module/spmspv_module.h:27-520
  eWiseAddModule              module/add_scalar_vector_dense_module.h:19-204
  AssignVectorDenseModule     module/assign_vector_dense_module.h:19-246
  AssignVectorSparseModule    module/assign_vector_sparse_module.h:19-335

The C++ twin (the actual drop-in for C++ callers) lives in include/graphlily/module/*.h and calls
the same entry points.  Device buffers are capi.DeviceBuffer handles and play the role of the
public cl::Buffer members (`vector_buf`, `mask_buf`, ...), including being shareable via bind_*.
`compute_reference_results` is deliberately absent: the CPU reference is test infrastructure
(oracle/) and the product never routes through it.
"""
import collections
import os
import sys

import numpy as np

from . import capi
from .capi import IDX_VAL

# ---------------------------------------------------------------- graphlily/global.h
kMulAdd, kLogicalAndOr, kAddMin = 0, 1, 2                    # :83-87
kNoMask, kMaskWriteToZero, kMaskWriteToOne = 0, 1, 2         # :103-107
SemiringType = collections.namedtuple("SemiringType", ["op", "one", "zero"])  # :90-94
UFIXED_INF = 255.0                                           # :79
FLOAT_INF = 999999999.0                                      # :80
ArithmeticSemiring = SemiringType(kMulAdd, 1.0, 0.0)         # :96
LogicalSemiring = SemiringType(kLogicalAndOr, 1.0, 0.0)      # :97
# :99-100 -- the reference ships the ap_ufixed line (zero = 255); with val_t = float the FLOAT_INF
# line is the one to enable.  `zero` is a runtime field here, so either works.
TropicalSemiring = SemiringType(kAddMin, 0.0, FLOAT_INF)
TropicalSemiringUfixed = SemiringType(kAddMin, 0.0, UFIXED_INF)
pack_size = 8                                                # :57
num_hbm_channels = 16                                        # :59
idx_marker = 0xFFFFFFFF                                      # :66


def _fatal(msg):
    """The reference reports errors by printing and exit(EXIT_FAILURE) (xcl2.hpp:40-46)."""
    print(msg, file=sys.stderr)
    raise SystemExit(1)


def make_sparse_vec(indices, vals, head_val=0.0):
    """aligned_sparse_vec_t with the count in [0].index (module/spmspv_module.h:53-60)."""
    indices = np.asarray(indices)
    v = np.zeros(len(indices) + 1, dtype=IDX_VAL)
    v["index"][0] = len(indices)
    v["val"][0] = head_val
    v["index"][1:] = indices
    v["val"][1:] = vals
    return v


def convert_sparse_vec_to_dense_vec(sparse_vector, rng, zero):
    """graphlily/global.h:153-164"""
    dense = np.full(rng, zero, dtype=np.float32)
    nnz = int(sparse_vector["index"][0])
    dense[sparse_vector["index"][1:nnz + 1]] = sparse_vector["val"][1:nnz + 1]
    return dense


class BaseModule:
    # Counts the module calls that write device memory (any module's run / send_* / copy).  An SpMSpVModule remembers the
    # value right after it uploaded a tiny vector; while nothing else has written since, its runs tell the library that
    # the vector is still that tiny one (gl_spmspv_plan_hint_work: one launch instead of two).
    device_writes_ = 0

    def __init__(self, kernel_name="overlay"):
        self.kernel_name_ = kernel_name
        self.target_ = "hw"
        self.device_ = 0
        self.blocking = True  # every reference call ends in command_queue_.finish()

    def get_kernel_name(self):
        return self.kernel_name_

    def set_target(self, target):
        assert target in ("sw_emu", "hw_emu", "hw")  # base_module.h:74-77
        self.target_ = target

    def set_device(self, device):
        self.device_ = int(device)

    def set_up_runtime(self, xclbin_file_path=None):
        """No bitstream on a GPU: the path is accepted and ignored; this selects the device."""
        capi.init(self.device_)

    def copy_buffer_device_to_device(self, src, dst, nbytes):
        BaseModule._wrote(dst)
        capi.copy_d2d(dst, src, nbytes)
        capi.sync()

    @staticmethod
    def _wrote(*buffers):
        """One more module call has written device memory; `buffers` are the ones it wrote (each remembers the call's number:
        SpMSpVModule.get_results_nnz trusts the operator's completion record only while its run is the last writer of
        results_buf)."""
        BaseModule.device_writes_ += 1
        for b in buffers:
            if b is not None:
                b.last_write_ = BaseModule.device_writes_

    def _finish(self, *written):
        BaseModule._wrote(*written)       # every run / send_* of every module ends here
        if self.blocking:
            capi.sync()


class SpMVModule(BaseModule):
    def __init__(self, num_channels=num_hbm_channels, out_buf_len=0, vec_buf_len=0):
        super().__init__()
        # FPGA buffer geometry; kept for signature parity, the GPU tiling is chosen by the plan
        self.num_channels_, self.out_buf_len_, self.vec_buf_len_ = num_channels, out_buf_len, vec_buf_len
        self.mask_type_ = kNoMask
        self.semiring_ = ArithmeticSemiring
        self.csr_matrix_ = None
        self.plan_ = None
        self.row_begin_, self.row_end_ = 0, None
        self.vector_buf = self.mask_buf = self.results_buf = None
        self.plan_flags_ = 0

    def set_semiring(self, semiring):
        self.semiring_ = semiring

    def set_mask_type(self, mask_type):
        self.mask_type_ = mask_type

    def set_row_shard(self, row_begin, row_end):
        """Multi-GPU extension: this device owns rows [row_begin, row_end) of the matrix."""
        self.row_begin_, self.row_end_ = int(row_begin), int(row_end)

    def set_plan_flags(self, flags):
        """Extension: GL_PLAN_* flags for the plans this module creates (before send_matrix_host_to_device), e.g.
        capi.GL_PLAN_REFERENCE_ORDER -- the diagnostic layout that evaluates compute_reference_results' own loop
        (module/spmv_module.h:478-510) on the device -- or capi.GL_PLAN_KEEP_VALUES."""
        self.plan_flags_ = int(flags)

    def get_num_rows(self):
        return self.csr_matrix_.num_rows

    def get_num_cols(self):
        return self.csr_matrix_.num_cols

    def get_nnz(self):
        return int(self.csr_matrix_.adj_indptr[self.csr_matrix_.num_rows])

    def load_and_format_matrix(self, csr_matrix_float, skip_empty_rows=True):
        # skip_empty_rows only changes the FPGA stream encoding; results are identical either way
        self.csr_matrix_ = csr_matrix_float
        self.skip_empty_rows_ = skip_empty_rows

    @staticmethod
    def _plan_flags(op):
        if op == kLogicalAndOr:     # pattern-only entries, x as bits (gl_spmv_bool.hip)
            return capi.GL_PLAN_BOOLEAN | capi.GL_PLAN_NO_MULADD
        return capi.GL_PLAN_NO_MULADD if op != kMulAdd else 0

    def _plan_serves(self, op):
        f = self.plan_.flags
        if f & capi.GL_PLAN_REFERENCE_ORDER:
            return True
        if (f & capi.GL_PLAN_BOOLEAN) and op != kLogicalAndOr:
            return False
        return not ((f & capi.GL_PLAN_NO_MULADD) and op == kMulAdd)

    def _make_plan(self):
        m = self.csr_matrix_
        re = m.num_rows if self.row_end_ is None else self.row_end_
        # the semiring known at upload time sizes the LDS split (accumulators vs hot-column table); a later
        # switch to (+,x) on a plan built for the 4-byte semirings re-formats the matrix (see run())
        flags = self._plan_flags(self.semiring_.op) | self.plan_flags_
        if os.environ.get("GRAPHLILY_SPMV_ORDER") == "reference":   # the C++ module layer's switch (spmv_module.h), mirrored
            flags |= capi.GL_PLAN_REFERENCE_ORDER
            self.plan_flags_ |= capi.GL_PLAN_REFERENCE_ORDER
        if flags & capi.GL_PLAN_REFERENCE_ORDER:
            flags = capi.GL_PLAN_REFERENCE_ORDER
        for client in getattr(self, "pull_clients_", ()):   # the old plan is about to go away
            client._attach(None)
        self.plan_ = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data,
                                   self.row_begin_, re, flags)
        for client in getattr(self, "pull_clients_", ()):
            client._attach(self)

    def send_matrix_host_to_device(self):
        m = self.csr_matrix_
        self._make_plan()
        self.results_buf = capi.DeviceBuffer(4 * m.num_rows)
        capi.fill_f32(self.results_buf, 0.0, m.num_rows)  # spmv_module.h:368-369

    @staticmethod
    def _upload_dense(host, n):
        """Fresh n-float buffer holding `host`, zero-padded when it is shorter (device blocks are recycled:
        nothing may be assumed about what a new buffer holds)."""
        host = np.ascontiguousarray(host, dtype=np.float32)[:n]
        buf = capi.DeviceBuffer(4 * n)
        buf.write(host)
        if host.shape[0] < n:
            tail = capi.DeviceBuffer(4 * (n - host.shape[0]), ptr=buf.ptr + 4 * host.shape[0], keepalive=buf)
            capi.fill_f32(tail, 0.0, n - host.shape[0])
        return buf

    def send_vector_host_to_device(self, vector):
        self.vector_buf = self._upload_dense(vector, self.get_num_cols())

    def send_mask_host_to_device(self, mask):
        self.mask_buf = self._upload_dense(mask, self.get_num_rows())

    def bind_mask_buf(self, src_buf):
        self.mask_buf = src_buf

    def bind_vector_buf(self, src_buf):
        self.vector_buf = src_buf

    def bind_results_buf(self, src_buf):
        self.results_buf = src_buf

    def run(self):
        mask = self.mask_buf if self.mask_type_ != kNoMask else None
        if not self._plan_serves(self.semiring_.op):   # semiring switched after upload: re-format
            capi.sync()
            self._make_plan()
        self.plan_.run(self.vector_buf, mask, self.results_buf, self.semiring_.op, self.semiring_.zero,
                       self.mask_type_)
        self._finish(self.results_buf)

    def chain(self, on):
        """Extension (gl_spmv_plan_chain): the caller feeds every result straight back as the next vector and touches neither in
        between (PageRank.pull, SSSP.pull) -- the run's epilogue then prepares the next run's packed x.  -> active?"""
        if self.plan_ is None or not hasattr(self.plan_, "chain"):
            return False
        if on and os.environ.get("GRAPHLILY_SPMV_CHAIN", "1") == "0":      # (A/B and tests: every run launches its helper)
            return False
        return self.plan_.chain(on)

    # extensions for row-sharded (||,&&) runs: x as a bit vector (gl_spmv_plan_bits_words / gl_spmv_run_bits)
    def bits_words(self):
        if self.plan_ is None or (self.plan_.flags & capi.GL_PLAN_REFERENCE_ORDER):
            return 0
        return self.plan_.bits_words() if self._plan_serves(self.semiring_.op) else 0

    def bfs_pull_step(self, bits_in, bits_out, distance_buf, level):
        """Extension (gl_bfs_pull_step): this module's masked (||,&&) run + eWiseAdd(+0) + dense assign(level) of a
        BFS pull iteration in one launch, frontier in and out as bit vectors.  Raises GraphLilyError
        (GL_ERR_UNSUPPORTED) for split plans."""
        self.plan_.bfs_pull_step(bits_in, bits_out, distance_buf, level)
        self._finish(bits_out, distance_buf)

    def fused_bfs_ok(self):
        if self.plan_ is None or not self._plan_serves(self.semiring_.op) or self.semiring_.zero != 0.0:
            return False
        info = self.plan_.info()
        return info["layout"] == "boolean" and info["segments"] == 1 and self.row_begin_ % 64 == 0

    def run_bits(self, bits_buf):
        mask = self.mask_buf if self.mask_type_ != kNoMask else None
        self.plan_.run_bits(bits_buf, mask, self.results_buf, self.semiring_.zero, self.mask_type_)
        self._finish(self.results_buf)

    def send_vector_device_to_host(self):
        return self.vector_buf.read(np.float32, self.get_num_cols())

    def send_mask_device_to_host(self):
        return self.mask_buf.read(np.float32, self.get_num_rows())

    def send_results_device_to_host(self):
        return self.results_buf.read(np.float32, self.get_num_rows())


class SpMSpVModule(BaseModule):
    def __init__(self, out_buf_len=0):
        super().__init__()
        self.out_buf_len_ = out_buf_len
        self.mask_type_ = kNoMask
        self.semiring_ = ArithmeticSemiring
        self.csc_matrix_ = None
        self.plan_ = None
        self.row_begin_, self.row_end_ = 0, None
        self.vector_buf = self.mask_buf = self.results_buf = None

    def set_semiring(self, semiring):
        self.semiring_ = semiring

    def set_mask_type(self, mask_type):
        self.mask_type_ = mask_type

    def set_row_shard(self, row_begin, row_end):
        self.row_begin_, self.row_end_ = int(row_begin), int(row_end)

    def get_num_rows(self):
        return self.csc_matrix_.num_rows

    def get_num_cols(self):
        return self.csc_matrix_.num_cols

    def get_nnz(self):
        return int(self.csc_matrix_.adj_indptr[self.csc_matrix_.num_cols])

    def load_and_format_matrix(self, csc_matrix_float):
        self.csc_matrix_ = csc_matrix_float

    def send_matrix_host_to_device(self):
        m = self.csc_matrix_
        re = m.num_rows if self.row_end_ is None else self.row_end_
        self.plan_ = capi.SpMSpVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data,
                                     self.row_begin_, re)
        # results: num_rows + 1 sparse elements, zero-initialised (spmspv_module.h:283-285)
        self.results_buf = capi.DeviceBuffer(8 * (m.num_rows + 1))
        self.results_buf.write(np.zeros(m.num_rows + 1, dtype=IDX_VAL))

        self._attach(getattr(self, "pull_module_", None))

    def attach_pull(self, spmv_module):
        """Extension (no counterpart in the reference): let heavy (||,&&) frontiers run row-wise on the boolean
        plan of an SpMVModule that holds the same matrix and row shard (gl_spmspv_plan_attach_pull)."""
        self.pull_module_ = spmv_module
        if not hasattr(spmv_module, "pull_clients_"):
            spmv_module.pull_clients_ = []
        if self not in spmv_module.pull_clients_:
            spmv_module.pull_clients_.append(self)
        self._attach(spmv_module)

    def hint_vector_nnz(self, nnz):
        """Extension: the caller knows the next run's vector holds at most `nnz` entries (gl_spmspv_plan_hint)."""
        if self.plan_ is not None:
            self.plan_.hint(nnz)

    def _attach(self, spmv_module):
        if self.plan_ is None:
            return
        plan = getattr(spmv_module, "plan_", None) if spmv_module is not None else None
        self.plan_.attach_pull(None)
        if plan is None:
            return
        mine = self.plan_
        if (plan.num_rows, plan.num_cols, plan.row_begin, plan.row_end) != (mine.num_rows, mine.num_cols, mine.row_begin, mine.row_end):
            return      # the SpMV module already holds the next matrix; this module attaches when its own is sent
        # boolean layout -> serves (||,&&); general / pattern layout -> (min,+) and, unless GL_PLAN_NO_MULADD, (+,x)
        self.plan_.attach_pull(plan)

    def enable_own_pull(self):
        """Extension for stand-alone use (no SpMVModule around, e.g. the bench_spmspv sweep): build a row-wise plan of
        this module's own matrix for its current semiring and attach it.  Costs a second formatted copy of the matrix."""
        from . import io as _io
        c = self.csc_matrix_
        # the CSC arrays of A are the CSR arrays of A^T; transposing that once more gives A by rows
        at = _io.CSRMatrix(c.num_cols, c.num_rows, c.adj_data, c.adj_indices, c.adj_indptr)
        a = _io.csr2csc(at)
        csr = _io.CSRMatrix(c.num_rows, c.num_cols, a.adj_data, a.adj_indices, a.adj_indptr)
        own = SpMVModule(num_hbm_channels, 0, 0)
        own.set_semiring(self.semiring_)
        own.set_mask_type(kNoMask)
        if self.row_end_ is not None:
            own.set_row_shard(self.row_begin_, self.row_end_)
        own.load_and_format_matrix(csr, True)
        own.send_matrix_host_to_device()
        self.own_pull_ = own
        self.attach_pull(own)

    def send_vector_host_to_device(self, vector):
        """The vector may be shorter than num_cols + 1; the device copy is always that long
        (spmspv_module.h:280, :379)."""
        vector = np.ascontiguousarray(vector, dtype=IDX_VAL)
        cnt = int(vector["index"][0]) if vector.shape[0] else 0
        self.hint_vector_nnz(cnt)
        self.vector_buf = capi.DeviceBuffer(8 * (self.get_num_cols() + 1))
        up = vector[:self.get_num_cols() + 1]
        if up.shape[0] and cnt + 1 > up.shape[0]:
            # the head claims more entries than the vector holds (the reference would read its zero-initialised mirror):
            # device blocks are recycled memory, so the head is clamped to what is uploaded
            up = up.copy()
            up["index"][0] = cnt = up.shape[0] - 1
        self.vector_buf.write(up)
        # the host holds the CSC: remember the vector's size, the non-zeros of its columns and the longest of them, see run()
        # (gl_spmspv_plan_hint_work: a tiny vector runs as one launch, and one below the direction switch's threshold without
        # the decision kernels)
        self.tiny_ = None
        BaseModule.device_writes_ += 1
        if 0 < cnt < vector.shape[0] and self.csc_matrix_ is not None:   # (a row shard holds no more than the whole matrix)
            cols = vector["index"][1:1 + cnt].astype(np.int64)
            cols = cols[cols < self.get_num_cols()]
            ip = self.csc_matrix_.adj_indptr
            lens = ip[cols + 1].astype(np.int64) - ip[cols].astype(np.int64)
            self.tiny_ = (cnt, int(lens.sum()), BaseModule.device_writes_, int(lens.max()) if lens.shape[0] else 0)

    def send_mask_host_to_device(self, mask):
        mask = np.ascontiguousarray(mask, dtype=np.float32)
        self.mask_buf = capi.DeviceBuffer(4 * mask.shape[0])
        self.mask_buf.write(mask)

    def bind_mask_buf(self, src_buf):
        self.mask_buf = src_buf

    def bind_vector_buf(self, src_buf):
        self.vector_buf = src_buf
        self.tiny_ = None

    def _hint_tiny(self):
        """Before a run: is the vector still the tiny one this module uploaded (no module call has written since)?"""
        t = getattr(self, "tiny_", None)
        if t is not None and t[2] == BaseModule.device_writes_:
            self.plan_.hint_work(t[0], t[1], t[3])
            return True
        self.tiny_ = None
        return False

    def run(self):
        mask = self.mask_buf if self.mask_type_ != kNoMask else None
        tiny = self._hint_tiny()
        self.plan_.run(self.vector_buf, mask, self.results_buf, self.semiring_.op, self.semiring_.zero,
                       self.mask_type_)
        self._finish_run(self.results_buf)
        if tiny:      # this run wrote the results and the accumulator, not the vector
            self.tiny_ = (self.tiny_[0], self.tiny_[1], BaseModule.device_writes_, self.tiny_[3])

    def run_assign(self, inout_buf, val):
        """Extension (gl_spmspv_run_assign): run() followed by AssignVectorSparseModule.run(val) with the results as
        its mask and `inout_buf` as its inout -- BFS's push iteration (app/bfs.h:146-148) -- in one call."""
        mask = self.mask_buf if self.mask_type_ != kNoMask else None
        self.plan_.run_assign(self.vector_buf, mask, self.results_buf, self.semiring_.op, self.semiring_.zero,
                              self.mask_type_, inout_buf, val)
        self._finish_run(self.results_buf, inout_buf)

    def _finish_run(self, *written):
        """A blocking run waits for the operator's own completion record (gl_spmspv_wait: the fold's last workgroup stores the
        result count to page-locked memory) instead of the whole stream, and remembers the count for get_results_nnz."""
        BaseModule._wrote(*written)
        self.nnz_known_ = None
        self.run_stamp_ = BaseModule.device_writes_
        if self.blocking:
            n = self.plan_.wait()
            if n is not None:
                self.nnz_known_ = (n, self.results_buf, BaseModule.device_writes_)

    def get_results_nnz(self):
        k = getattr(self, "nnz_known_", None)
        if k is not None and k[1] is self.results_buf and k[2] == BaseModule.device_writes_:
            return k[0]       # (no module call has written device memory since the run reported it)
        # the run's own completion record, if it kept one (no device -> host copy) -- trusted only while that run is the last
        # module call that wrote results_buf (SSSP's relax step, enqueued behind the run, only reads it: app/sssp.h:218-221)
        if not self.blocking and getattr(self.plan_, "wait", None) is not None and \
                getattr(self, "run_stamp_", None) is not None and getattr(self.results_buf, "last_write_", None) == self.run_stamp_:
            n = self.plan_.wait()
            if n is not None:
                return n
        return capi.sparse_nnz(self.results_buf)

    def send_vector_device_to_host(self):
        return self.vector_buf.read(IDX_VAL, self.get_num_cols() + 1)

    def send_mask_device_to_host(self):
        return self.mask_buf.read(np.float32)

    def send_results_device_to_host(self):
        return self.results_buf.read(IDX_VAL, self.get_num_rows() + 1)


class eWiseAddModule(BaseModule):
    def __init__(self):
        super().__init__()
        self.in_buf = self.out_buf = None

    def send_in_host_to_device(self, in_vec):
        self.in_buf = capi.DeviceBuffer.from_host(np.ascontiguousarray(in_vec, dtype=np.float32))

    def allocate_out_buf(self, length):
        self.out_buf = capi.DeviceBuffer(4 * length)

    def bind_in_buf(self, src_buf):
        self.in_buf = src_buf

    def bind_out_buf(self, src_buf):
        self.out_buf = src_buf

    def run(self, length, val):
        capi.ewise_add(self.in_buf, self.out_buf, length, val)
        self._finish(self.out_buf)

    def send_out_device_to_host(self):
        return self.out_buf.read(np.float32)


class AssignVectorDenseModule(BaseModule):
    def __init__(self):
        super().__init__()
        self.mask_type_ = None
        self.mask_buf = self.inout_buf = None

    def set_mask_type(self, mask_type):
        if mask_type == kNoMask:
            _fatal("Please set the mask type")  # assign_vector_dense_module.h:88-95
        self.mask_type_ = mask_type

    def send_mask_host_to_device(self, mask):
        self.mask_buf = capi.DeviceBuffer.from_host(np.ascontiguousarray(mask, dtype=np.float32))

    def send_inout_host_to_device(self, inout):
        self.inout_buf = capi.DeviceBuffer.from_host(np.ascontiguousarray(inout, dtype=np.float32))

    def bind_mask_buf(self, src_buf):
        self.mask_buf = src_buf

    def bind_inout_buf(self, src_buf):
        self.inout_buf = src_buf

    def run(self, length, val):
        if self.mask_type_ not in (kMaskWriteToZero, kMaskWriteToOne):
            _fatal("Invalid mask type")  # assign_vector_dense_module.h:242-245
        capi.assign_dense(self.mask_buf, self.inout_buf, length, val, self.mask_type_)
        self._finish(self.inout_buf)

    def send_mask_device_to_host(self):
        return self.mask_buf.read(np.float32)

    def send_inout_device_to_host(self):
        return self.inout_buf.read(np.float32)


class AssignVectorSparseModule(BaseModule):
    def __init__(self, generate_new_frontier):
        super().__init__()
        self.generate_new_frontier_ = bool(generate_new_frontier)
        self.mask_buf = self.inout_buf = self.new_frontier_buf = None

    def send_mask_host_to_device(self, mask):
        mask = np.ascontiguousarray(mask, dtype=IDX_VAL)
        self.mask_buf = capi.DeviceBuffer.from_host(mask)
        if self.generate_new_frontier_:
            # the new frontier can never be longer than the mask (assign_vector_sparse_module.h:238-253)
            self.new_frontier_buf = capi.DeviceBuffer(mask.nbytes)
            self.new_frontier_buf.write(np.zeros(mask.shape[0], dtype=IDX_VAL))

    def send_inout_host_to_device(self, inout):
        self.inout_buf = capi.DeviceBuffer.from_host(np.ascontiguousarray(inout, dtype=np.float32))

    def bind_mask_buf(self, src_buf):
        self.mask_buf = src_buf

    def bind_inout_buf(self, src_buf):
        self.inout_buf = src_buf

    def bind_new_frontier_buf(self, src_buf):
        if not self.generate_new_frontier_:
            _fatal("[ERROR]: this->generate_new_frontier_ should be true")
        self.new_frontier_buf = src_buf

    def _max_entries(self):
        cap = self.mask_buf.nbytes // 8 - 1
        if self.generate_new_frontier_:
            cap = min(cap, self.new_frontier_buf.nbytes // 8 - 1)
        return max(cap, 0)

    def run(self, val=None):
        if val is None:
            # SSSP mode (assign_vector_sparse_module.h:295-303)
            if not self.generate_new_frontier_:
                _fatal("[ERROR]: this->generate_new_frontier_ should be true")
            capi.assign_sparse_new_frontier(self.mask_buf, self.inout_buf, self.new_frontier_buf,
                                            self._max_entries())
        else:
            # BFS mode (assign_vector_sparse_module.h:278-292)
            if self.generate_new_frontier_:
                _fatal("[ERROR]: this->generate_new_frontier_ should be false")
            capi.assign_sparse(self.mask_buf, self.inout_buf, val, self._max_entries())
        self._finish(self.inout_buf, self.new_frontier_buf)

    def send_mask_device_to_host(self):
        return self.mask_buf.read(IDX_VAL)

    def send_inout_device_to_host(self):
        return self.inout_buf.read(np.float32)

    def send_new_frontier_device_to_host(self):
        if not self.generate_new_frontier_:
            _fatal("[ERROR]: this->generate_new_frontier_ should be true")
        return self.new_frontier_buf.read(IDX_VAL)
