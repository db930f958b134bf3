"""ctypes binding of libgraphlily_hip.so (include/graphlily_hip.h).

This is the same C ABI the C++ module headers under include/graphlily/ call; the
Python side exists so pytest and bench.py can drive the HIP path.  There is no
CPU fallback here: if the shared library is missing, or no HIP device is
present, the calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRAPHLILY_HIP_LIB: another build of the same library (same-box A/B runs of kernel changes, scripts/ab_variants.sh)
LIB_PATH = os.environ.get("GRAPHLILY_HIP_LIB") or os.path.join(_HERE, "lib", "libgraphlily_hip.so")

# graphlily/global.h:83-87, :103-107
GL_OP_MULADD, GL_OP_ANDOR, GL_OP_ADDMIN = 0, 1, 2
GL_NOMASK, GL_MASK_WRITETOZERO, GL_MASK_WRITETOONE = 0, 1, 2

GL_OK = 0
GL_ERR_INVALID_ARG = -1
GL_ERR_HIP = -2
GL_ERR_NOT_INITIALIZED = -3
GL_ERR_IO = -4

# graphlily/global.h:69 idx_val_t with val_t = float
IDX_VAL = np.dtype([("index", np.uint32), ("val", np.float32)])

# every symbol include/graphlily_hip.h declares (tests check the .so exports them all)
EXPORTS = [
    "gl_init", "gl_host_bind_near_device", "gl_device_count", "gl_set_stream", "gl_reset_stream", "gl_sync", "gl_last_error", "gl_version",
    "gl_graph_begin_capture", "gl_graph_end_capture", "gl_graph_launch", "gl_graph_destroy",
    "gl_bfs_bits_begin", "gl_bfs_bits_push_step", "gl_bfs_bits_pull_step",     "gl_bfs_bits_shard_step", "gl_bfs_bits_shard_finish", "gl_dist_all_gather_bits_tally",
    "gl_buf_d2h_async", "gl_levels_pack", "gl_host_levels_unpack", "gl_host_unpack_threads", "gl_buf_d2h_levels", "gl_sync_levels_unpack",
    "gl_levels_stream_bytes", "gl_levels_stream_arm", "gl_levels_pack_stream", "gl_sync_levels_unpack_stream",
            "gl_dist_unique_id", "gl_dist_init", "gl_dist_destroy", "gl_dist_rank", "gl_dist_all_gather_f32",     "gl_dist_all_gather_sparse", "gl_dist_slice_plan",
    "gl_spmv_run_typed", "gl_spmspv_run_typed", "gl_ewise_add_typed", "gl_assign_dense_typed", "gl_assign_sparse_typed",
    "gl_assign_sparse_new_frontier_typed", "gl_sparse_to_dense_typed",
    "gl_buf_alloc", "gl_buf_free", "gl_buf_h2d", "gl_buf_d2h", "gl_buf_d2d", "gl_buf_fill_f32", "gl_buf_fill_u32",     "gl_host_alloc", "gl_host_free", "gl_host_pool_alloc", "gl_host_pool_free", "gl_pool_trim", "gl_pool_stats", "gl_host_pool_reserve", "gl_host_fill_u32", "gl_host_sparse_to_dense",
    "gl_spmv_plan_create", "gl_spmv_plan_create_ex", "gl_spmv_plan_destroy", "gl_spmv_plan_describe", "gl_spmv_plan_export", "gl_spmv_run", "gl_spmv_plan_chain",
    "gl_spmv_plan_bits_words", "gl_pack_bits", "gl_unpack_bits", "gl_bfs_bits_begin_from", "gl_spmv_run_bits", "gl_bfs_pull_step",
    "gl_prof_begin", "gl_prof_end", "gl_span_begin", "gl_span_end",
    "gl_spmspv_plan_create", "gl_spmspv_plan_destroy", "gl_spmspv_plan_info", "gl_spmspv_run", "gl_spmspv_run_assign",
    "gl_spmspv_plan_attach_pull", "gl_spmspv_plan_hint", "gl_spmspv_plan_hint_work", "gl_spmspv_last_direction", "gl_spmspv_wait", "gl_spmspv_failed_runs",
    "gl_sparse_nnz", "gl_ewise_add", "gl_assign_dense", "gl_assign_sparse",
    "gl_assign_sparse_new_frontier", "gl_sparse_to_dense",
    "gl_csr2csc", "gl_csr_normalize_by_outdegree", "gl_npz_csr_open", "gl_npz_csr_read", "gl_npz_csr_close",
]


class GraphLilyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("graphlily_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libgraphlily_hip.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GraphLilyError(GL_ERR_NOT_INITIALIZED,
                             "%s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7; if torch is going to be used
    # at all (bench.py, multi-GPU) it must be the first to load it, or its device discovery fails.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, u64, f32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float, ctypes.c_int
    P = ctypes.POINTER
    sigs = {
        "gl_graph_begin_capture": [], "gl_graph_end_capture": [P(vp)], "gl_graph_launch": [vp], "gl_graph_destroy": [vp],
        "gl_bfs_bits_begin": [vp, u32, vp, u32, vp, u32, u32, u32],
        "gl_buf_d2h_async": [vp, vp, ctypes.c_size_t],
        "gl_levels_pack": [vp, u32, i32, vp, u32, vp], "gl_host_levels_unpack": [vp, vp, ctypes.c_size_t, i32], "gl_host_unpack_threads": [], "gl_buf_d2h_levels": [vp, vp, ctypes.c_size_t, f32, P(i32)], "gl_sync_levels_unpack": [vp, vp, ctypes.c_size_t, i32],
        "gl_levels_stream_bytes": [u32, i32, u32, P(ctypes.c_size_t)], "gl_levels_stream_arm": [vp, u32, i32, u32],
        "gl_levels_pack_stream": [vp, u32, i32, vp, u32, vp], "gl_sync_levels_unpack_stream": [vp, vp, ctypes.c_size_t, i32, vp, u32],
        "gl_bfs_bits_push_step": [vp, vp, vp, vp, vp, u32, vp, f32, vp, u32, f32, i32],
        "gl_bfs_bits_pull_step": [vp, vp, vp, vp, vp, f32, vp, u32, f32, i32, f32],
        "gl_bfs_bits_shard_step": [vp, vp, vp, vp, u32, vp, f32, vp, vp, vp, u32, i32, i32, vp, u64, f32, i32, f32],
        "gl_bfs_bits_shard_finish": [vp, vp, vp, vp, vp, u32, i32, i32, u64, f32, i32, f32],
        "gl_dist_all_gather_bits_tally": [vp, vp, vp, vp, u32],
        "gl_dist_unique_id": [vp], "gl_dist_init": [P(vp), i32, i32, vp], "gl_dist_destroy": [vp], "gl_dist_rank": [vp, P(i32), P(i32)],
        "gl_dist_all_gather_f32": [vp, vp, vp],
        "gl_dist_all_gather_sparse": [vp, vp, vp, u32, f32, P(u32)], "gl_dist_slice_plan": [i32, i32, vp, vp, vp],
        "gl_spmv_run_typed": [vp, vp, vp, vp, i32, u32, i32, i32], "gl_spmspv_run_typed": [vp, vp, vp, vp, i32, u32, i32, i32],
        "gl_ewise_add_typed": [vp, vp, u32, u32, i32], "gl_assign_dense_typed": [vp, vp, u32, u32, i32, i32],
        "gl_assign_sparse_typed": [vp, vp, u32, u32], "gl_assign_sparse_new_frontier_typed": [vp, vp, vp, u32, i32],
        "gl_sparse_to_dense_typed": [vp, vp, u32, u32, u32],
        "gl_init": [i32], "gl_host_bind_near_device": [P(i32), P(i32)], "gl_device_count": [P(i32)], "gl_set_stream": [vp], "gl_reset_stream": [], "gl_sync": [],
        "gl_buf_alloc": [P(vp), ctypes.c_size_t], "gl_buf_free": [vp],
        "gl_host_alloc": [P(vp), ctypes.c_size_t], "gl_host_free": [vp],
        "gl_host_pool_alloc": [P(vp), ctypes.c_size_t], "gl_host_pool_free": [vp], "gl_pool_trim": [], "gl_pool_stats": [P(u64), P(u64), P(u32)],
        "gl_host_pool_reserve": [ctypes.c_size_t, u32], "gl_host_fill_u32": [vp, u32, ctypes.c_size_t], "gl_host_sparse_to_dense": [vp, u32, u32, vp],
        "gl_buf_h2d": [vp, vp, ctypes.c_size_t], "gl_buf_d2h": [vp, vp, ctypes.c_size_t],
        "gl_buf_d2d": [vp, vp, ctypes.c_size_t], "gl_buf_fill_f32": [vp, f32, ctypes.c_size_t],
        "gl_buf_fill_u32": [vp, u32, ctypes.c_size_t],
        "gl_spmv_plan_create": [P(vp), u32, u32, vp, vp, vp, u32, u32],
        "gl_spmv_plan_create_ex": [P(vp), u32, u32, vp, vp, vp, u32, u32, u32],
        "gl_spmv_plan_destroy": [vp], "gl_spmv_plan_describe": [vp, vp],
        "gl_spmv_plan_export": [vp, i32, vp, ctypes.c_size_t, P(ctypes.c_size_t)], "gl_spmv_plan_chain": [vp, i32, P(i32)],
        "gl_spmv_plan_bits_words": [vp, P(u64)], "gl_pack_bits": [vp, u32, vp], "gl_unpack_bits": [vp, u32, vp], "gl_bfs_bits_begin_from": [vp, u32, vp, u32, vp, u32, vp, vp], "gl_spmv_run_bits": [vp, vp, vp, vp, f32, i32],
        "gl_bfs_pull_step": [vp, vp, vp, vp, f32],
        "gl_spmv_run": [vp, vp, vp, vp, i32, f32, i32],
        "gl_prof_begin": [u32, u32], "gl_prof_end": [P(ctypes.c_double), P(u32)], "gl_span_begin": [], "gl_span_end": [P(ctypes.c_double)],
        "gl_spmspv_plan_create": [P(vp), u32, u32, vp, vp, vp, u32, u32],
        "gl_spmspv_plan_destroy": [vp], "gl_spmspv_plan_info": [vp, P(u64), P(u64)],
        "gl_spmspv_run": [vp, vp, vp, vp, i32, f32, i32],
        "gl_spmspv_run_assign": [vp, vp, vp, vp, i32, f32, i32, vp, f32],
        "gl_spmspv_plan_attach_pull": [vp, vp], "gl_spmspv_plan_hint": [vp, u32], "gl_spmspv_plan_hint_work": [vp, u32, u64, u32], "gl_spmspv_last_direction": [vp, P(i32)], "gl_spmspv_wait": [vp, P(u32)], "gl_spmspv_failed_runs": [vp, P(u32)],
        "gl_sparse_nnz": [vp, P(u32)],
        "gl_ewise_add": [vp, vp, u32, f32], "gl_assign_dense": [vp, vp, u32, f32, i32],
        "gl_assign_sparse": [vp, vp, f32, u32], "gl_assign_sparse_new_frontier": [vp, vp, vp, u32],
        "gl_sparse_to_dense": [vp, vp, u32, f32, u32],
        "gl_csr2csc": [u32, u32, vp, vp, vp, vp, vp, vp],
        "gl_csr_normalize_by_outdegree": [u32, u32, vp, vp, vp],
        "gl_npz_csr_open": [ctypes.c_char_p, P(vp), P(u32), P(u32), P(u64)],
        "gl_npz_csr_read": [vp, vp, vp, vp], "gl_npz_csr_close": [vp],
    }
    for name, argtypes in sigs.items():
        if os.environ.get("GRAPHLILY_HIP_LIB") and not hasattr(L, name):
            continue   # an older build in an A/B run: entry points added since stay unbound
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = i32
    L.gl_last_error.restype = ctypes.c_char_p
    L.gl_last_error.argtypes = []
    L.gl_version.restype = ctypes.c_char_p
    L.gl_version.argtypes = []
    _lib = L
    return L


def check(rc):
    if rc != GL_OK:
        raise GraphLilyError(rc, lib().gl_last_error().decode("utf-8", "replace"))


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None and a.size else ctypes.c_void_p(0)


def init(device=0):
    check(lib().gl_init(int(device)))


def host_bind_near_device():
    """gl_host_bind_near_device: the calling thread onto the CPUs of the device's NUMA node -> (node, cpus), (-1, 0) if nothing applied.
    gl_init does this itself unless GRAPHLILY_BIND_NUMA=0."""
    node, cpus = ctypes.c_int(-1), ctypes.c_int(0)
    check(lib().gl_host_bind_near_device(ctypes.byref(node), ctypes.byref(cpus)))
    return node.value, cpus.value


def pool_stats():
    """-> (live device blocks, cached bytes, slabs) of the device block pool (gl_pool_stats)."""
    a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint32(0)
    check(lib().gl_pool_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def pool_trim():
    check(lib().gl_pool_trim())


def device_count():
    n = ctypes.c_int(0)
    rc = lib().gl_device_count(ctypes.byref(n))
    return n.value if rc == GL_OK else 0


def set_stream(hip_stream):
    check(lib().gl_set_stream(ctypes.c_void_p(hip_stream or 0)))


def reset_stream():
    check(lib().gl_reset_stream())


def sync():
    check(lib().gl_sync())


class _PinnedBlock:
    def __init__(self, nbytes):
        p = ctypes.c_void_p(0)
        check(lib().gl_host_alloc(ctypes.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        try:
            if self.ptr:
                lib().gl_host_free(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


_pinned_free = {}      # nbytes -> [_PinnedBlock]: blocks whose arrays have been dropped (pinned_recycled)


class _RecycledBlock:
    """Owner of a page-locked block handed out by pinned_recycled(): returns it to the free list when the array dies."""

    def __init__(self, blk):
        self.blk = blk
        self.ptr, self.nbytes = blk.ptr, blk.nbytes

    def __del__(self):
        try:
            lst = _pinned_free.setdefault(self.nbytes, [])
            if len(lst) < 4:
                lst.append(self.blk)
        except Exception:
            pass


def pinned_recycled(count, dtype):
    """pinned_empty() from a free list: page-locking 12 MB costs ~2 ms, so result arrays that a driver returns by value
    take their block from the arrays the caller has dropped (a caller that keeps every result just gets new blocks).
    A copy into page-locked memory can be enqueued behind the kernels that produce its source (DeviceBuffer.read_async)."""
    dtype = np.dtype(dtype)
    nbytes = max(count * dtype.itemsize, 4)
    lst = _pinned_free.get(nbytes)
    blk = _RecycledBlock(lst.pop() if lst else _PinnedBlock(nbytes))
    raw = (ctypes.c_char * blk.nbytes).from_address(blk.ptr)
    arr = np.frombuffer(raw, dtype=dtype, count=count).view(_OwnedArray)
    arr._owner = (raw, blk)
    return arr


def pinned_empty(count, dtype):
    """numpy array in page-locked host memory (freed with the array): read-backs into it run at the
    full PCIe rate instead of the pageable-memory rate."""
    dtype = np.dtype(dtype)
    blk = _PinnedBlock(max(count * dtype.itemsize, 4))
    raw = (ctypes.c_char * blk.nbytes).from_address(blk.ptr)
    arr = np.frombuffer(raw, dtype=dtype, count=count)
    arr = arr.view(_OwnedArray)
    arr._owner = (raw, blk)
    return arr


class _OwnedArray(np.ndarray):
    """ndarray that keeps its pinned block alive (views inherit the reference through .base)."""
    _owner = None

    def __array_finalize__(self, obj):
        if obj is not None and getattr(obj, "_owner", None) is not None:
            self._owner = obj._owner



class DeviceBuffer:
    """A device allocation handle: the role cl::Buffer plays in the reference's module API
    (shareable between modules via bind_*_buf).  Either owns memory from gl_buf_alloc or
    wraps a foreign device pointer (e.g. a torch tensor kept alive by `keepalive`)."""

    def __init__(self, nbytes=0, ptr=None, keepalive=None):
        self.nbytes = int(nbytes)
        self._owned = ptr is None
        self.keepalive = keepalive
        if self._owned:
            p = ctypes.c_void_p(0)
            check(lib().gl_buf_alloc(ctypes.byref(p), self.nbytes))
            self.ptr = p.value or 0
        else:
            self.ptr = int(ptr)

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        b.write(arr)
        return b

    @classmethod
    def from_torch(cls, t):
        return cls(t.numel() * t.element_size(), ptr=t.data_ptr(), keepalive=t)

    def write(self, arr, offset_bytes=0):
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= self.nbytes, "host array larger than device buffer"
        check(lib().gl_buf_h2d(ctypes.c_void_p(self.ptr + offset_bytes), _np_ptr(arr), arr.nbytes))

    def read(self, dtype, count=None, offset_bytes=0, out=None):
        """Blocking device->host copy.  `out` may be a preallocated array (e.g. pinned_empty(), which
        makes the copy run at the full PCIe rate; pinning is slow, so callers allocate it once)."""
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset_bytes) // dtype.itemsize
        if out is None:
            out = np.empty(count, dtype=dtype)
        else:
            assert out.dtype == dtype and out.shape[0] >= count and out.flags["C_CONTIGUOUS"]
            out = out[:count]
        assert offset_bytes + out.nbytes <= self.nbytes
        check(lib().gl_buf_d2h(_np_ptr(out), ctypes.c_void_p(self.ptr + offset_bytes), out.nbytes))
        return out

    def read_async(self, out, offset_bytes=0):
        """Enqueue the device->host copy into `out` (page-locked: pinned_recycled / pinned_empty) behind the kernels already
        on the library stream, without waiting; capi.sync() before `out` is read."""
        assert out.flags["C_CONTIGUOUS"] and offset_bytes + out.nbytes <= self.nbytes
        check(lib().gl_buf_d2h_async(_np_ptr(out), ctypes.c_void_p(self.ptr + offset_bytes), out.nbytes))
        return out

    def free(self):
        if self._owned and self.ptr:
            lib().gl_buf_free(ctypes.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def copy_d2d(dst, src, nbytes):
    assert nbytes <= dst.nbytes and nbytes <= src.nbytes
    check(lib().gl_buf_d2d(ctypes.c_void_p(dst.ptr), ctypes.c_void_p(src.ptr), int(nbytes)))


def _p(buf):
    return ctypes.c_void_p(buf.ptr if buf is not None else 0)


GL_PLAN_NO_MULADD = 1
GL_PLAN_BOOLEAN = 2
GL_PLAN_KEEP_VALUES = 4
GL_VAL_FLOAT, GL_VAL_UNSIGNED, GL_VAL_UFIXED_32_8 = 0, 1, 2
GL_PLAN_HOST_FORMAT = 8
GL_PLAN_DEVICE_FORMAT = 16
GL_PLAN_REFERENCE_ORDER = 32
PLAN_ARRAYS = {"entries": 0, "bases": 1, "units": 2, "hub_rows": 3, "spans": 4, "hot": 5, "hot_hdr": 6, "present": 7}
GL_ERR_UNSUPPORTED = -5


class _PlanDesc(ctypes.Structure):
    """gl_spmv_plan_desc (include/graphlily_hip.h)"""
    _fields_ = [("nnz", ctypes.c_uint64), ("device_bytes", ctypes.c_uint64), ("groups", ctypes.c_uint64), ("hot_nnz", ctypes.c_uint64),
                ("num_units", ctypes.c_uint32), ("blocks", ctypes.c_uint32), ("segments", ctypes.c_uint32), ("max_block_rows", ctypes.c_uint32),
                ("hot_columns", ctypes.c_uint32), ("packed_columns", ctypes.c_uint32),
                ("layout", ctypes.c_int), ("mix", ctypes.c_int), ("helper", ctypes.c_int)]


class SpMVPlan:
    def __init__(self, num_rows, num_cols, indptr, indices, data, row_begin=0, row_end=None, flags=0):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        row_end = num_rows if row_end is None else row_end
        h = ctypes.c_void_p(0)
        check(lib().gl_spmv_plan_create_ex(ctypes.byref(h), num_rows, num_cols, _np_ptr(self.indptr),
                                           _np_ptr(self.indices), _np_ptr(self.data), row_begin, row_end,
                                           int(flags)))
        self.flags = int(flags)
        self.handle = h.value
        self.num_rows, self.num_cols, self.row_begin, self.row_end = num_rows, num_cols, row_begin, row_end
        # host CSR copies are only needed during creation
        self.indptr = self.indices = self.data = None

    def info(self):
        cached = getattr(self, "_info", None)      # (a plan never changes: the drivers ask on every call)
        if cached is not None:
            return dict(cached)
        d = _PlanDesc()
        check(lib().gl_spmv_plan_describe(ctypes.c_void_p(self.handle), ctypes.byref(d)))
        self._info = {"helper": ("gather", "spread", "self-hot", "none")[d.helper], "packed_columns": d.packed_columns,
                      "nnz": d.nnz, "device_bytes": d.device_bytes, "num_units": d.num_units, "blocks": d.blocks,
                      "segments": d.segments, "max_block_rows": d.max_block_rows, "groups": d.groups,
                      "hot_columns": d.hot_columns, "hot_nnz": d.hot_nnz, "mix": d.mix,
                      "layout": ("general", "pattern", "boolean", "reference-order")[d.layout]}
        return dict(self._info)

    def run(self, x, mask, y, op, zero, mask_type):
        check(lib().gl_spmv_run(ctypes.c_void_p(self.handle), _p(x), _p(mask), _p(y), int(op), float(zero),
                                int(mask_type)))

    def run_typed(self, x, mask, y, op, zero_bits, mask_type, val_type):
        """gl_spmv_run_typed: the buffers hold 32-bit value words of `val_type` (GL_VAL_*)."""
        check(lib().gl_spmv_run_typed(ctypes.c_void_p(self.handle), _p(x), _p(mask), _p(y), int(op), int(zero_bits), int(mask_type),
                                      int(val_type)))

    def chain(self, on):
        """gl_spmv_plan_chain: while on, a run's epilogue also leaves y in the next run's packed form and a run on the previous
        run's y skips the helper launch.  -> whether this plan can chain at all."""
        active = ctypes.c_int(0)
        check(lib().gl_spmv_plan_chain(ctypes.c_void_p(self.handle), 1 if on else 0, ctypes.byref(active)))
        return bool(active.value)

    def export(self, name):
        """One of the plan's formatted device arrays (PLAN_ARRAYS) as uint32 words (gl_spmv_plan_export)."""
        n = ctypes.c_size_t(0)
        check(lib().gl_spmv_plan_export(ctypes.c_void_p(self.handle), PLAN_ARRAYS[name], None, 0, ctypes.byref(n)))
        out = np.empty(n.value // 4, dtype=np.uint32)
        check(lib().gl_spmv_plan_export(ctypes.c_void_p(self.handle), PLAN_ARRAYS[name], _np_ptr(out), out.nbytes, ctypes.byref(n)))
        return out

    def bits_words(self):
        v = ctypes.c_uint64(0)
        check(lib().gl_spmv_plan_bits_words(ctypes.c_void_p(self.handle), ctypes.byref(v)))
        return v.value

    def bfs_pull_step(self, bits_in, bits_out, distance, level):
        check(lib().gl_bfs_pull_step(ctypes.c_void_p(self.handle), _p(bits_in), _p(bits_out), _p(distance), float(level)))

    def run_bits(self, bits, mask, y, zero, mask_type):
        check(lib().gl_spmv_run_bits(ctypes.c_void_p(self.handle), _p(bits), _p(mask), _p(y), float(zero), int(mask_type)))

    def destroy(self):
        if getattr(self, "handle", None):
            lib().gl_spmv_plan_destroy(ctypes.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class SpMSpVPlan:
    def __init__(self, num_rows, num_cols, indptr, indices, data, row_begin=0, row_end=None):
        indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        data = np.ascontiguousarray(data, dtype=np.float32)
        row_end = num_rows if row_end is None else row_end
        h = ctypes.c_void_p(0)
        check(lib().gl_spmspv_plan_create(ctypes.byref(h), num_rows, num_cols, _np_ptr(indptr), _np_ptr(indices),
                                          _np_ptr(data), row_begin, row_end))
        self.handle = h.value
        self.num_rows, self.num_cols, self.row_begin, self.row_end = num_rows, num_cols, row_begin, row_end

    def info(self):
        nnz, nbytes = ctypes.c_uint64(0), ctypes.c_uint64(0)
        check(lib().gl_spmspv_plan_info(ctypes.c_void_p(self.handle), ctypes.byref(nnz), ctypes.byref(nbytes)))
        return {"nnz": nnz.value, "device_bytes": nbytes.value}

    def attach_pull(self, spmv_plan):
        """Let heavy (||,&&) frontiers run row-wise on a GL_PLAN_BOOLEAN SpMV plan of the same matrix."""
        check(lib().gl_spmspv_plan_attach_pull(ctypes.c_void_p(self.handle),
                                               ctypes.c_void_p(spmv_plan.handle) if spmv_plan is not None else None))
        self._pull_keepalive = spmv_plan

    def hint(self, vector_nnz_upper_bound):
        check(lib().gl_spmspv_plan_hint(ctypes.c_void_p(self.handle), int(vector_nnz_upper_bound)))

    def hint_work(self, vector_nnz, work, longest_column):
        """gl_spmspv_plan_hint_work: the next run's vector holds `vector_nnz` entries whose columns hold `work` non-zeros, the
        longest `longest_column` of them."""
        check(lib().gl_spmspv_plan_hint_work(ctypes.c_void_p(self.handle), int(vector_nnz), int(work), int(longest_column)))

    def last_direction(self):
        v = ctypes.c_int(0)
        check(lib().gl_spmspv_last_direction(ctypes.c_void_p(self.handle), ctypes.byref(v)))
        return "row-wise" if v.value else "scatter"

    def run(self, vector, mask, result, op, zero, mask_type):
        check(lib().gl_spmspv_run(ctypes.c_void_p(self.handle), _p(vector), _p(mask), _p(result), int(op),
                                  float(zero), int(mask_type)))

    def wait(self):
        """gl_spmspv_wait: block until the last run on this plan has written its results; returns the result count (None
        when that run kept no completion record: a gated, captured or one-launch run -- then this is gl_sync)."""
        n = ctypes.c_uint32(0)
        check(lib().gl_spmspv_wait(ctypes.c_void_p(self.handle), ctypes.byref(n)))
        return None if n.value == 0xffffffff else n.value

    def failed_runs(self):
        """gl_spmspv_failed_runs: runs on this plan whose workgroup rendezvous timed out so far (sticky; what a caller of
        recorded graphs checks after a replay)."""
        n = ctypes.c_uint32(0)
        check(lib().gl_spmspv_failed_runs(ctypes.c_void_p(self.handle), ctypes.byref(n)))
        return n.value

    def run_typed(self, vector, mask, result, op, zero_bits, mask_type, val_type):
        check(lib().gl_spmspv_run_typed(ctypes.c_void_p(self.handle), _p(vector), _p(mask), _p(result), int(op), int(zero_bits),
                                        int(mask_type), int(val_type)))

    def run_assign(self, vector, mask, result, op, zero, mask_type, inout, val):
        """gl_spmspv_run_assign: run + gl_assign_sparse(result, inout, val) in the result-writing pass."""
        check(lib().gl_spmspv_run_assign(ctypes.c_void_p(self.handle), _p(vector), _p(mask), _p(result), int(op),
                                         float(zero), int(mask_type), _p(inout), float(val)))

    def destroy(self):
        if getattr(self, "handle", None):
            lib().gl_spmspv_plan_destroy(ctypes.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Graph:
    """A recorded launch sequence (gl_graph_*): `with Graph.capture() as g: ...enqueue...`, then g.launch()."""

    live = None      # weak set of the graphs alive (Dist.destroy ends them first: RCCL waits for graphs holding its operations)

    def __init__(self):
        import weakref
        self.handle = None
        if Graph.live is None:
            Graph.live = weakref.WeakSet()
        Graph.live.add(self)

    def destroy(self):
        h, self.handle = self.handle, None
        if h:
            lib().gl_graph_destroy(ctypes.c_void_p(h))

    @classmethod
    def capture(cls):
        # No garbage collection while the stream records: a collected plan or buffer of some earlier object is freed with
        # hipFree, which is not allowed on a capturing thread and invalidates the capture (seen as "operation failed due to a
        # previous error during capture" when a BFS schedule was recorded right after tests that had dropped many plans).
        import gc
        g = cls()
        g._gc_was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            check(lib().gl_graph_begin_capture())
        except Exception:
            if g._gc_was_enabled:
                gc.enable()
            raise
        return g

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        h = ctypes.c_void_p(0)
        rc = lib().gl_graph_end_capture(ctypes.byref(h))
        if getattr(self, "_gc_was_enabled", False):
            import gc
            gc.enable()
        if exc_type is None:
            check(rc)
            self.handle = h.value
        return False

    def launch(self):
        if not self.handle:
            raise GraphLilyError(GL_ERR_INVALID_ARG, "Graph.launch: the graph has been destroyed")
        check(lib().gl_graph_launch(ctypes.c_void_p(self.handle)))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def dist_slice_plan(kind, bounds_or_counts):
    """gl_dist_slice_plan -> (lo_bytes, hi_bytes) per rank; kind 0 f32 bounds, 1 bit-vector row bounds, 2 sparse counts."""
    a = np.ascontiguousarray(bounds_or_counts, dtype=np.uint32)
    world = a.shape[0] - (0 if kind == 2 else 1)
    lo, hi = np.zeros(world, np.uint64), np.zeros(world, np.uint64)
    check(lib().gl_dist_slice_plan(int(kind), int(world), _np_ptr(a), ctypes.c_void_p(lo.ctypes.data), ctypes.c_void_p(hi.ctypes.data)))
    return lo, hi


class Dist:
    """gl_dist_*: the RCCL exchange step of the row-sharded path through the C ABI (one communicator per process)."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().gl_dist_unique_id(buf))
        return buf.raw

    def __init__(self, rank, world_size, unique_id):
        h = ctypes.c_void_p(0)
        check(lib().gl_dist_init(ctypes.byref(h), int(rank), int(world_size), ctypes.c_char_p(unique_id)))
        self.handle, self.rank, self.world_size = h.value, int(rank), int(world_size)

    def all_gather_f32(self, full, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.uint32)
        check(lib().gl_dist_all_gather_f32(ctypes.c_void_p(self.handle), _p(full), _np_ptr(b)))

    def all_gather_bits_tally(self, bits, row_bounds, tally_slot):
        """the slot's bit vector and (tally_slot not None) every rank's tallies of the slot in one grouped operation"""
        b = np.ascontiguousarray(row_bounds, dtype=np.uint32)
        check(lib().gl_dist_all_gather_bits_tally(ctypes.c_void_p(self.handle), _p(bits), _np_ptr(b), _p(tally_slot),
                                                  4 * GL_BFS_TALLY_RANK_WORDS if tally_slot is not None else 0))

    def all_gather_sparse(self, local, full, capacity, head_val):
        n = ctypes.c_uint32(0)
        check(lib().gl_dist_all_gather_sparse(ctypes.c_void_p(self.handle), _p(local), _p(full), int(capacity), float(head_val),
                                              ctypes.byref(n)))
        return n.value

    def destroy(self):
        if getattr(self, "handle", None):
            # (graphs that recorded an exchange hold RCCL operations: the communicator's destroy would wait for them)
            for g in list(Graph.live or ()):
                g.destroy()
            check(lib().gl_dist_destroy(ctypes.c_void_p(self.handle)))
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def fill_u32(buf, value, count):
    check(lib().gl_buf_fill_u32(_p(buf), int(value), int(count)))


def bfs_bits_begin(ctl, ctl_words, distance, n, bits, bits_words, nvec, first_pull_slot=0xffffffff):
    check(lib().gl_bfs_bits_begin(_p(ctl), int(ctl_words), _p(distance), int(n), _p(bits), int(bits_words), int(nvec),
                                  int(first_pull_slot)))


def bfs_bits_push_step(csc_plan, rows_plan, bits_in, bits_out, bits_spare, bits_words, distance, level, ctl, slot, threshold, may_continue):
    check(lib().gl_bfs_bits_push_step(ctypes.c_void_p(csc_plan.handle), ctypes.c_void_p(rows_plan.handle if rows_plan is not None else 0),
                                      _p(bits_in), _p(bits_out), _p(bits_spare), int(bits_words),
                                      _p(distance), float(level), _p(ctl), int(slot), float(threshold), int(may_continue)))


def bfs_bits_pull_step(pull_plan, csc_plan, bits_in, bits_out, distance, level, ctl, slot, threshold, may_continue, back_threshold):
    check(lib().gl_bfs_bits_pull_step(ctypes.c_void_p(pull_plan.handle), ctypes.c_void_p(csc_plan.handle), _p(bits_in), _p(bits_out),
                                      _p(distance), float(level), _p(ctl), int(slot), float(threshold), int(may_continue),
                                      float(back_threshold)))




GL_BFS_TALLY_HEAD_WORDS = 64
GL_BFS_TALLY_RANK_WORDS = 64


def bfs_tally_words(slots, world):
    return GL_BFS_TALLY_HEAD_WORDS + int(slots) * int(world) * GL_BFS_TALLY_RANK_WORDS


def bfs_bits_shard_step(csc_plan, rows_plan, bits_in, bits_out, words, distance, level, ctl, tally, tally_in, slot, rank, world, col_len,
                        nnz_global, threshold, may_continue_prev, back_threshold):
    """gl_bfs_bits_shard_step: one slot of the row-sharded bit-frontier schedule in ONE launch."""
    check(lib().gl_bfs_bits_shard_step(ctypes.c_void_p(csc_plan.handle), ctypes.c_void_p(rows_plan.handle), _p(bits_in), _p(bits_out),
                                       int(words), _p(distance), float(level), _p(ctl), _p(tally), _p(tally_in), int(slot), int(rank),
                                       int(world), _p(col_len), int(nnz_global), float(threshold), int(may_continue_prev),
                                       float(back_threshold)))


def bfs_bits_shard_finish(csc_plan, rows_plan, ctl, tally, tally_in, last_slot, rank, world, nnz_global, threshold, may_continue_last,
                          back_threshold):
    check(lib().gl_bfs_bits_shard_finish(ctypes.c_void_p(csc_plan.handle), ctypes.c_void_p(rows_plan.handle), _p(ctl), _p(tally),
                                         _p(tally_in), int(last_slot), int(rank), int(world), int(nnz_global), float(threshold),
                                         int(may_continue_last), float(back_threshold)))


def levels_pack(levels, n, bits, tail, tail_words, out):
    """gl_levels_pack: n float levels (small integers) -> bytes (bits 8) or nibbles (bits 4) + `tail_words` raw words of `tail`."""
    check(lib().gl_levels_pack(_p(levels), int(n), int(bits), _p(tail), int(tail_words), _p(out)))


def levels_packed_words(n, bits):
    """32-bit words of n packed levels, rounded up to the 16-byte boundary the tail starts on"""
    return (n // (32 // bits) + 3) & ~3


def host_levels_unpack(dst, src, n, bits):
    """gl_host_levels_unpack: dst[:n] (float32 array) = the levels packed in src (uint8 array), on a few host threads."""
    assert dst.dtype == np.float32 and src.dtype == np.uint8 and dst.shape[0] >= n and src.shape[0] * (8 // bits) >= n
    check(lib().gl_host_levels_unpack(_np_ptr(dst), _np_ptr(src), int(n), int(bits)))


def sync_levels_unpack(dst, src, n, bits):
    """gl_sync_levels_unpack: wait for the library's stream, then host_levels_unpack -- the threads start before the wait."""
    assert dst.dtype == np.float32 and src.dtype == np.uint8 and dst.shape[0] >= n and src.shape[0] * (8 // bits) >= n
    check(lib().gl_sync_levels_unpack(_np_ptr(dst), _np_ptr(src), int(n), int(bits)))


def levels_stream_bytes(n, bits, tail_words):
    """gl_levels_stream_bytes: size of the page-locked block of a streamed read-back (packed levels, tail words, chunk flags)"""
    b = ctypes.c_size_t(0)
    check(lib().gl_levels_stream_bytes(int(n), int(bits), int(tail_words), ctypes.byref(b)))
    return int(b.value)


def levels_stream_arm(block, n, bits, tail_words):
    """gl_levels_stream_arm: clear the block's chunk flags -- on the host, before the pack (or the graph holding it) is launched"""
    check(lib().gl_levels_stream_arm(_np_ptr(block), int(n), int(bits), int(tail_words)))


def levels_pack_stream(levels, n, bits, tail, tail_words, block):
    """gl_levels_pack_stream: levels_pack whose kernel stores into the page-locked host `block` (pinned_empty) chunk by chunk"""
    check(lib().gl_levels_pack_stream(_p(levels), int(n), int(bits), _p(tail), int(tail_words), _np_ptr(block)))


def sync_levels_unpack_stream(dst, block, n, bits, tail_words):
    """gl_sync_levels_unpack_stream: the floats of a streamed read-back (chunks expanded as they land) -> the tail words"""
    assert dst.dtype == np.float32 and dst.shape[0] >= n and block.dtype == np.uint8
    tail = np.empty(tail_words, np.uint32)
    check(lib().gl_sync_levels_unpack_stream(_np_ptr(dst), _np_ptr(block), int(n), int(bits), _np_ptr(tail), int(tail_words)))
    return tail


def d2h_levels(dst, buf, n, max_level):
    """gl_buf_d2h_levels: dst[:n] = the n floats of `buf`, packed on the way if they are levels <= max_level.  -> packed?"""
    assert dst.dtype == np.float32 and dst.shape[0] >= n
    used = ctypes.c_int(0)
    check(lib().gl_buf_d2h_levels(_np_ptr(dst), _p(buf), int(n), float(max_level), ctypes.byref(used)))
    return bool(used.value)


def host_unpack_threads():
    return int(lib().gl_host_unpack_threads())


def pack_bits(x, n, bits):
    """bits[i / 32] bit (i % 32) = (x[i] != 0), i < n; x and bits are DeviceBuffers (or views)."""
    check(lib().gl_pack_bits(_p(x), int(n), _p(bits)))


def prof_begin(max_launches, every=1):
    check(lib().gl_prof_begin(int(max_launches), int(every)))


def prof_end():
    """-> (summed kernel milliseconds, launches) of the dominant SpMV kernel since prof_begin."""
    ms, n = ctypes.c_double(0.0), ctypes.c_uint32(0)
    check(lib().gl_prof_end(ctypes.byref(ms), ctypes.byref(n)))
    return ms.value, n.value


def span_begin():
    check(lib().gl_span_begin())


def span_end():
    """-> GPU milliseconds between span_begin() and here on the library's stream (waits for the work in between)."""
    ms = ctypes.c_double(0.0)
    check(lib().gl_span_end(ctypes.byref(ms)))
    return ms.value


def sparse_nnz(buf):
    n = ctypes.c_uint32(0)
    check(lib().gl_sparse_nnz(_p(buf), ctypes.byref(n)))
    return n.value


def ewise_add(inp, out, length, val):
    check(lib().gl_ewise_add(_p(inp), _p(out), int(length), float(val)))


def assign_dense(mask, inout, length, val, mask_type):
    check(lib().gl_assign_dense(_p(mask), _p(inout), int(length), float(val), int(mask_type)))


def assign_sparse(mask, inout, val, max_entries):
    check(lib().gl_assign_sparse(_p(mask), _p(inout), float(val), int(max_entries)))


def assign_sparse_new_frontier(mask, inout, new_frontier, max_entries):
    check(lib().gl_assign_sparse_new_frontier(_p(mask), _p(inout), _p(new_frontier), int(max_entries)))


def sparse_to_dense(sparse, dense, rng, zero, max_entries):
    check(lib().gl_sparse_to_dense(_p(sparse), _p(dense), int(rng), float(zero), int(max_entries)))


def ewise_add_typed(inp, out, length, val_bits, val_type):
    check(lib().gl_ewise_add_typed(_p(inp), _p(out), int(length), int(val_bits), int(val_type)))


def assign_dense_typed(mask, inout, length, val_bits, mask_type, val_type):
    check(lib().gl_assign_dense_typed(_p(mask), _p(inout), int(length), int(val_bits), int(mask_type), int(val_type)))


def assign_sparse_typed(mask, inout, val_bits, max_entries):
    check(lib().gl_assign_sparse_typed(_p(mask), _p(inout), int(val_bits), int(max_entries)))


def assign_sparse_new_frontier_typed(mask, inout, new_frontier, max_entries, val_type):
    check(lib().gl_assign_sparse_new_frontier_typed(_p(mask), _p(inout), _p(new_frontier), int(max_entries), int(val_type)))


def sparse_to_dense_typed(sparse, dense, rng, zero_bits, max_entries):
    check(lib().gl_sparse_to_dense_typed(_p(sparse), _p(dense), int(rng), int(zero_bits), int(max_entries)))


IDX_WORD = np.dtype([("index", np.uint32), ("val", np.uint32)])   # sparse element of the integer value types
UFIXED_ONE = 1 << 24                                                 # 1.0 in ap_ufixed<32, 8>


def words_from_float(val_type, values):
    """float -> 32-bit value words, what the reference's csr_matrix_convert_from_float<val_t> does to the float matrix
    (io/data_loader.h:75-84): `unsigned` truncates toward zero, ap_ufixed<32,8,AP_RND,AP_SAT> rounds half up to 24
    fraction bits and saturates to [0, 2^32 - 1]."""
    v = np.asarray(values, dtype=np.float32).astype(np.float64)
    if val_type == GL_VAL_UNSIGNED:
        return np.clip(np.trunc(v), 0, 4294967295.0).astype(np.uint64).astype(np.uint32)
    if val_type == GL_VAL_UFIXED_32_8:
        q = np.floor(np.where(v > 0, v, 0.0) * 16777216.0 + 0.5)
        return np.clip(q, 0, 4294967295.0).astype(np.uint64).astype(np.uint32)
    raise ValueError("not an integer value type: %r" % (val_type,))


def words_to_float(val_type, words):
    w = np.asarray(words, dtype=np.uint32)
    return w.astype(np.float32) if val_type == GL_VAL_UNSIGNED else (w.astype(np.float64) / 16777216.0).astype(np.float32)


def fill_f32(buf, value, count):
    check(lib().gl_buf_fill_f32(_p(buf), float(value), int(count)))


def host_csr2csc(num_rows, num_cols, indptr, indices, data):
    """-> (csc_indptr, csc_indices, csc_data); host only."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    nnz = int(indptr[num_rows])
    o_indptr = np.empty(num_cols + 1, dtype=np.uint32)
    o_indices = np.empty(nnz, dtype=np.uint32)
    o_data = np.empty(nnz, dtype=np.float32)
    # gl_csr2csc: on the GPU when the runtime is up and the matrix is large, on the host otherwise; same output
    check(lib().gl_csr2csc(num_rows, num_cols, _np_ptr(indptr), _np_ptr(indices), _np_ptr(data),
                           ctypes.c_void_p(o_indptr.ctypes.data), _np_ptr(o_indices), _np_ptr(o_data)))
    return o_indptr, o_indices, o_data


def csr_normalize_by_outdegree(num_rows, num_cols, indptr, indices):
    """-> adj_data with data[i] = float(1.0 / #entries in the column of i) (io/data_formatter.h:36-51)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    out = np.empty(int(indptr[num_rows]), dtype=np.float32)
    check(lib().gl_csr_normalize_by_outdegree(num_rows, num_cols, _np_ptr(indptr), _np_ptr(indices), _np_ptr(out)))
    return out


def npz_load_csr(path):
    """-> (num_rows, num_cols, data f32, indices u32, indptr u32); host only, no GPU needed."""
    h = ctypes.c_void_p(0)
    nr, nc, nnz = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint64(0)
    check(lib().gl_npz_csr_open(os.fsencode(path), ctypes.byref(h), ctypes.byref(nr), ctypes.byref(nc),
                                ctypes.byref(nnz)))
    data = np.empty(nnz.value, dtype=np.float32)
    indices = np.empty(nnz.value, dtype=np.uint32)
    indptr = np.empty(nr.value + 1, dtype=np.uint32)
    check(lib().gl_npz_csr_read(h, _np_ptr(data), _np_ptr(indices), _np_ptr(indptr)))
    return nr.value, nc.value, data, indices, indptr
