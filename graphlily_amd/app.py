"""BFS / PageRank / SSSP drivers written against the module API (graphlily::app).

Same public surface and the same module call sequences as the reference drivers
  ModuleCollection  app/module_collection.h:13-114
  BFS               app/bfs.h:20-361      (pull :106-126, push :129-157, pull_push :160-219)
  PageRank          app/pagerank.h:17-160 (pull :80-90)
  SSSP              app/sssp.h:70-254     (pull :152-166, push :169-194, pull_push :197-243)
re-expressed for one process per GPU: every driver optionally takes a dist.Comm; the matrix is
then row-sharded (nnz-balanced ranges), the element-wise steps run on the owned slice, and one
all-gather per iteration rebuilds the dense vector.  With comm=None the sequence
of module calls is exactly the reference's.

Differences that do not change results:
  * modules run non-blocking and the driver synchronises only where the host needs a value
    (get_results_nnz, final read-back) -- the reference finishes the queue after every call;
  * the push->pull switch converts the frontier on the device (gl_sparse_to_dense) instead of
    round-tripping through the host (app/bfs.h:196-201);
  * vectors are allocated by the driver and bound into the modules, so the same buffers can be
    handed to the collective.
"""
import os
import time

import numpy as np

from . import capi, io
from . import module as M
from .dist import Comm, partition_rows_by_nnz


class HipBackend:
    """Allocation / transfer hooks of the drivers.  The CPU tests substitute a stand-in with the
    same methods to exercise the distributed control flow over gloo."""
    SpMVModule = M.SpMVModule
    SpMSpVModule = M.SpMSpVModule
    eWiseAddModule = M.eWiseAddModule
    AssignVectorDenseModule = M.AssignVectorDenseModule
    AssignVectorSparseModule = M.AssignVectorSparseModule

    def __init__(self, device=0, use_torch=False):
        self.device = device
        self.use_torch = use_torch

    def init(self):
        capi.init(self.device)
        if self.use_torch:
            import torch
            torch.cuda.set_device(self.device)
            capi.set_stream(torch.cuda.current_stream().cuda_stream)

    def alloc(self, count, dtype):
        """Device array of `count` elements; dtype is np.float32 or capi.IDX_VAL.  In torch mode the
        buffer carries `.tensor` (float32, or int64 with one element per (index,val) pair)."""
        itemsize = np.dtype(dtype).itemsize
        if self.use_torch:
            import torch
            t = torch.zeros(count, dtype=torch.float32 if itemsize == 4 else torch.int64,
                            device="cuda:%d" % self.device)
            buf = capi.DeviceBuffer.from_torch(t)
            buf.tensor = t
            return buf
        buf = capi.DeviceBuffer(count * itemsize)
        buf.tensor = None
        return buf

    def view(self, buf, first, count, itemsize):
        v = capi.DeviceBuffer(count * itemsize, ptr=buf.ptr + first * itemsize, keepalive=buf)
        v.tensor = None
        return v

    def upload(self, buf, arr):
        buf.write(np.ascontiguousarray(arr))

    def download(self, buf, dtype, count):
        return buf.read(dtype, count)

    def download_result(self, buf, count):
        """Final float32 read-back of a driver: a fresh host array per call, like the by-value return
        of the reference's send_*_device_to_host.  (A page-locked destination -- capi.pinned_empty +
        read(out=) -- halves the copy time but must be reused across calls to pay off; measured: pinning
        12 MB per call costs more than it saves.)"""
        return buf.read(np.float32, count)

    def copy(self, dst, src, nbytes):
        capi.copy_d2d(dst, src, nbytes)

    def fill(self, buf, value, count):
        capi.fill_f32(buf, value, count)

    def sparse_to_dense(self, sparse, dense, rng, zero, max_entries):
        capi.sparse_to_dense(sparse, dense, rng, zero, max_entries)

    def sync(self):
        capi.sync()


class ModuleCollection:
    def __init__(self):
        self.modules_ = []
        self.target_ = "hw"

    def add_module(self, module):
        self.modules_.append(module)

    def set_target(self, target):
        assert target in ("sw_emu", "hw_emu", "hw")  # module_collection.h:56-59
        self.target_ = target

    def set_up_runtime(self, xclbin_file_path=None):
        """One device context for all modules (module_collection.h:69-114); the bitstream path is
        accepted for signature parity and ignored."""
        self.backend.init()
        for m in self.modules_:
            m.blocking = False


class _GraphApp(ModuleCollection):
    def __init__(self, num_channels, comm, backend):
        super().__init__()
        self.num_channels_ = num_channels
        self.comm = comm if comm is not None else Comm(None)
        self.backend = backend if backend is not None else HipBackend()
        self.bounds_ = None

    def _pad(self, csr):
        d = self.num_channels_ * M.pack_size      # app/bfs.h:86-89
        io.util_round_csr_matrix_dim(csr, d, d)

    def _load(self, src):
        return io.load_csr_matrix_from_float_npz(src) if isinstance(src, (str, bytes)) else src.copy()

    def _shard(self, csr):
        self.bounds_ = partition_rows_by_nnz(csr.adj_indptr, self.comm.world_size)
        self.r0_, self.r1_ = self.bounds_[self.comm.rank], self.bounds_[self.comm.rank + 1]

    def get_nnz(self):
        return self.SpMV_.get_nnz()

    # slice helpers -----------------------------------------------------------------------------
    def _own(self, buf):
        return self.backend.view(buf, self.r0_, self.r1_ - self.r0_, 4)

    def _gather(self, buf):
        if self.comm.distributed:
            self.comm.all_gather_slices(buf.tensor, self.bounds_)

    def _new_dense(self, n, fill, source=None, source_value=None):
        """Dense vector built on the device: a fill kernel plus (optionally) one 4-byte store, instead of
        the reference's full host-side vector + upload (e.g. app/bfs.h:107-112)."""
        B = self.backend
        buf = B.alloc(n, np.float32)
        B.fill(buf, float(fill), n)
        if source is not None:
            B.fill(B.view(buf, source, 1, 4), float(source_value), 1)   # a 1-element fill kernel: no blocking copy
        return buf

    def _gather_sparse(self, local_buf, out_buf, n, head_val):
        """All ranks contribute their (ascending, disjoint) slice of a sparse vector; every rank ends
        up with the concatenation, head {total, head_val} included.  Returns the total count."""
        B = self.backend
        cnt = self.comm_sparse_count(local_buf)
        if not self.comm.distributed:
            return cnt
        total = self.comm.all_gather_sparse(local_buf.tensor[1:], cnt, n, out_buf.tensor[1:])
        head = np.zeros(1, dtype=capi.IDX_VAL)
        head["index"][0], head["val"][0] = total, head_val
        B.upload(B.view(out_buf, 0, 1, 8), head)
        return total

    def comm_sparse_count(self, buf):
        return int(self.backend.download(buf, capi.IDX_VAL, 1)["index"][0])


class BFS(_GraphApp):
    def __init__(self, num_channels=M.num_hbm_channels, spmv_out_buf_len=0, spmspv_out_buf_len=0, vec_buf_len=0,
                 comm=None, backend=None):
        super().__init__(num_channels, comm, backend)
        B = self.backend
        self.semiring_ = M.LogicalSemiring
        self.SpMV_ = B.SpMVModule(num_channels, spmv_out_buf_len, vec_buf_len)
        self.SpMV_.set_semiring(self.semiring_)
        self.SpMV_.set_mask_type(M.kMaskWriteToZero)
        self.DenseAssign_ = B.AssignVectorDenseModule()
        self.DenseAssign_.set_mask_type(M.kMaskWriteToOne)
        self.SpMSpV_ = B.SpMSpVModule(spmspv_out_buf_len)
        self.SpMSpV_.set_semiring(self.semiring_)
        self.SpMSpV_.set_mask_type(M.kMaskWriteToZero)
        self.SparseAssign_ = B.AssignVectorSparseModule(False)
        self.eWiseAdd_ = B.eWiseAddModule()
        for m in (self.SpMV_, self.DenseAssign_, self.SpMSpV_, self.SparseAssign_, self.eWiseAdd_):
            self.add_module(m)

    def load_and_format_matrix(self, csr_float_npz_path, skip_empty_rows=True):
        csr = self._load(csr_float_npz_path)
        self._pad(csr)
        csr.adj_data = np.ones(csr.nnz, dtype=np.float32)       # app/bfs.h:90
        csc = io.csr2csc(csr)
        self._shard(csr)
        for m in (self.SpMV_, self.SpMSpV_):
            m.set_row_shard(self.r0_, self.r1_)
        self.SpMV_.load_and_format_matrix(csr, skip_empty_rows)
        self.SpMSpV_.load_and_format_matrix(csc)
        self.n_ = self.SpMV_.get_num_rows()
        assert self.n_ == self.SpMV_.get_num_cols()
        # the bit-frontier schedule takes its decisions from GLOBAL lengths (gl_bfs_bits_shard_step):
        # every rank keeps the two n-word arrays of the whole matrix
        self.row_len_ = np.diff(csr.adj_indptr.astype(np.int64)).astype(np.uint32)
        self.col_len_ = np.diff(csc.adj_indptr.astype(np.int64)).astype(np.uint32)
        self.nnz_global_ = int(csr.adj_indptr[csr.num_rows])

    def send_matrix_host_to_device(self):
        self.SpMV_.send_matrix_host_to_device()
        self.SpMSpV_.send_matrix_host_to_device()
        if hasattr(self.SpMSpV_, "attach_pull"):
            self.SpMSpV_.attach_pull(self.SpMV_)     # heavy frontiers of a push iteration go row-wise
        self.results_ = self.bits_a_ = self.bits_b_ = None   # per-matrix scratch of the pull loop
        # the device-resident schedules hold buffers sized for the old n and hipGraphs with the old plans' device
        # pointers baked in: a matrix sent again (or another one) must rebuild them
        self.bits_loop_ = None

    # -- pull ------------------------------------------------------------------------------------
    def _bind_pull(self, vector, distance):
        B, n = self.backend, self.n_
        results = getattr(self, "results_", None)     # scratch of the unfused path, kept across calls
        if results is None:
            results = self.results_ = B.alloc(n, np.float32)
        self.SpMV_.bind_vector_buf(vector)
        self.SpMV_.bind_mask_buf(distance)
        self.SpMV_.bind_results_buf(results)
        self.DenseAssign_.bind_mask_buf(self._own(vector))
        self.DenseAssign_.bind_inout_buf(self._own(distance))
        self.eWiseAdd_.bind_in_buf(self._own(results))
        self.eWiseAdd_.bind_out_buf(self._own(vector))
        # The frontier lives as BITS wherever the boolean SpMV layout allows it: row-sharded runs then exchange n/8
        # bytes per iteration instead of 4n, and on unsplit plans the whole pull iteration (masked SpMV, eWiseAdd,
        # dense assign, packing of the next frontier) is ONE launch (gl_bfs_pull_step).  Slices are whole 64-bit
        # words (shard bounds are 64-aligned).
        self.bits_ = self.bits_next_ = None
        self.fused_ = False
        self.distance_ = distance
        words = self.SpMV_.bits_words() if hasattr(self.SpMV_, "bits_words") else 0
        aligned = (not self.comm.distributed) or all(b % 64 == 0 for b in self.bounds_)
        if words and aligned:
            self.fused_ = hasattr(self.SpMV_, "fused_bfs_ok") and self.SpMV_.fused_bfs_ok()
            if self.fused_ or self.comm.distributed:
                # opaque 32-bit words, kept across calls: pack_bits / the fused step rewrite every word of the row
                # range, words past it stay 0 from the allocation
                if getattr(self, "bits_a_", None) is None:
                    self.bits_a_, self.bits_b_ = B.alloc(words, np.float32), B.alloc(words, np.float32)
                    B.fill(self.bits_a_, 0.0, words)
                    B.fill(self.bits_b_, 0.0, words)
                self.bits_ = self.bits_a_
                capi.pack_bits(vector, n, self.bits_)
            if self.fused_:
                self.bits_next_ = self.bits_b_

    def _pull_iteration(self, vector, it):
        B, own = self.backend, self.r1_ - self.r0_
        if self.fused_:
            self.SpMV_.bfs_pull_step(self.bits_, self.bits_next_, self.distance_, float(it + 1))
            self.bits_, self.bits_next_ = self.bits_next_, self.bits_
            if self.comm.distributed:
                self.comm.all_gather_slices(self.bits_.tensor, [b // 32 for b in self.bounds_])
            return
        if self.bits_ is not None:
            self.SpMV_.run_bits(self.bits_)
        else:
            self.SpMV_.run()
        self.eWiseAdd_.run(own, 0.0)                 # results -> vector (app/bfs.h:119-122)
        self.DenseAssign_.run(own, float(it + 1))
        if self.bits_ is not None:
            capi.pack_bits(self._own(vector), own, B.view(self.bits_, self.r0_ // 32, own // 32, 4))
            self.comm.all_gather_slices(self.bits_.tensor, [b // 32 for b in self.bounds_])
        else:
            self._gather(vector)

    def _finish_distance(self, distance):
        self._gather(distance)
        self.backend.sync()
        return self.backend.download_result(distance, self.n_)

    def pull(self, source, num_iterations):
        if self._bits_loop_ok():
            return self._pull_push_bits(source, num_iterations, -1.0, pull_only=True)
        n = self.n_
        vector = self._new_dense(n, self.semiring_.zero, source, 1.0)
        distance = self._new_dense(n, 0.0, source, 1.0)
        self._bind_pull(vector, distance)
        for it in range(1, num_iterations + 1):
            self._pull_iteration(vector, it)
        return self._finish_distance(distance)

    # -- push ------------------------------------------------------------------------------------
    def _start_push(self, source):
        B, n = self.backend, self.n_
        frontier = B.alloc(n + 1, capi.IDX_VAL)
        B.upload(B.view(frontier, 0, 2, 8), M.make_sparse_vec([source], [1.0]))
        self._hint_frontier(1)
        distance = self._new_dense(n, 0.0, source, 1.0)
        local = B.alloc(n + 1, capi.IDX_VAL)        # this rank's slice of the next frontier
        self.SpMSpV_.bind_vector_buf(frontier)
        self.SpMSpV_.bind_mask_buf(distance)
        self.SpMSpV_.results_buf = local
        self.SparseAssign_.bind_mask_buf(local)
        self.SparseAssign_.bind_inout_buf(distance)
        return frontier, distance, local

    def _hint_frontier(self, nnz):
        if hasattr(self.SpMSpV_, "hint_vector_nnz"):
            self.SpMSpV_.hint_vector_nnz(nnz)

    def _push_iteration(self, frontier, local, it):
        """SpMSpV on the shard, mark the newly reached vertices, publish the next frontier.  Returns
        (frontier size, frontier buffer, result buffer) for the next iteration."""
        # SpMSpV, then distance[new] = level (app/bfs.h:146-148); the assign rides on the pass that writes the results
        if hasattr(self.SpMSpV_, "run_assign"):
            self.SpMSpV_.run_assign(self.SparseAssign_.inout_buf, float(it + 1))
        else:
            self.SpMSpV_.run()
            self.SparseAssign_.run(float(it + 1))
        if self.comm.distributed:
            total = self._gather_sparse(local, frontier, self.n_, self.semiring_.zero)
            self._hint_frontier(total)
            return total, frontier, local
        nnz = self.SpMSpV_.get_results_nnz()
        # the reference copies results -> vector here (app/bfs.h:149-152); the two buffers swap roles instead
        frontier, local = local, frontier
        self.SpMSpV_.bind_vector_buf(frontier)
        self.SpMSpV_.results_buf = local
        self.SparseAssign_.bind_mask_buf(local)
        self._hint_frontier(nnz)
        return nnz, frontier, local

    def push(self, source, num_iterations):
        frontier, distance, local = self._start_push(source)
        for it in range(1, num_iterations + 1):
            _, frontier, local = self._push_iteration(frontier, local, it)
        return self._finish_distance(distance)

    # -- pull / pull_push without the host in the loop ------------------------------------------
    def _pull_push_bits(self, source, num_iterations, threshold, pull_only=False):
        """The reference decides push vs pull on the host from a count it reads back every iteration (app/bfs.h:180-190) and
        converts the frontier on the host at the switch (:195-205).  Here the WHOLE run is enqueued up front as a device-resident
        schedule with the frontier as BITS only (gl_bfs_bits_shard_step, csrc/gl_bfs_shard.h): ONE launch per iteration slot --
        it begins with the previous slot's decision (the reference's loop condition, replayed from the slot's tallies) and then
        runs the step the state asks for: the scattering push straight into the next frontier's bit vector, the streaming
        (||,&&) pull, which also serves a push whose frontier is heavy, or the bottom-up scan of the rows not reached yet.
        Slot s reads bit vector s and writes vector s + 1.  No synchronisation and no device->host copy between the first
        launch and the read-back; the schedule depends only on (iterations, threshold) -- the source is a device word -- so it
        is recorded once as a hipGraph and replayed.  (pull_only: BFS.pull, app/bfs.h:106-126 -- no slot scatters.)

        ROW SHARDS (comm.distributed) run the same launches on their rows; ONE all-gather per slot rebuilds the slot's bit
        vector (n/8 bytes) and carries every rank's tallies (256 bytes each), from which every rank takes the same decision;
        every rank reads back ITS SLICE of the distances (`gather_result_` = True all-gathers them first).

        Rounds 2-3 kept four earlier generations of this loop alive behind switches (a list-based gated schedule, two- and
        three-launch slots); round 4 retired them: same-box A/B in profiles/r04_ab_schedules.txt.  What remains beside this
        schedule is the reference's own module-call loop (GRAPHLILY_BFS_HOST_LOOP=1, and wherever the plans do not offer
        the bit layout)."""
        B, n, N = self.backend, self.n_, num_iterations
        self.fused_ = True          # (every pull step of this schedule is the fused one, see _bind_pull)
        sharded = self.comm.distributed
        st = getattr(self, "bits_loop_", None)
        if st is None or st["N"] < N:
            words = (int(self.SpMV_.bits_words()) + 3) & ~3
            nvec, ctl_words = N + 2, (18 + 2 * N + 15) & ~15
            both = B.alloc(n + ctl_words, np.float32)          # distances, then the control words: one read-back fetches both
            # row shards: the ranks' tallies (gl_bfs_bits_shard_step) live behind the bit vectors and are cleared with them
            tally_words = capi.bfs_tally_words(N, self.comm.world_size)
            nvec_all = nvec + (tally_words + words - 1) // words
            vecs = B.alloc(nvec_all * words, np.float32)
            st = self.bits_loop_ = {"N": N, "both": both, "vecs": vecs, "words": words, "nvec": nvec, "ctl_words": ctl_words,
                                    "ctl": B.view(both, n, ctl_words, 4), "distance": B.view(both, 0, n, 4),
                                    "bits": [B.view(vecs, k * words, words, 4) for k in range(nvec)], "graphs": {},
                                    "src": np.zeros(1, np.uint32), "warm": set(), "nvec_all": nvec_all}
            st["col_len"] = capi.DeviceBuffer.from_host(self.col_len_)
            st["tally"] = B.view(vecs, nvec * words, tally_words, 4)
            if sharded:
                st["row_len"] = capi.DeviceBuffer.from_host(self.row_len_)
        ctl, distance, bits, words = st["ctl"], st["distance"], st["bits"], st["words"]
        # Once the reference's rule has switched to pulling (frontier / n >= threshold, app/bfs.h:180-190), every later slot is
        # handed back to the push step (an extension: the reference pulls to the end), which leaves heavy frontiers to the
        # streaming pull anyway: from then on the direction follows the work.  Distances do not depend on the direction.
        back = 0.0 if pull_only else 1.0
        csc_plan, pull_plan = self.SpMSpV_.plan_, self.SpMV_.plan_
        # (one-GPU emulation of a rank without an exchange step, dist.EmulatedComm: the slots read the whole run's vectors)
        gathered = bits
        if sharded and getattr(self.comm, "emulated", False) and not self.comm.copy:
            gathered = [self.comm.truth_vector(k) for k in range(st["nvec"])]
        rank, world = self.comm.rank, self.comm.world_size
        tally, tally_in = st.get("tally"), None
        if getattr(self.comm, "emulated", False):
            table = self.comm.truth_tally((source, N), N, self.bounds_, self.col_len_, self.row_len_, n)
            tally_in = None if self.comm.copy else table

        def may_of(it):
            return (1 if it + 1 < N else 0) | (2 if it + 1 <= N else 0)

        def schedule():
            capi.bfs_bits_begin(ctl, st["ctl_words"], distance, n, st["vecs"], words, st["nvec_all"], 0 if pull_only else 0xffffffff)
            for it in range(1, N + 1):
                capi.bfs_bits_shard_step(csc_plan, pull_plan, gathered[it], bits[it + 1], words, distance, float(it + 1), ctl, tally,
                                         tally_in, it, rank, world, st["col_len"], self.nnz_global_, threshold,
                                         may_of(it - 1) if it > 1 else 0, back)
                if sharded:
                    self._exchange_bits(st, it + 1, it)
            capi.bfs_bits_shard_finish(csc_plan, pull_plan, ctl, tally, tally_in, N, rank, world, self.nnz_global_, threshold, may_of(N), back)

        # Levels are small integers: when they fit a byte (N + 1 <= 255; a nibble up to 14 iterations) the result crosses PCIe
        # PACKED -- 1.5 or 3 MB instead of 12 MB on orkut, 28 or 55 us instead of 225, the control words behind them in the
        # same copy -- and a few host threads turn them into the floats the caller gets (woken while the GPU is still busy).
        # GRAPHLILY_BFS_U8=0: the floats themselves.  The pack and the copy kernel are part of the recorded schedule (fixed
        # buffers) unless the bench wants the schedule's GPU time alone, or the distances are all-gathered first.
        cw = st["ctl_words"]
        timed = getattr(self, "time_schedule_", False)      # bench: GPU time of the schedule without the read-back
        sliced = sharded and not getattr(self, "gather_result_", True)
        lo, hi = (self.r0_, self.r1_) if sliced else (0, n)
        own = hi - lo
        pbits = 4 if N + 1 <= 15 else 8
        # (from half a million rows on: same-box A/B, profiles/r04_ab_schedules.txt -- the googleplus stand-in's 108 K levels are
        # 0.43 MB as floats and came back 24 us SOONER unpacked, ogbl-ppa's 576 K tie, hollywood's 1 M gain 22 us packed)
        can_pack = (N + 1 <= 255 and own % 8 == 0 and lo % 4 == 0 and own >= (1 << 19)
                    and os.environ.get("GRAPHLILY_BFS_U8", "1") != "0" and capi.host_unpack_threads() >= 4)
        # The packed read-back's second half runs on HOST threads: on a busy box it loses to the plain float copy (round 3:
        # 0.39 - 0.52 ms for the same call over the round's boxes; same-box, same-process spread 0.39 - 0.55).  So the driver
        # MEASURES: both ways are timed (whole call), the faster one by the MEDIAN of its last seven calls is used, and every
        # 32nd call tries the other one again.  The books are kept per schedule (pull and pull_push each have their own: round 4
        # kept one exponential average for both, and the first calls of the second mode -- which enqueue and record its graph,
        # 12 ms -- poisoned the average of whichever way was active: the mode measured second ran on the slower way, 33 %
        # apart between two legs of one bench process).  GRAPHLILY_BFS_U8=0 / =2 pin the float / packed way.
        rb = st.setdefault("readback", {}).setdefault((N, float(threshold), back, pull_only, lo, own),
                                                      {"packed": None, "float": None, "calls": 0, "t_packed": [], "t_float": []})
        pin = os.environ.get("GRAPHLILY_BFS_U8", "1")
        if not can_pack or timed:
            as_bytes = can_pack
        elif pin == "2" or rb["packed"] is None:
            as_bytes = True
        elif rb["float"] is None:
            as_bytes = rb["calls"] < 8        # (the first calls warm up and record the packed way's graph)
        else:
            better = rb["packed"] <= rb["float"]
            as_bytes = better if rb["calls"] % 32 != 31 else not better
        t_call = time.perf_counter()
        in_graph = as_bytes and not timed and (sliced or not sharded)
        # Round 6: the pack kernel stores into the page-locked block itself, chunk by chunk with a flag behind each, and the host
        # threads expand a chunk as soon as it has landed -- the expansion overlaps the PCIe transfer (GRAPHLILY_BFS_STREAM=0: the
        # round-5 way, pack -> copy -> wait -> expand)
        streamed = as_bytes and os.environ.get("GRAPHLILY_BFS_STREAM", "1") != "0"
        if as_bytes:
            pw = capi.levels_packed_words(own, pbits)
            if st.get("lev8_key") != (own, pbits, streamed):
                st["lev8_key"] = (own, pbits, streamed)
                st["lev8"] = None if streamed else capi.DeviceBuffer(4 * (pw + cw))
                st["h8"] = capi.pinned_empty(capi.levels_stream_bytes(own, pbits, cw) if streamed else 4 * (pw + cw), np.uint8)
                st["graphs"] = {k: v for k, v in st["graphs"].items() if not k[5]}   # (graphs that recorded the old buffers)
            if streamed:
                capi.levels_stream_arm(st["h8"], own, pbits, cw)      # (before anything of this run is launched)

        def packed_read_back():
            if streamed:
                capi.levels_pack_stream(B.view(distance, lo, own, 4), own, pbits, ctl, cw, st["h8"])
                return
            capi.levels_pack(B.view(distance, lo, own, 4), own, pbits, ctl, cw, st["lev8"])   # (levels, then the control words)
            st["lev8"].read_async(st["h8"])

        capi.fill_u32(B.view(ctl, 2, 1, 4), int(source), 1)   # ctl[2] = source (see _pull_push_device)
        key = (N, float(threshold), back, pull_only, True, in_graph, lo, own)
        g = st["graphs"].get(key)
        # (a torch.distributed collective is not recorded by the library's capture: those runs are enqueued call by call)
        capturable = not sharded or getattr(self.comm, "capturable", False)
        if g is None and capturable and key in st["warm"]:
            try:
                with capi.Graph.capture() as g:
                    schedule()
                    if in_graph:
                        packed_read_back()
                st["graphs"][key] = g
            except capi.GraphLilyError as e:
                g = st["graphs"][key] = False              # capture not possible here: keep enqueueing
                st["graph_error"] = str(e)
        if timed:
            capi.span_begin()
        if g:
            g.launch()
        else:
            schedule()
            st["warm"].add(key)
        if timed:
            self.schedule_ms_ = capi.span_end()
        if sharded and not sliced:
            self.comm.all_gather_slices(st["both"].tensor[:n] if st["both"].tensor is not None else st["both"], self.bounds_)
        if as_bytes:
            if not (g and in_graph):
                packed_read_back()
            res = capi.pinned_recycled(own, np.float32)       # (recycled: already paged in)
            if streamed:
                c = capi.sync_levels_unpack_stream(res, st["h8"], own, pbits, cw)   # (the host threads start now: chunk by chunk)
            else:
                capi.sync_levels_unpack(res, st["h8"], own, pbits)   # (the host threads start now and spin until the stream is done)
                c = st["h8"][4 * pw:].view(np.uint32).copy()
        else:
            # the distances (this rank's slice of them) + the control words: two copies behind the schedule, one wait
            out = capi.pinned_recycled(own + cw, np.float32)
            if own == n:
                st["both"].read_async(out)                    # (the control words follow the distances: one copy)
            else:
                st["both"].read_async(out[:own], 4 * lo)
                st["both"].read_async(out[own:], 4 * n)
            B.sync()
            res, c = out[:own], out[own:].view(np.uint32)
        if can_pack and not timed:
            # (the first two calls of a way enqueue / record its schedule: not what the steady state costs)
            way, dt = ("packed" if as_bytes else "float"), time.perf_counter() - t_call
            seen = rb.setdefault("n_" + way, 0)
            rb["n_" + way] = seen + 1
            if seen >= 2:
                ts = rb["t_" + way]
                ts.append(dt)
                del ts[:-7]
                rb[way] = float(np.median(ts))
            rb["calls"] += 1
            self.readback_ = {"way": way, "packed_ms": None if rb["packed"] is None else round(rb["packed"] * 1e3, 4),
                              "float_ms": None if rb["float"] is None else round(rb["float"] * 1e3, 4)}
        self.result_range_ = (lo, hi)
        self.push_iterations_ = int(c[1])          # the reference's count (first push phase)
        self.push_iterations_again_ = int(c[3])    # pushes after a pull step handed back
        S = (cw - 16) // 2
        self.bfs_slot_counts_ = c[17:17 + N].copy()             # vertices reached per slot
        self.bfs_slot_modes_ = c[17 + S:17 + S + N].copy()      # 1 scattered, 2 streamed row-wise, 3 bottom-up, 0 nothing ran
        return res

    def _exchange_bits(self, st, k, tally_slot):
        """The one exchange step of a sharded slot: every rank's rows of bit vector k to every rank, and every rank's tallies
        of slot `tally_slot` (256 bytes each) along with them."""
        comm, W = self.comm, self.comm.world_size
        per = capi.GL_BFS_TALLY_RANK_WORDS
        first = capi.GL_BFS_TALLY_HEAD_WORDS + (tally_slot - 1) * W * per
        if hasattr(comm, "exchange_bits"):                 # the C-ABI communicator (gl_dist_*) or the one-GPU emulation
            comm.exchange_bits(st["bits"][k], k, self.bounds_, self.backend.view(st["tally"], first, W * per, 4), tally_slot)
            return
        t = st["vecs"].tensor[k * st["words"]:(k + 1) * st["words"]]
        base = st["nvec"] * st["words"] + first               # ONE collective per slot: the tallies ride behind the rank's bits
        comm.all_gather_slices_with_tail(t, [b // 32 for b in self.bounds_], st["vecs"].tensor[base:base + W * per], per)

    def _bits_loop_ok(self):
        """Can this BFS run as the device-resident schedule?  (GRAPHLILY_BFS_HOST_LOOP=1: no -- the reference's loop.)"""
        if os.environ.get("GRAPHLILY_BFS_HOST_LOOP", "0") != "0" or not hasattr(capi, "bfs_bits_shard_step"):
            return False
        if getattr(self.SpMV_, "plan_", None) is None or getattr(self.SpMSpV_, "plan_", None) is None:
            return False
        if not (hasattr(self.SpMV_, "bits_words") and self.SpMV_.bits_words() > 0 and self.SpMV_.semiring_.zero == 0.0):
            return False
        if self.comm.distributed:      # any boolean plan whose row range is cut on multiples of 64 rows
            return all(b % 64 == 0 for b in self.bounds_[:-1])
        return hasattr(self.SpMV_, "fused_bfs_ok") and self.SpMV_.fused_bfs_ok()

    def pull_push(self, source, num_iterations, threshold=0.05):
        if self._bits_loop_ok():
            return self._pull_push_bits(source, num_iterations, threshold)
        n = self.n_
        frontier, distance, local = self._start_push(source)
        it = 1
        while True:
            nnz, frontier, local = self._push_iteration(frontier, local, it)
            it += 1
            if not (it < num_iterations and float(nnz) / n < threshold):
                break
        self.push_iterations_ = it - 1
        # switch: the frontier becomes the dense SpMV input (app/bfs.h:195-205), on the device
        vector = self.backend.alloc(n, np.float32)
        self.backend.sparse_to_dense(frontier, vector, n, M.LogicalSemiring.zero, n)
        self._bind_pull(vector, distance)
        while it <= num_iterations:
            self._pull_iteration(vector, it)
            it += 1
        return self._finish_distance(distance)


    def pull_push_time_breakdown(self, source, num_iterations, threshold=0.05):
        """app/bfs.h:222-347: pull_push with the reference's four wall-clock buckets.  Every step is followed by
        a device sync (the reference's module calls are blocking), so the total is larger than pull_push's.
        Returns the same distance vector; the buckets (ms) are left in `time_breakdown_` and printed."""
        B, n = self.backend, self.n_
        tb = {"spmv_spmspv": 0.0, "assign": 0.0, "data_transfer": 0.0}

        def timed(bucket, fn):
            B.sync()
            t0 = time.perf_counter()
            r = fn()
            B.sync()
            tb[bucket] += (time.perf_counter() - t0) * 1e3
            return r

        B.sync()
        t_start = time.perf_counter()
        frontier, distance, local = timed("data_transfer", lambda: self._start_push(source))
        it = 1
        while True:
            timed("spmv_spmspv", self.SpMSpV_.run)
            timed("assign", lambda: self.SparseAssign_.run(float(it + 1)))
            if self.comm.distributed:
                nnz = timed("data_transfer", lambda: self._gather_sparse(local, frontier, n, self.semiring_.zero))
            else:
                nnz = self.SpMSpV_.get_results_nnz()
                timed("data_transfer", lambda: B.copy(frontier, local, 8 * (1 + nnz)))
            self._hint_frontier(nnz)
            it += 1
            if not (it < num_iterations and float(nnz) / n < threshold):
                break
        self.push_iterations_ = it - 1
        print("SpMSpV runs for %d iterations" % (it - 1))
        vector = B.alloc(n, np.float32)
        timed("data_transfer", lambda: B.sparse_to_dense(frontier, vector, n, M.LogicalSemiring.zero, n))
        self._bind_pull(vector, distance)
        own = self.r1_ - self.r0_
        while it <= num_iterations:
            timed("spmv_spmspv", self.SpMV_.run)
            timed("data_transfer", lambda: (self.eWiseAdd_.run(own, 0.0), self._gather(vector)))
            timed("assign", lambda: self.DenseAssign_.run(own, float(it + 1)))
            it += 1
        result = timed("data_transfer", lambda: self._finish_distance(distance))
        total = (time.perf_counter() - t_start) * 1e3
        tb["total"] = total
        tb["overhead"] = total - tb["spmv_spmspv"] - tb["assign"] - tb["data_transfer"]
        self.time_breakdown_ = tb
        for k in ("total", "spmv_spmspv", "assign", "data_transfer", "overhead"):
            print("%s_time_ms: %.4f" % (k, tb[k]))
        return result


class PageRank(_GraphApp):
    def __init__(self, num_channels=M.num_hbm_channels, spmv_out_buf_len=0, vec_buf_len=0, comm=None, backend=None):
        super().__init__(num_channels, comm, backend)
        B = self.backend
        self.semiring_ = M.ArithmeticSemiring
        self.SpMV_ = B.SpMVModule(num_channels, spmv_out_buf_len, vec_buf_len)
        self.SpMV_.set_semiring(self.semiring_)
        self.SpMV_.set_mask_type(M.kNoMask)
        self.eWiseAdd_ = B.eWiseAddModule()
        self.add_module(self.SpMV_)
        self.add_module(self.eWiseAdd_)

    def load_and_format_matrix(self, csr_float_npz_path, damping, skip_empty_rows=True):
        csr = self._load(csr_float_npz_path)
        self._pad(csr)
        io.util_normalize_csr_matrix_by_outdegree(csr)
        csr.adj_data = (csr.adj_data * np.float32(damping)).astype(np.float32)   # app/pagerank.h:67
        self._shard(csr)
        self.SpMV_.set_row_shard(self.r0_, self.r1_)
        self.SpMV_.load_and_format_matrix(csr, skip_empty_rows)
        self.n_ = self.SpMV_.get_num_rows()
        assert self.n_ == self.SpMV_.get_num_cols()

    def send_matrix_host_to_device(self):
        self.SpMV_.send_matrix_host_to_device()

    def pull(self, damping, num_iterations):
        B, n = self.backend, self.n_
        # rank starts at float(1.0 / n) over the PADDED n (app/pagerank.h:81), teleport is the float
        # expression (1 - damping) / n (:87)
        vector = self._new_dense(n, np.float32(1.0 / n))
        teleport = np.float32(np.float32(1) - np.float32(damping)) / np.float32(n)
        results = B.alloc(n, np.float32)
        # The reference runs SpMV with zero = 0 and then eWiseAdd(results, teleport) -> vector (app/pagerank.h:84-88).
        # The device epilogue computes zero + sum in float, so passing the teleport term AS the semiring's zero gives
        # the same float (a + b == b + a), and the two buffers swap roles instead of being copied: one launch and
        # 8n bytes less per iteration.
        if getattr(self.SpMV_, "plan_flags_", 0) & capi.GL_PLAN_REFERENCE_ORDER:
            # diagnostic run in the reference's own evaluation order: the literal module sequence, SpMV with zero = 0
            # and then eWiseAdd(teleport) (teleport + sum started from the teleport term rounds differently)
            self.SpMV_.bind_vector_buf(vector)
            self.SpMV_.bind_results_buf(results)
            self.eWiseAdd_.bind_in_buf(self._own(results))
            self.eWiseAdd_.bind_out_buf(self._own(vector))
            for _ in range(num_iterations):
                self.SpMV_.run()
                self.eWiseAdd_.run(self.r1_ - self.r0_, float(teleport))
                self._gather(vector)
            B.sync()
            return B.download_result(vector, n)
        saved = self.SpMV_.semiring_
        self.SpMV_.set_semiring(M.SemiringType(M.kMulAdd, 1.0, float(teleport)))
        # every result IS the next vector, untouched in between: the run's epilogue prepares the next run's packed x (one GPU)
        chained = (not self.comm.distributed) and hasattr(self.SpMV_, "chain") and self.SpMV_.chain(True)
        try:
            for _ in range(num_iterations):
                self.SpMV_.bind_vector_buf(vector)
                self.SpMV_.bind_results_buf(results)
                self.SpMV_.run()
                self._gather(results)
                vector, results = results, vector
        finally:
            if chained:
                self.SpMV_.chain(False)
            self.SpMV_.set_semiring(saved)
        B.sync()
        return B.download_result(vector, n)


    def pull_time_breakdown(self, damping, num_iterations):
        """app/pagerank.h:93-147: pull with per-iteration wall-clock buckets (spmv, ewise, data transfer); every
        step is followed by a device sync.  Returns the rank vector (the reference returns the wrong buffer
        there); the buckets (ms per iteration) are left in `time_breakdown_` and printed."""
        B, n = self.backend, self.n_
        tb = {"spmv": 0.0, "ewise": 0.0, "data_transfer": 0.0}

        def timed(bucket, fn):
            B.sync()
            t0 = time.perf_counter()
            r = fn()
            B.sync()
            tb[bucket] += (time.perf_counter() - t0) * 1e3
            return r

        B.sync()
        t_start = time.perf_counter()
        vector = timed("data_transfer", lambda: self._new_dense(n, np.float32(1.0 / n)))
        teleport = np.float32(np.float32(1) - np.float32(damping)) / np.float32(n)
        results = B.alloc(n, np.float32)
        self.SpMV_.bind_vector_buf(vector)
        self.SpMV_.bind_results_buf(results)
        self.eWiseAdd_.bind_in_buf(self._own(results))
        self.eWiseAdd_.bind_out_buf(self._own(vector))
        own = self.r1_ - self.r0_
        for _ in range(num_iterations):
            timed("spmv", self.SpMV_.run)
            timed("ewise", lambda: self.eWiseAdd_.run(own, float(teleport)))
            timed("data_transfer", lambda: self._gather(vector))
        result = timed("data_transfer", lambda: B.download_result(vector, n))
        tb["total"] = (time.perf_counter() - t_start) * 1e3
        self.time_breakdown_ = {k: v / num_iterations for k, v in tb.items()}
        for k in ("total", "spmv", "ewise", "data_transfer"):
            print("%s_time_ms per iteration: %.4f" % (k, self.time_breakdown_[k]))
        return result


class SSSP(_GraphApp):
    def __init__(self, num_channels=M.num_hbm_channels, spmv_out_buf_len=0, spmspv_out_buf_len=0, vec_buf_len=0,
                 comm=None, backend=None, semiring=M.TropicalSemiring):
        super().__init__(num_channels, comm, backend)
        B = self.backend
        self.semiring_ = semiring
        self.SpMV_ = B.SpMVModule(num_channels, spmv_out_buf_len, vec_buf_len)
        self.SpMV_.set_semiring(self.semiring_)
        self.SpMV_.set_mask_type(M.kNoMask)
        self.SpMSpV_ = B.SpMSpVModule(spmspv_out_buf_len)
        self.SpMSpV_.set_semiring(self.semiring_)
        self.SpMSpV_.set_mask_type(M.kNoMask)
        self.SparseAssign_ = B.AssignVectorSparseModule(True)
        self.eWiseAdd_ = B.eWiseAddModule()
        for m in (self.SpMV_, self.SpMSpV_, self.SparseAssign_, self.eWiseAdd_):
            self.add_module(m)

    def load_and_format_matrix(self, csr_float_npz_path, skip_empty_rows=True):
        csr = self._load(csr_float_npz_path)
        io.sssp_add_self_edges(csr)                  # app/sssp.h:132 (_preprocess)
        self._pad(csr)
        csc = io.csr2csc(csr)
        self._shard(csr)
        for m in (self.SpMV_, self.SpMSpV_):
            m.set_row_shard(self.r0_, self.r1_)
        self.SpMV_.load_and_format_matrix(csr, skip_empty_rows)
        self.SpMSpV_.load_and_format_matrix(csc)
        self.n_ = self.SpMV_.get_num_rows()
        assert self.n_ == self.SpMV_.get_num_cols()

    def _initial_distance(self, source):
        return self._new_dense(self.n_, self.semiring_.zero, source, 0.0)

    def _pull_loop(self, vector, first_it, num_iterations):
        B, n = self.backend, self.n_
        results = B.alloc(n, np.float32)
        # The reference copies results -> vector after every SpMV (eWiseAdd with 0, app/sssp.h:163); here the two
        # buffers swap roles instead: same values, one launch and 8n bytes less per iteration.
        chained = (not self.comm.distributed) and hasattr(self.SpMV_, "chain") and self.SpMV_.chain(True)   # (as in PageRank.pull)
        try:
            for _ in range(first_it, num_iterations + 1):
                self.SpMV_.bind_vector_buf(vector)
                self.SpMV_.bind_results_buf(results)
                self.SpMV_.run()
                self._gather(results)
                vector, results = results, vector
        finally:
            if chained:
                self.SpMV_.chain(False)
        B.sync()
        return B.download_result(vector, n)

    def pull(self, source, num_iterations):
        return self._pull_loop(self._initial_distance(source), 1, num_iterations)

    def _start_push(self, source):
        B, n = self.backend, self.n_
        frontier = B.alloc(n + 1, capi.IDX_VAL)
        B.upload(B.view(frontier, 0, 2, 8), M.make_sparse_vec([source], [0.0]))
        distance = self._initial_distance(source)
        candidates = B.alloc(n + 1, capi.IDX_VAL)    # SpMSpV result of this shard
        self.SpMSpV_.bind_vector_buf(frontier)
        self.SpMSpV_.bind_mask_buf(distance)
        self.SpMSpV_.results_buf = candidates
        self.SparseAssign_.bind_mask_buf(candidates)
        self.SparseAssign_.bind_inout_buf(distance)
        if self.comm.distributed:
            local = B.alloc(n + 1, capi.IDX_VAL)     # relaxed entries of this shard
            self.SparseAssign_.bind_new_frontier_buf(local)
        else:
            local = None
            self.SparseAssign_.bind_new_frontier_buf(frontier)   # app/sssp.h:185-187
        return frontier, distance, candidates, local

    def _push_iteration(self, frontier, local):
        self.SpMSpV_.run()
        self.SparseAssign_.run()
        if self.comm.distributed:
            self._gather_sparse(local, frontier, self.n_, 0.0)

    def push(self, source, num_iterations):
        frontier, distance, _, local = self._start_push(source)
        for _ in range(num_iterations):
            self._push_iteration(frontier, local)
        self._gather(distance)
        self.backend.sync()
        return self.backend.download_result(distance, self.n_)

    def send_matrix_host_to_device(self):
        self.SpMV_.send_matrix_host_to_device()
        self.SpMSpV_.send_matrix_host_to_device()
        if hasattr(self.SpMSpV_, "attach_pull"):
            self.SpMSpV_.attach_pull(self.SpMV_)     # heavy frontiers of a push iteration go row-wise

    def pull_push(self, source, num_iterations, threshold=0.05):
        """app/sssp.h:197-243, host-driven like the reference -- but the count that decides the loop is the SpMSpV's own
        completion record (gl_spmspv_wait: stored to page-locked memory by the operator's last workgroup), read while the
        relax step that was enqueued behind it runs; no copy, no stream synchronisation per iteration.  (Rounds 2-3 also had
        a device-resident schedule with both steps of every slot enqueued and one gated off: five no-op launches per slot --
        same-box it lost on five of six stand-ins, 3.60 against 3.18 ms on ogbn-products' 23 slots,
        profiles/r04_ab_schedules.txt -- retired in round 4.)"""
        n = self.n_
        frontier, distance, candidates, local = self._start_push(source)
        it = 1
        while True:
            self._push_iteration(frontier, local)
            # app/sssp.h:221 (SpMSpV result size): the run's own completion record (the relax step is already enqueued behind
            # it), or the head element's copy where there is none
            nnz = self.SpMSpV_.get_results_nnz() if not self.comm.distributed and hasattr(self.SpMSpV_, "plan_") else self.comm_sparse_count(candidates)
            if self.comm.distributed:
                import torch
                gloo = self.comm.dist.get_backend(self.comm.group) == "gloo"
                t = torch.tensor([nnz], dtype=torch.int64, device="cpu" if gloo else distance.tensor.device)
                self.comm.dist.all_reduce(t, group=self.comm.group)
                nnz = int(t.item())
            it += 1
            if not (it < num_iterations and float(nnz) / n < threshold):
                break
        self.push_iterations_ = it - 1
        self._gather(distance)
        return self._pull_loop(distance, it, num_iterations)   # app/sssp.h:227-242
