"""Row-range partitioning and the one exchange step of the multi-GPU path.

The reference is single-device; SURVEY.md 8(e) defines the multi-GPU decomposition this build
adds: the matrix is cut into contiguous, nnz-balanced ROW RANGES (the axis the FPGA already tiles
on: out_buf_len row partitions, module/spmv_module.h:287), one range per GPU / process.  Each
iteration a rank produces y[r0:r1) and applies the element-wise ops on that slice; ONE all-gather
(RCCL over xGMI on the GPU box; gloo in the CPU tests) then rebuilds the full dense vector that is
the next iteration's x.  No reduction collectives are needed.

Everything here works on torch tensors of either device, so the control flow is testable with
world_size-2 gloo runs.
"""
import numpy as np


def partition_rows_by_nnz(indptr, world_size, align=64, equal_rows_tolerance=0.03):
    """Boundaries b[0..world_size] with b[0]=0, b[-1]=num_rows, balancing non-zeros per range and
    rounding interior cuts to a multiple of `align` rows (64 = one wavefront of dense-vector
    elements, keeps slices 256-byte aligned for the all-gather).

    When plain equal-length ranges are already balanced to within `equal_rows_tolerance` (randomly
    labelled graphs are), they are preferred: equal slices let the all-gather run in place, without
    the pad / unpack copies uneven slices need."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n = indptr.shape[0] - 1
    nnz = int(indptr[n])
    if world_size > 1 and n % (world_size * align) == 0 and nnz > 0:
        eq = [n // world_size * r for r in range(world_size + 1)]
        per = np.diff(indptr[eq])
        if per.max() <= (1.0 + equal_rows_tolerance) * nnz / world_size:
            return eq
    bounds = [0]
    for k in range(1, world_size):
        target = nnz * k // world_size
        r = int(np.searchsorted(indptr, target, side="left"))
        r = min(n, max(bounds[-1], (r + align // 2) // align * align))
        bounds.append(r)
    bounds.append(n)
    return bounds


class Comm:
    """Thin wrapper around a torch.distributed process group (None = single process)."""

    def __init__(self, group=None, force=False):
        # force: take the collective paths even in a one-rank group (lets ONE GPU exercise RCCL itself: communicator
        # set-up, the in-place all_gather_into_tensor, the sparse-list gather -- tests/test_gpu_dist.py)
        self.force = bool(force) and group is not None
        self.group = group
        if group is None:
            self.rank, self.world_size = 0, 1
        else:
            import torch.distributed as dist
            self.dist = dist
            self.rank = dist.get_rank(group) if group is not True else dist.get_rank()
            self.world_size = dist.get_world_size(group) if group is not True else dist.get_world_size()
            if group is True:
                self.group = None  # default group

    @property
    def distributed(self):
        return self.world_size > 1 or self.force

    def all_gather_slices(self, full, bounds):
        """`full` is a 1-D tensor over the whole vertex range whose slice
        [bounds[rank], bounds[rank+1]) is valid on this rank; afterwards all of it is valid on every
        rank.  Ranges are nnz-balanced, hence of different lengths: each rank pads its slice to the
        longest one, ONE all_gather_into_tensor moves everything (the same call on RCCL and gloo),
        and one concatenation writes the received slices back in place."""
        if not self.distributed:
            return
        import torch
        W = self.world_size
        lens = [bounds[r + 1] - bounds[r] for r in range(W)]
        mx = max(lens)
        if mx == 0:
            return
        gloo_on_gpu = full.is_cuda and self.dist.get_backend(self.group) == "gloo"
        if min(lens) == mx and bounds[0] == 0 and full.is_contiguous() and not gloo_on_gpu:
            # equal slices: gather straight into the vector (the clone keeps input and output disjoint)
            mine = full[bounds[self.rank]:bounds[self.rank + 1]].clone()
            self.dist.all_gather_into_tensor(full[:bounds[W]], mine, group=self.group)
            return
        key = (full.device, full.dtype, W * mx)
        if getattr(self, "_stage_key", None) != key:
            self._stage = torch.empty(W * mx, dtype=full.dtype, device=full.device)
            self._send = torch.zeros(mx, dtype=full.dtype, device=full.device)
            self._stage_key = key
        self._send[:lens[self.rank]].copy_(full[bounds[self.rank]:bounds[self.rank + 1]])
        if full.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # debugging aid (two ranks on one GPU, no RCCL): stage the collective through the host
            stage_h, send_h = self._stage.cpu(), self._send.cpu()
            self.dist.all_gather_into_tensor(stage_h, send_h, group=self.group)
            self._stage.copy_(stage_h)
        else:
            self.dist.all_gather_into_tensor(self._stage, self._send, group=self.group)
        torch.cat([self._stage[r * mx:r * mx + lens[r]] for r in range(W)], out=full[bounds[0]:bounds[W]])

    def all_gather_slices_with_tail(self, full, bounds, tail, per):
        """all_gather_slices of `full` and, in the SAME collective, an all-gather of equal blocks: `tail` holds world_size
        blocks of `per` elements, rank r's at r * per (the ranks' tallies of a BFS slot travel with the slot's bit vector:
        one collective per slot, not two).  Always staged: every rank sends [its slice, padded to the longest | its block]."""
        if not self.distributed:
            return
        import torch
        W = self.world_size
        lens = [bounds[r + 1] - bounds[r] for r in range(W)]
        mx = max(lens) + per
        key = ("tail", full.device, full.dtype, W * mx)
        if getattr(self, "_tstage_key", None) != key:
            self._tstage = torch.empty(W * mx, dtype=full.dtype, device=full.device)
            self._tsend = torch.zeros(mx, dtype=full.dtype, device=full.device)
            self._tstage_key = key
        self._tsend[:lens[self.rank]].copy_(full[bounds[self.rank]:bounds[self.rank + 1]])
        self._tsend[mx - per:].copy_(tail[self.rank * per:(self.rank + 1) * per])
        if full.is_cuda and self.dist.get_backend(self.group) == "gloo":
            stage_h, send_h = self._tstage.cpu(), self._tsend.cpu()      # debugging aid, see all_gather_slices
            self.dist.all_gather_into_tensor(stage_h, send_h, group=self.group)
            self._tstage.copy_(stage_h)
        else:
            self.dist.all_gather_into_tensor(self._tstage, self._tsend, group=self.group)
        torch.cat([self._tstage[r * mx:r * mx + lens[r]] for r in range(W)], out=full[bounds[0]:bounds[W]])
        torch.cat([self._tstage[r * mx + mx - per:(r + 1) * mx] for r in range(W)], out=tail[:W * per])

    def all_gather_sparse(self, local, count, capacity_full, out):
        """Concatenate per-rank sparse lists ((index,val) pairs as an int64-viewable [k,2] float/int
        tensor) in rank order.  `local` holds this rank's entries (without a head), `count` their
        number.  Returns the total count; `out` receives the concatenation."""
        import torch
        if not self.distributed:
            out[:count] = local[:count]
            return int(count)
        cnt = torch.tensor([count], dtype=torch.int64,
                           device="cpu" if self.dist.get_backend(self.group) == "gloo" else local.device)
        counts = [torch.zeros_like(cnt) for _ in range(self.world_size)]
        self.dist.all_gather(counts, cnt, group=self.group)
        counts = [int(c.item()) for c in counts]
        total = sum(counts)
        assert total <= capacity_full
        offs = np.concatenate([[0], np.cumsum(counts)])
        pieces = [out[int(offs[r]):int(offs[r + 1])] for r in range(self.world_size)]
        mx = max(counts)
        if mx == 0:
            return 0
        # pad to a common length so one all_gather_into_tensor moves everything
        host = local.is_cuda and self.dist.get_backend(self.group) == "gloo"   # debugging aid, see above
        cdev = "cpu" if host else local.device
        send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
        send[:count] = local[:count]
        recv = torch.empty((self.world_size * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        for r in range(self.world_size):
            pieces[r].copy_(recv[r * mx:r * mx + counts[r]])
        return total

    def barrier(self):
        if self.distributed:
            self.dist.barrier(group=self.group)


class EmulatedComm:
    """ONE rank of a `world_size`-rank row-sharded run on ONE GPU, with the exchange step stubbed: no 8-GPU node is
    available to this build, so the per-rank work of the sharded schedules is measured this way (bench.py
    --emulate-rank k/N).  The other ranks' rows of every exchanged bit vector are copied from the vectors of a
    whole-matrix run of the same BFS on the same GPU (`set_truth`: slot s of the bit-frontier schedule writes vector
    s + 1 = the vertices at distance s + 1, whatever the sharding), so the rank's kernels see exactly the frontiers, and
    take exactly the decisions, of a real N-rank run.  Two forms:
      copy=False (default)  no exchange step at all: slot s READS the whole run's vector s and writes the rank's own rows
                            of vector s + 1 into the rank's own buffer (compared with the truth by the tests);
      copy=True             the exchange step is there, as (at most two) device-to-device copies of the other ranks' rows
                            into the rank's vector -- the place RCCL's all-gather takes in a real run.
    What is NOT measured either way: the collective itself (tabulated separately)."""
    emulated = True
    capturable = True          # plain device-to-device copies: the schedule can be recorded as a hipGraph
    group = None

    def __init__(self, rank, world_size, copy=False):
        self.rank, self.world_size, self.copy = int(rank), int(world_size), bool(copy)
        self.truth = None

    def truth_vector(self, k):
        """vector k of the whole-matrix run (copy=False: the frontier the rank's slot k reads)"""
        from . import capi
        nb = 4 * self.truth_words
        return capi.DeviceBuffer(nb, ptr=self.truth.ptr + k * nb, keepalive=self.truth)

    @property
    def distributed(self):
        return True

    def set_truth(self, vecs_buf, words):
        """vecs_buf: the (N + 2) x words bit vectors a whole-matrix BFS._pull_push_bits left behind (device)."""
        self.truth, self.truth_words = vecs_buf, int(words)

    def exchange_bits(self, bits_buf, k, bounds, tally_slot_buf=None, tally_slot=None):
        from . import capi
        assert self.truth is not None, "EmulatedComm.set_truth first"
        if not self.copy:
            return
        lo, hi = bounds[self.rank] // 8, bounds[self.rank + 1] // 8          # bytes (bounds are multiples of 64 rows)
        total = bits_buf.nbytes
        base = 4 * k * self.truth_words
        for a, b in ((0, lo), (hi, total)):
            if b > a:
                dst = capi.DeviceBuffer(b - a, ptr=bits_buf.ptr + a, keepalive=bits_buf)
                src = capi.DeviceBuffer(b - a, ptr=self.truth.ptr + base + a, keepalive=self.truth)
                capi.copy_d2d(dst, src, b - a)
        if tally_slot_buf is not None:      # the other ranks' tallies of the slot (their 256-byte blocks of the recorded table)
            per = 4 * capi.GL_BFS_TALLY_RANK_WORDS
            tbase = 4 * capi.GL_BFS_TALLY_HEAD_WORDS + (tally_slot - 1) * self.world_size * per
            for a, b in ((0, self.rank * per), ((self.rank + 1) * per, self.world_size * per)):
                if b > a:
                    dst = capi.DeviceBuffer(b - a, ptr=tally_slot_buf.ptr + a, keepalive=tally_slot_buf)
                    src = capi.DeviceBuffer(b - a, ptr=self._tally.ptr + tbase + a, keepalive=self._tally)
                    capi.copy_d2d(dst, src, b - a)

    def all_gather_slices(self, tensor_or_buf, bounds):
        """The dense exchange of the pull loops (PageRank, SSSP pull: every rank's rows of the result vector) -- stubbed out:
        the rank's SpMV reads whatever the other rows hold, which changes its values and not its time (the dense kernels
        stream the same entries whatever x is).  The time of the all-gather itself is tabulated, not measured."""
        return

    def truth_tally(self, key, slots, bounds, col_len, row_len, n):
        """Every rank's tallies of every slot of the whole-matrix run (gl_bfs_bits_shard_step's table: per slot and rank
        {vertices reached in the rank's rows, their global column lengths, their row lengths}), computed on the host from
        the recorded bit vectors: what the other ranks of a real run would have sent.  Cached per (source, slots)."""
        from . import capi
        if getattr(self, "_tally_key", None) == (key, self.truth.ptr):
            return self._tally
        W = self.world_size
        table = np.zeros(capi.bfs_tally_words(slots, W), np.uint32)
        vec = self.truth.read(np.uint32, (slots + 2) * self.truth_words).reshape(-1, self.truth_words)
        cl, rl = np.asarray(col_len, np.uint64), np.asarray(row_len, np.uint64)
        for s in range(1, slots + 1):
            bits = np.unpackbits(vec[s + 1].view(np.uint8), bitorder="little")[:n].astype(bool)
            for r in range(W):
                lo, hi = bounds[r], bounds[r + 1]
                sl = bits[lo:hi]
                base = capi.GL_BFS_TALLY_HEAD_WORDS + ((s - 1) * W + r) * capi.GL_BFS_TALLY_RANK_WORDS
                table[base] = int(sl.sum())
                table[base + 2:base + 4] = np.array([int(cl[lo:hi][sl].sum())], np.uint64).view(np.uint32)
                table[base + 4:base + 6] = np.array([int(rl[lo:hi][sl].sum())], np.uint64).view(np.uint32)
        if getattr(self, "_tally", None) is None or self._tally.nbytes != table.nbytes:
            self._tally = capi.DeviceBuffer(table.nbytes)
        self._tally.write(table)
        self._tally_key = (key, self.truth.ptr)
        self.truth_tally_host = table
        return self._tally

    def all_gather_slices(self, full, bounds):
        return                 # nothing to gather from: only this rank's slice of `full` is valid

    def all_gather_sparse(self, *a, **k):
        raise NotImplementedError("the emulation covers the bit-frontier schedules only")

    def barrier(self):
        return


class CabiComm(Comm):
    """Comm whose bit-vector exchange goes through the C ABI (gl_dist_all_gather_bits_tally: grouped ncclSend / ncclRecv on the
    library's stream) instead of torch.distributed -- what a C++ caller of the drop-in headers uses.  The communicator's
    unique id travels through the torch process group."""

    # the exchange is a grouped ncclSend / ncclRecv on the library's own stream: it is recorded INTO the BFS schedule's hipGraph
    # (one launch + one exchange per slot replayed with a single call), which torch.distributed's collectives are not
    capturable = True

    def __init__(self, group=True, force=False):
        super().__init__(group, force=force)
        from . import capi
        import torch.distributed as dist
        uid = [capi.Dist.unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(uid, src=0, group=self.group)
        self.gl = capi.Dist(self.rank, self.world_size, uid[0])

    def exchange_bits(self, bits_buf, k, bounds, tally_slot_buf=None, tally_slot=None):
        self.gl.all_gather_bits_tally(bits_buf, bounds, tally_slot_buf)


def preflight(comm, device, expect_world, cabi=None, rows=3_072_512, watchdog_s=60.0, fail_cabi=False):
    """First contact with the peers, BEFORE anything is timed (bench.py --gpus N; round 4's verdict: the row-sharded path had
    only ever met one RCCL rank, gloo ranks and an emulation):

      * the process group holds `expect_world` ranks;
      * a BIT all-gather of `rows` bits (orkut: 384 KB -- a BFS slot's exchange) and a DENSE all-gather of `rows` floats
        (12 MB -- a PageRank / SSSP iteration's exchange), each on nnz-style UNEVEN bounds and on equal bounds, every word
        verified on every rank against the pattern its owner wrote;
      * the same two through the C ABI's communicator (`cabi`: a CabiComm, gl_dist_all_gather_bits_tally with tallies /
        gl_dist_all_gather_f32) when one is given -- a failure there (exception or wrong words) is RECORDED and the caller
        falls back to the torch path;
      * how long one exchange of each kind takes (median of 10, device-synchronised) -- `exchange_ms`, timed apart from compute;
      * a watchdog: if all of this takes longer than `watchdog_s` (a collective that never completes), the process prints one
        JSON line saying so and exits with code 3 instead of hanging its launcher.

    Returns {"ranks", "backend", "verified", "exchange_path", "cabi_error", "exchange_ms": {...}}; raises on a torch-path failure
    (there is nothing to fall back to)."""
    import json
    import os
    import threading
    import time
    import torch
    W, rank = comm.world_size, comm.rank
    out = {"ranks": W, "backend": None, "verified": False, "exchange_path": "torch", "cabi_error": None, "exchange_ms": {}}
    if not comm.distributed:
        out.update({"backend": "none", "verified": True, "exchange_path": "none"})
        return out
    done = threading.Event()

    def watchdog():
        if not done.wait(watchdog_s):
            print(json.dumps({"error": "multi-GPU pre-flight did not finish within %.0f s (a collective hung); rank %d of %d"
                                       % (watchdog_s, rank, W), "preflight": out}), flush=True)
            os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        out["backend"] = comm.dist.get_backend(comm.group)
        if W != expect_world:
            raise RuntimeError("the process group holds %d ranks, --gpus asked for %d" % (W, expect_world))
        rows = max(rows // (64 * W) * (64 * W), 64 * W)
        words = rows // 32
        cuda = torch.device(device).type == "cuda"

        def sync():
            if cuda:
                torch.cuda.synchronize()

        def bounds_of(uneven):
            if not uneven or W == 1:
                return [rows // W * r for r in range(W + 1)]
            # nnz-balanced ranges differ in length: shift the interior cuts by alternating multiples of 64 rows
            b = [0] + [rows // W * r + (64 * (1 + r % 3) if r % 2 else -64 * (1 + r % 2)) for r in range(1, W)] + [rows]
            return b

        def owner_pattern(idx, bnds):
            own = torch.zeros_like(idx)
            for r in range(1, W):
                own += (idx >= bnds[r]).to(idx.dtype)
            return own

        def check(kind, vec, bnds, unit):
            idx = torch.arange(vec.shape[0], device=vec.device, dtype=torch.int64)
            own = owner_pattern(idx, [b // unit for b in bnds])
            want = (idx * 7 + own * 1000003 + 11) % 16777213       # (exact in float32 below 2^24)
            ok = bool(torch.equal(vec.to(torch.int64), want))
            flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=vec.device if out["backend"] != "gloo" else "cpu")
            comm.dist.all_reduce(flag, group=comm.group)
            if int(flag.item()):
                raise RuntimeError("%s all-gather returned wrong words on %d rank(s)" % (kind, int(flag.item())))

        def fill(vec, bnds, unit):
            vec.zero_()
            lo, hi = bnds[rank] // unit, bnds[rank + 1] // unit
            idx = torch.arange(lo, hi, device=vec.device, dtype=torch.int64)
            vec[lo:hi] = ((idx * 7 + rank * 1000003 + 11) % 16777213).to(vec.dtype)

        def median_ms(fn, n=10):
            ts = []
            for _ in range(n):
                sync()
                t0 = time.perf_counter()
                fn()
                sync()
                ts.append(time.perf_counter() - t0)
            return round(float(np.median(ts)) * 1e3, 4)

        bits = torch.zeros(words, dtype=torch.int32, device=device)
        dense = torch.zeros(rows, dtype=torch.float32, device=device)
        for uneven in (True, False):
            bnds = bounds_of(uneven)
            fill(bits, bnds, 32)
            comm.all_gather_slices(bits, [b // 32 for b in bnds])
            check("bit (torch path, %s bounds)" % ("uneven" if uneven else "equal"), bits, bnds, 32)
            fill(dense, bnds, 1)
            comm.all_gather_slices(dense, bnds)
            check("dense (torch path, %s bounds)" % ("uneven" if uneven else "equal"), dense, bnds, 1)
        bnds = bounds_of(True)
        out["exchange_ms"]["torch"] = {"bits_384KB": median_ms(lambda: comm.all_gather_slices(bits, [b // 32 for b in bnds])),
                                       "dense_12MB": median_ms(lambda: comm.all_gather_slices(dense, bnds))}
        out["verified"] = True
        if cabi is not None:
            # Every rank walks the SAME sequence of collectives whatever it finds locally (a rank that left the
            # sequence on a rank-local verdict -- wrong words, wrong tallies -- met the others' next collective with its final
            # flag all-reduce, and the job ended in the watchdog).  A step = one local action (may be a C ABI collective) + one
            # local verdict; the verdicts are all-reduced after EACH step, so that all ranks skip the remaining steps together.
            # (A rank whose C ABI call throws BEFORE it has joined that call's collective still strands its peers inside it:
            # that case is the watchdog's.)
            cabi_err = None

            def words_ok(vec, bnds, unit):      # local compare only -- no collective inside a step's verdict
                idx = torch.arange(vec.shape[0], device=vec.device, dtype=torch.int64)
                own = owner_pattern(idx, [b // unit for b in bnds])
                return bool(torch.equal(vec.to(torch.int64), (idx * 7 + own * 1000003 + 11) % 16777213))

            def step(name, fn):
                nonlocal cabi_err
                ok, why = True, ""
                if cabi_err is None:            # (the same on every rank: set from all-reduced verdicts only)
                    try:
                        ok = bool(fn())
                    except Exception as e:
                        ok, why = False, ": " + repr(e)
                    flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=device if out["backend"] != "gloo" else "cpu")
                    comm.dist.all_reduce(flag, group=comm.group)
                    if int(flag.item()):
                        cabi_err = "%s failed on %d rank(s)%s" % (name, int(flag.item()), why if not ok else "")

            state = {}

            def s_setup():
                if fail_cabi:
                    raise RuntimeError("forced by the test hook (GRAPHLILY_DEBUG dist_preflight_fail_cabi=1)")
                from . import capi as capi_lib
                capi = getattr(cabi, "capi_module", None) or capi_lib     # (tests drive this section on gloo ranks with a stand-in)
                state["capi"] = capi
                state["bbits"], state["bdense"] = capi.DeviceBuffer.from_torch(bits), capi.DeviceBuffer.from_torch(dense)
                state["tally"] = torch.zeros(W * capi.GL_BFS_TALLY_RANK_WORDS, dtype=torch.int32, device=device)
                state["btally"] = capi.DeviceBuffer.from_torch(state["tally"])
                return True

            step("C ABI pre-flight setup", s_setup)
            for uneven in (True, False):
                bnds = bounds_of(uneven)
                tag = "uneven" if uneven else "equal"

                def s_bits(bnds=bnds):
                    capi, tally = state["capi"], state["tally"]
                    fill(bits, bnds, 32)
                    tally.zero_()
                    tally[rank * capi.GL_BFS_TALLY_RANK_WORDS:(rank + 1) * capi.GL_BFS_TALLY_RANK_WORDS] = rank + 1
                    sync()
                    cabi.gl.all_gather_bits_tally(state["bbits"], bnds, state["btally"])
                    capi.sync()
                    want = torch.arange(1, W + 1, device=device, dtype=torch.int32).repeat_interleave(capi.GL_BFS_TALLY_RANK_WORDS)
                    return words_ok(bits, bnds, 32) and bool(torch.equal(tally, want))

                def s_dense(bnds=bnds):
                    fill(dense, bnds, 1)
                    sync()
                    cabi.gl.all_gather_f32(state["bdense"], bnds)
                    state["capi"].sync()
                    return words_ok(dense, bnds, 1)

                step("bit all-gather with tallies (C ABI path, %s bounds)" % tag, s_bits)
                step("dense all-gather (C ABI path, %s bounds)" % tag, s_dense)
            bnds = bounds_of(True)

            def s_time():
                def cabi_bits():
                    cabi.gl.all_gather_bits_tally(state["bbits"], bnds, state["btally"])
                    state["capi"].sync()

                def cabi_dense():
                    cabi.gl.all_gather_f32(state["bdense"], bnds)
                    state["capi"].sync()

                out["exchange_ms"]["cabi"] = {"bits_384KB_with_tallies": median_ms(cabi_bits), "dense_12MB": median_ms(cabi_dense)}
                return True

            step("C ABI exchange timing", s_time)
            if cabi_err is None:
                out["exchange_path"] = "cabi"
            else:                               # recorded; the caller keeps the torch path -- on every rank alike
                out["cabi_error"] = cabi_err
        return out
    finally:
        done.set()
