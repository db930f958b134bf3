"""CPU suite, part 3: the N>1 path over gloo, world_size 2 -- row partitioning, the padded slice
all-gather, the sparse-frontier all-gather, and the drivers' distributed control flow (BFS pull /
pull_push, PageRank, SSSP pull / push / pull_push) with a CPU stand-in for the device modules."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphlily_amd import datasets, io
from graphlily_amd.dist import Comm, partition_rows_by_nnz
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, out_q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = globals()[fn_name](Comm(True))
        out_q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    # drain the queue BEFORE joining: a worker blocks in put() until its payload has been read
    import time
    deadline = time.time() + 240
    while len(out) < world and time.time() < deadline:
        if not q.empty():
            r, res = q.get()
            out[r] = res
        elif any(p.exitcode not in (None, 0) for p in procs):
            break
        else:
            time.sleep(0.05)
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0, "worker failed (exit code %s)" % p.exitcode
    assert len(out) == world
    return out


def test_partition_balances_nnz():
    m = datasets.rmat(20000, 400000, seed=11)
    for w in (1, 2, 4, 8):
        b = partition_rows_by_nnz(m.adj_indptr, w)
        assert b[0] == 0 and b[-1] == m.num_rows and len(b) == w + 1
        assert all(b[i] <= b[i + 1] for i in range(w))
        assert all(x % 64 == 0 for x in b[1:-1])
        per = np.diff(m.adj_indptr.astype(np.int64)[b])
        assert per.sum() == m.nnz
        if w > 1:
            # balanced up to the granularity imposed by the alignment and the heaviest row
            slack = np.diff(m.adj_indptr.astype(np.int64)).max() * 64
            assert per.max() <= m.nnz / w + slack


# ------------------------------------------------------------------ worker bodies (module level)
def _body_gather_slices(comm):
    n = 1000
    bounds = [0, 384, n]          # uneven on purpose
    full = torch.full((n,), -1.0)
    full[bounds[comm.rank]:bounds[comm.rank + 1]] = torch.arange(bounds[comm.rank], bounds[comm.rank + 1],
                                                                 dtype=torch.float32)
    comm.all_gather_slices(full, bounds)
    return bool(torch.equal(full, torch.arange(n, dtype=torch.float32)))


def _body_gather_sparse(comm):
    cap = 64
    local = torch.arange(10 * comm.rank, 10 * comm.rank + (3 if comm.rank == 0 else 5), dtype=torch.int64)
    out = torch.zeros(cap, dtype=torch.int64)
    total = comm.all_gather_sparse(local, local.numel(), cap, out)
    return total, out[:total].tolist()


def _graph():
    m = datasets.rmat(3000, 30000, seed=21, symmetric=True)
    return m


def _skewed_graph():
    """Power-law rows in their generator order (no relabelling): the nnz-balanced row ranges of 4 and 8 ranks are far from
    equal length, so every all-gather takes the pad / unpack branch (dist.Comm.all_gather_slices)."""
    n, m = 4096, 60000
    rng = np.random.default_rng(5)
    w = 1.0 / np.arange(1, n + 1) ** 0.9
    rows = rng.choice(n, size=m, p=w / w.sum())
    cols = rng.integers(0, n, size=m)
    rows, cols = np.concatenate([rows, cols]), np.concatenate([cols, rows])
    key = np.unique(rows.astype(np.int64) * n + cols)
    r, c = (key // n).astype(np.int64), (key % n).astype(np.uint32)
    indptr = np.zeros(n + 1, np.int64)
    np.add.at(indptr, r + 1, 1)
    return io.CSRMatrix(n, n, np.ones(key.shape[0], np.float32), c, np.cumsum(indptr).astype(np.uint32))


def _apps(comm, which, graph=None):
    from cpu_backend import CpuBackend
    from graphlily_amd import app, module as M
    m = _graph() if graph is None else graph
    if which == "bfs":
        a = app.BFS(16, 1024, 512, 256, comm=comm, backend=CpuBackend())
        a.set_up_runtime()
        a.load_and_format_matrix(m, True)
        a.send_matrix_host_to_device()
        return [a.pull(0, 6), a.pull_push(0, 6, 0.05), a.push(0, 6)]
    if which == "pagerank":
        a = app.PageRank(16, 1024, 256, comm=comm, backend=CpuBackend())
        a.set_up_runtime()
        a.load_and_format_matrix(m, 0.9, True)
        a.send_matrix_host_to_device()
        return [a.pull(0.9, 5)]
    a = app.SSSP(16, 1024, 512, 256, comm=comm, backend=CpuBackend(), semiring=M.TropicalSemiringUfixed)
    a.set_up_runtime()
    a.load_and_format_matrix(m, True)
    a.send_matrix_host_to_device()
    return [a.pull(0, 6), a.pull_push(0, 6, 0.05), a.push(0, 6)]


def _body_bfs(comm):
    return _apps(comm, "bfs")


def _body_pagerank(comm):
    return _apps(comm, "pagerank")


def _body_sssp(comm):
    return _apps(comm, "sssp")


def _body_bfs_skewed(comm):
    return _apps(comm, "bfs", _skewed_graph())


def _body_pagerank_skewed(comm):
    return _apps(comm, "pagerank", _skewed_graph())


def _body_sssp_skewed(comm):
    return _apps(comm, "sssp", _skewed_graph())


def _body_gather_uneven_many(comm):
    """float slices, bit-vector words and sparse lists over uneven ranges, one of them EMPTY"""
    W, n = comm.world_size, 64 * 41
    cuts = sorted(set([0, n] + [64 * k for k in (1, 2, 9, 9, 20, 33, 40)]))[: W + 1]
    bounds = (cuts + [n] * (W + 1))[: W + 1]
    bounds[-1] = n
    bounds[2] = bounds[1]                      # rank 1 owns nothing
    lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
    full = torch.full((n,), -1.0)
    full[lo:hi] = torch.arange(lo, hi, dtype=torch.float32)
    comm.all_gather_slices(full, bounds)
    ok = bool(torch.equal(full, torch.arange(n, dtype=torch.float32)))
    words = torch.zeros(n // 32, dtype=torch.int32)
    words[lo // 32:hi // 32] = torch.arange(lo // 32, hi // 32, dtype=torch.int32) + 1
    comm.all_gather_slices(words, [b // 32 for b in bounds])
    ok = ok and bool(torch.equal(words, torch.arange(n // 32, dtype=torch.int32) + 1))
    cnt = (comm.rank * 7) % 5                  # some ranks contribute nothing
    local = torch.arange(100 * comm.rank, 100 * comm.rank + cnt, dtype=torch.int64)
    out = torch.zeros(64, dtype=torch.int64)
    total = comm.all_gather_sparse(local, cnt, 64, out)
    want = [100 * r + k for r in range(W) for k in range((r * 7) % 5)]
    return ok and total == len(want) and out[:total].tolist() == want


# ------------------------------------------------------------------ tests
def test_all_gather_slices_uneven():
    assert all(_spawn("_body_gather_slices").values())


def _body_gather_slices_equal(comm):
    n = 1024
    bounds = [0, 512, n]
    full = torch.full((n,), -1.0)
    full[bounds[comm.rank]:bounds[comm.rank + 1]] = torch.arange(bounds[comm.rank], bounds[comm.rank + 1],
                                                                 dtype=torch.float32)
    comm.all_gather_slices(full, bounds)
    return bool(torch.equal(full, torch.arange(n, dtype=torch.float32)))


def test_all_gather_slices_equal_in_place():
    assert all(_spawn("_body_gather_slices_equal").values())


def test_partition_prefers_equal_rows_when_balanced():
    m = datasets.uniform(4096, 5, seed=1)           # perfectly uniform rows
    assert partition_rows_by_nnz(m.adj_indptr, 4) == [0, 1024, 2048, 3072, 4096]
    skew = datasets.rmat(4096, 60000, seed=3)       # a few hub rows: equal rows are NOT balanced
    b = partition_rows_by_nnz(skew.adj_indptr, 4, equal_rows_tolerance=0.0)
    assert b[0] == 0 and b[-1] == 4096


def test_all_gather_sparse():
    out = _spawn("_body_gather_sparse")
    for r in (0, 1):
        assert out[r] == (8, [0, 1, 2, 10, 11, 12, 13, 14])


def _oracle_graph(sssp=False, pagerank=False):
    m = _graph()
    om = O.CSR(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)
    if sssp:
        O.sssp_preprocess(om)
    O.util_round_csr_matrix_dim(om, 128, 128)
    if pagerank:
        O.util_normalize_csr_matrix_by_outdegree(om)
        om.adj_data = (om.adj_data * np.float32(0.9)).astype(np.float32)
    elif not sssp:
        om.adj_data[:] = 1
    return om


def test_distributed_bfs_matches_oracle():
    ref = O.bfs(_oracle_graph(), 0, 6)
    out = _spawn("_body_bfs")
    for r in (0, 1):
        for got in out[r]:
            assert np.array_equal(got, ref)


def test_distributed_pagerank_matches_oracle():
    ref = O.pagerank(_oracle_graph(pagerank=True), 0.9, 5)
    out = _spawn("_body_pagerank")
    for r in (0, 1):
        assert np.array_equal(out[r][0], ref)     # same per-row order on the stand-in => bit-equal


def test_distributed_sssp_matches_oracle():
    ref = O.sssp(_oracle_graph(sssp=True), 0, 6, 255.0)
    out = _spawn("_body_sssp")
    for r in (0, 1):
        for got in out[r]:
            assert np.array_equal(got, ref)


# ------------------------------------------------------------------ worlds of 4 and 8, uneven ranges
def _oracle_skewed(sssp=False, pagerank=False):
    m = _skewed_graph()
    om = O.CSR(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)
    if sssp:
        O.sssp_preprocess(om)
    O.util_round_csr_matrix_dim(om, 128, 128)
    if pagerank:
        O.util_normalize_csr_matrix_by_outdegree(om)
        om.adj_data = (om.adj_data * np.float32(0.9)).astype(np.float32)
    elif not sssp:
        om.adj_data[:] = 1
    return om


def test_skewed_partition_is_uneven():
    m = _skewed_graph()
    io.util_round_csr_matrix_dim(m, 128, 128)
    for w in (4, 8):
        b = partition_rows_by_nnz(m.adj_indptr, w)
        lens = np.diff(b)
        assert lens.max() >= 3 * max(lens.min(), 1), "hub rows first: the balanced ranges differ several-fold in length"
        per = np.diff(m.adj_indptr.astype(np.int64)[b])
        assert per.max() <= 2.5 * m.nnz / w


def _body_gather_with_tail(comm):
    """The BFS slot exchange of the one-launch shard step: every rank's rows of the bit vector (uneven word ranges, one
    of them empty) AND its 64-word tally block in ONE collective; bit patterns travel in a float tensor (NaNs included)."""
    W, per = comm.world_size, 64
    cuts = [0, 40, 40, 1000, 1017, 1500, 1501, 2900, 3072][:W] + [3072]
    cuts = sorted(cuts)
    words = torch.zeros(3072, dtype=torch.int32)
    lo, hi = cuts[comm.rank], cuts[comm.rank + 1]
    words[lo:hi] = (torch.arange(lo, hi, dtype=torch.int32) * 2654435 + 0x7fc00001)       # (NaN patterns as floats)
    tail = torch.zeros(W * per + 8, dtype=torch.int32)
    tail[comm.rank * per:(comm.rank + 1) * per] = torch.arange(per, dtype=torch.int32) + 1000 * (comm.rank + 1)
    tail[W * per:] = -7                                                                    # (not part of the exchange)
    comm.all_gather_slices_with_tail(words.view(torch.float32), cuts, tail.view(torch.float32), per)
    ok = bool(torch.equal(words, torch.arange(3072, dtype=torch.int32) * 2654435 + 0x7fc00001))
    want = torch.cat([torch.arange(per, dtype=torch.int32) + 1000 * (r + 1) for r in range(W)] + [torch.full((8,), -7, dtype=torch.int32)])
    return ok and bool(torch.equal(tail, want))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bits_and_tallies_travel_in_one_collective(world):
    assert all(_spawn("_body_gather_with_tail", world).values())


@pytest.mark.parametrize("world", [4, 8])
def test_gathers_over_many_uneven_ranges(world):
    assert all(_spawn("_body_gather_uneven_many", world).values())


@pytest.mark.parametrize("world", [4, 8])
def test_distributed_bfs_many_ranks(world):
    ref = O.bfs(_oracle_skewed(), 0, 6)
    out = _spawn("_body_bfs_skewed", world)
    for r in range(world):
        for got in out[r]:
            assert np.array_equal(got, ref)
    assert (ref != 0).sum() > ref.shape[0] // 2


@pytest.mark.parametrize("world", [4, 8])
def test_distributed_pagerank_many_ranks(world):
    ref = O.pagerank(_oracle_skewed(pagerank=True), 0.9, 5)
    out = _spawn("_body_pagerank_skewed", world)
    for r in range(world):
        assert np.array_equal(out[r][0], ref)


def test_distributed_sssp_four_ranks():
    ref = O.sssp(_oracle_skewed(sssp=True), 0, 6, 255.0)
    out = _spawn("_body_sssp_skewed", 4)
    for r in range(4):
        for got in out[r]:
            assert np.array_equal(got, ref)


def test_gl_dist_slice_plan_arithmetic():
    """gl_dist_slice_plan: the bounds -> byte ranges arithmetic of the C ABI's three all-gathers (csrc/gl_dist.hip), which
    no one-GPU box can exercise with peers: float slices, whole words of a bit vector, sparse lists behind the head."""
    from graphlily_amd import capi
    lo, hi = capi.dist_slice_plan(0, [0, 100, 100, 4096])
    assert lo.tolist() == [0, 400, 400] and hi.tolist() == [400, 400, 16384]
    # bit vectors: multiples of 32 rows inside, the end of the vector anywhere (rounded up to a whole word)
    lo, hi = capi.dist_slice_plan(1, [0, 64, 64, 320, 1000])
    assert lo.tolist() == [0, 8, 8, 40] and hi.tolist() == [8, 8, 40, 128]
    assert all(h0 == l1 for h0, l1 in zip(hi[:-1], lo[1:])), "the ranges tile the vector: no word has two writers"
    with pytest.raises(capi.GraphLilyError):
        capi.dist_slice_plan(1, [0, 48, 128])              # a boundary inside a word
    with pytest.raises(capi.GraphLilyError):
        capi.dist_slice_plan(0, [0, 200, 100])             # decreasing
    # sparse lists: entries follow the 8-byte head, rank order
    lo, hi = capi.dist_slice_plan(2, [3, 0, 5, 1])
    assert lo.tolist() == [8, 32, 32, 72] and hi.tolist() == [32, 32, 72, 80]
    # the nnz-balanced bounds of a power-law matrix at 8 ranks
    m = _skewed_graph()
    io.util_round_csr_matrix_dim(m, 128, 128)
    b = partition_rows_by_nnz(m.adj_indptr, 8)
    lo, hi = capi.dist_slice_plan(1, b)
    assert lo[0] == 0 and hi[-1] == 4 * ((m.num_rows + 31) // 32) and np.all(hi[:-1] == lo[1:]) and np.all(hi >= lo)


# ---- the multi-GPU pre-flight of bench.py --gpus N (graphlily_amd.dist.preflight), on gloo ranks
def _body_preflight(comm):
    from graphlily_amd.dist import preflight
    pf = preflight(comm, "cpu", comm.world_size, rows=64 * comm.world_size * 32)
    ok = (pf["ranks"] == comm.world_size and pf["backend"] == "gloo" and pf["verified"] is True and pf["exchange_path"] == "torch"
          and pf["exchange_ms"]["torch"]["bits_384KB"] > 0 and pf["exchange_ms"]["torch"]["dense_12MB"] > 0)
    # a world of the wrong size is refused before any collective
    try:
        preflight(comm, "cpu", comm.world_size + 1, rows=64 * comm.world_size * 32)
        ok = False
    except RuntimeError as e:
        ok = ok and "ranks" in str(e)
    return ok


class _FakeCabi:
    """Stands in for dist.CabiComm in preflight()'s C ABI section on gloo ranks: the two exchanges run over the torch path, and
    rank `bad_rank` (if any) finds wrong words in its copy of the first dense all-gather -- a RANK-LOCAL verdict."""

    class _Mod:
        GL_BFS_TALLY_RANK_WORDS = 64

        class DeviceBuffer:
            @staticmethod
            def from_torch(t):
                return t

        @staticmethod
        def sync():
            pass

    def __init__(self, comm, bad_rank):
        self.capi_module, self.comm, self.bad_rank, self.dense_calls = self._Mod, comm, bad_rank, 0
        self.gl = self

    def all_gather_bits_tally(self, bits, bnds, tally):
        import torch
        self.comm.all_gather_slices(bits, [b // 32 for b in bnds])
        W, n = self.comm.world_size, self._Mod.GL_BFS_TALLY_RANK_WORDS
        parts = [torch.zeros(n, dtype=tally.dtype) for _ in range(W)]
        self.comm.dist.all_gather(parts, tally[self.comm.rank * n:(self.comm.rank + 1) * n].clone(), group=self.comm.group)
        tally.copy_(torch.cat(parts))

    def all_gather_f32(self, dense, bnds):
        self.comm.all_gather_slices(dense, bnds)
        self.dense_calls += 1
        if self.comm.rank == self.bad_rank and self.dense_calls == 1:
            dense[0] += 1.0


def _body_preflight_cabi_lockstep(comm):
    from graphlily_amd.dist import preflight
    rows = 64 * comm.world_size * 32
    good = preflight(comm, "cpu", comm.world_size, cabi=_FakeCabi(comm, None), rows=rows, watchdog_s=60.0)
    ok = good["exchange_path"] == "cabi" and good["cabi_error"] is None and good["exchange_ms"]["cabi"]["dense_12MB"] > 0
    # ONE rank finds wrong words: every rank must leave the section at the same step and fall back together (the
    # failing rank used to jump to the final flag all-reduce while its peers sat in the next exchange -- a 60 s watchdog exit)
    bad = preflight(comm, "cpu", comm.world_size, cabi=_FakeCabi(comm, comm.world_size - 1), rows=rows, watchdog_s=60.0)
    ok = ok and bad["exchange_path"] == "torch" and bad["verified"] is True
    ok = ok and "dense all-gather (C ABI path, uneven bounds) failed on 1 rank(s)" in (bad["cabi_error"] or "")
    ok = ok and "cabi" not in bad["exchange_ms"]
    # ... and a rank whose section raises (the test hook) takes everybody with it just the same
    forced = preflight(comm, "cpu", comm.world_size, cabi=_FakeCabi(comm, None), rows=rows, fail_cabi=comm.rank == 0)
    ok = ok and forced["exchange_path"] == "torch" and "setup failed on 1 rank(s)" in (forced["cabi_error"] or "")
    return ok


@pytest.mark.parametrize("world", [2, 4])
def test_preflight_cabi_section_keeps_ranks_in_lockstep(world):
    assert all(_spawn("_body_preflight_cabi_lockstep", world).values())


@pytest.mark.parametrize("world", [2, 4, 8])
def test_preflight_verifies_both_all_gathers(world):
    """bench.py's first contact with its peers: the process group's size, a bit and a dense all-gather on uneven and on equal
    bounds with every word verified on every rank, the exchange timed apart from compute -- here on gloo ranks."""
    assert all(_spawn("_body_preflight", world).values())
