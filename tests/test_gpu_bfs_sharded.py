"""GPU parity of the ROW-SHARDED bit-frontier BFS schedule (reference loop app/bfs.h:160-219), one rank
at a time on the one GPU of the test box: `EmulatedComm` stands in for the all-gather by copying the other ranks' rows of
every exchanged bit vector from a whole-matrix run, so rank k of N executes exactly the kernels, sees exactly the frontiers
and must take exactly the decisions of a real N-rank run -- shard plans (split ones included), deferred decisions
(gl_bfs_bits_decide on the gathered vector), bottom-up slots on a shard's rows, slice-only read-back.  Every rank's slice
must equal the oracle's distances bit for bit and the control words (push iterations, vertices per slot) the one-GPU run's."""
import numpy as np
import pytest

from graphlily_amd import app, capi, datasets, io
from graphlily_amd.dist import EmulatedComm
from oracle import oracle as O

from helpers import to_oracle, set_knob

pytestmark = pytest.mark.gpu


def _prepared(g):
    m = g.copy()
    io.util_round_csr_matrix_dim(m, 128, 128)
    m.adj_data = np.ones(m.nnz, np.float32)
    return m


def _whole(g):
    bfs = app.BFS(16, 0, 0, 0)
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(g, True)
    bfs.send_matrix_host_to_device()
    return bfs


def _rank(g, k, world, whole, copy=False):
    comm = EmulatedComm(k, world, copy=copy)
    b = app.BFS(16, 0, 0, 0, comm=comm)
    b.set_up_runtime()
    b.load_and_format_matrix(g, True)
    b.send_matrix_host_to_device()
    b.gather_result_ = False
    st = whole.bits_loop_
    comm.set_truth(st["vecs"], st["words"])
    return b


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name", ["rmat_sym", "rmat_directed"])
def test_every_rank_of_a_sharded_bfs_matches_the_oracle(gpu, name, world):
    g = datasets.rmat(60000, 1500000, seed=31, symmetric=True) if name == "rmat_sym" else datasets.rmat(50000, 900000, seed=32, symmetric=False)
    deg = np.diff(g.adj_indptr.astype(np.int64))
    src = int(np.argmax(deg > 0))
    iters = 8
    ref = O.bfs(to_oracle(_prepared(g)), src, iters)
    whole = _whole(g)
    for threshold, mode in ((0.001, "pull_push"), (0.05, "pull_push"), (0.9, "pull_push"), (None, "pull")):
        run_whole = (lambda: whole.pull_push(src, iters, threshold)) if mode == "pull_push" else (lambda: whole.pull(src, iters))
        assert np.array_equal(run_whole(), ref)
        pushes, counts = whole.push_iterations_, whole.bfs_slot_counts_.copy()
        assert counts.sum() + 1 == (ref != 0).sum()
        covered = 0
        for k in range(world):
            # (with and without the stand-in for the exchange step: copies of the other ranks' rows / the whole run's vectors read in place)
            b = _rank(g, k, world, whole, copy=(k % 2 == 1))
            for rep in range(3):                 # enqueued, captured, replayed
                run_whole()                      # (the truth vectors are the whole run's, rewritten by every run)
                got = b.pull_push(src, iters, threshold) if mode == "pull_push" else b.pull(src, iters)
                r0, r1 = b.result_range_
                assert (r0, r1) == (b.bounds_[k], b.bounds_[k + 1]) and got.shape[0] == r1 - r0
                assert np.array_equal(got, ref[r0:r1]), "rank %d/%d %s thr %s rep %d: %d rows differ" % (
                    k, world, mode, threshold, rep, int((got != ref[r0:r1]).sum()))
                assert b.push_iterations_ == pushes, "the same decisions as the one-GPU run"
                assert np.array_equal(b.bfs_slot_counts_, counts)
            # the rank's own tallies of every slot (gl_bfs_bits_shard_step: what it would have sent along with its rows of the
            # bit vector) are the host's sums over the whole run's vectors
            tab = b.bits_loop_["tally"].read(np.uint32)
            truth_tab = getattr(b.comm, "truth_tally_host", None)
            H, R = capi.GL_BFS_TALLY_HEAD_WORDS, capi.GL_BFS_TALLY_RANK_WORDS
            for s in range(1, iters + 1) if truth_tab is not None else ():
                blk = tab[H + ((s - 1) * world + k) * R:H + ((s - 1) * world + k + 1) * R].reshape(8, 8)
                want = truth_tab[H + ((s - 1) * world + k) * R:][:6]
                assert int(blk[:, 0].sum()) == int(want[0]), (s, k)
                assert int(blk[:, 2:4].copy().view(np.uint64).sum()) == int(want[2:4].copy().view(np.uint64)[0]), (s, k)
                assert int(blk[:, 4:6].copy().view(np.uint64).sum()) == int(want[4:6].copy().view(np.uint64)[0]), (s, k)
            # the rank's own rows of every vector it wrote are the whole run's (what an all-gather would have published)
            words = whole.bits_loop_["words"]
            mine = b.bits_loop_["vecs"].read(np.uint32).reshape(-1, words)[1:iters + 2, r0 // 32:r1 // 32]
            truth = whole.bits_loop_["vecs"].read(np.uint32).reshape(-1, words)[1:iters + 2, r0 // 32:r1 // 32]
            assert np.array_equal(mine, truth)
            covered += r1 - r0
            del b
        assert covered == ref.shape[0]


def test_sharded_bfs_on_split_shard_plans(gpu, monkeypatch):
    """Shard plans cut into column segments (several units share a block's rows): the pull step claims rows with atomicOr."""
    set_knob(monkeypatch, "spmv_segments", "3")
    g = datasets.rmat(60000, 1500000, seed=33, symmetric=True)
    src = int(np.argmax(np.diff(g.adj_indptr.astype(np.int64)) > 0))
    ref = O.bfs(to_oracle(_prepared(g)), src, 7)
    set_knob(monkeypatch, "spmv_segments", None)
    whole = _whole(g)
    assert np.array_equal(whole.pull(src, 7), ref)
    set_knob(monkeypatch, "spmv_segments", "3")
    for k in range(4):
        b = _rank(g, k, 4, whole)
        assert b.SpMV_.plan_.info()["segments"] > 1
        for fn in (lambda: b.pull(src, 7), lambda: b.pull_push(src, 7, 0.001)):
            whole.pull(src, 7)
            got = fn()
            r0, r1 = b.result_range_
            assert np.array_equal(got, ref[r0:r1])


def test_a_world_of_one_equals_the_one_gpu_run(gpu):
    """The same launches on one GPU and on a "shard" that is the whole matrix (a world of one, exchange stubbed): distances, push
    iterations (first phase and after a pull handed back), per-slot counts and modes, for three thresholds; and the
    reference's module-call loop (GRAPHLILY_BFS_HOST_LOOP) agrees on distances and the first push phase."""
    import os
    g = datasets.rmat(80000, 2400000, seed=34, symmetric=True)
    src = int(np.argmax(np.diff(g.adj_indptr.astype(np.int64)) > 0))
    whole = _whole(g)
    for thr in (0.0005, 0.01, 0.2):
        ref = whole.pull_push(src, 9, thr).copy()
        pushes, again, counts, modes = whole.push_iterations_, whole.push_iterations_again_, whole.bfs_slot_counts_.copy(), whole.bfs_slot_modes_.copy()
        for rep in range(3):                 # enqueued, captured, replayed
            got = whole.pull_push(src, 9, thr)
            assert np.array_equal(got, ref)
            assert (whole.push_iterations_, whole.push_iterations_again_) == (pushes, again)
            assert np.array_equal(whole.bfs_slot_counts_, counts) and np.array_equal(whole.bfs_slot_modes_, modes)
        b = _rank(g, 0, 1, whole)
        got = b.pull_push(src, 9, thr)
        assert np.array_equal(got, ref)
        assert (b.push_iterations_, b.push_iterations_again_) == (pushes, again)
        assert np.array_equal(b.bfs_slot_counts_, counts)
        os.environ["GRAPHLILY_BFS_HOST_LOOP"] = "1"
        try:
            assert np.array_equal(whole.pull_push(src, 9, thr), ref) and whole.push_iterations_ == pushes
        finally:
            del os.environ["GRAPHLILY_BFS_HOST_LOOP"]


def test_shard_step_refuses_what_it_cannot_run(gpu):
    """gl_bfs_bits_shard_step fails loudly: a row plan without the bit layout, plans of different row ranges, a slot of 0."""
    from graphlily_amd import module as M
    g = datasets.rmat(20000, 300000, seed=35, symmetric=True)
    whole = _whole(g)
    src = int(np.argmax(np.diff(g.adj_indptr.astype(np.int64)) > 0))
    whole.pull_push(src, 4, 0.01)
    st = whole.bits_loop_
    csc_plan, rows_plan = whole.SpMSpV_.plan_, whole.SpMV_.plan_
    args = lambda rows, slot: (csc_plan, rows, st["bits"][1], st["bits"][2], st["words"], st["distance"], 2.0, st["ctl"], st["tally"], None,
                               slot, 0, 1, st["col_len"], whole.nnz_global_, 0.01, 0, 1.0)
    m = _prepared(g)
    general = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, np.full(m.nnz, 0.5, np.float32), flags=capi.GL_PLAN_KEEP_VALUES)
    with pytest.raises(capi.GraphLilyError, match="GL_PLAN_BOOLEAN"):
        capi.bfs_bits_shard_step(*args(general, 1))
    half = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, 0, m.num_rows // 128 * 64,
                         flags=capi.GL_PLAN_BOOLEAN | capi.GL_PLAN_NO_MULADD)
    with pytest.raises(capi.GraphLilyError):
        capi.bfs_bits_shard_step(*args(half, 1))
    with pytest.raises(capi.GraphLilyError):
        capi.bfs_bits_shard_step(*args(rows_plan, 0))
    # and the schedule still runs afterwards
    ref = O.bfs(to_oracle(m), src, 4)
    assert np.array_equal(whole.pull_push(src, 4, 0.01), ref)


def test_dense_frontiers_pushed_through_the_scatter_body(gpu, monkeypatch):
    """The scattering push of the one-launch slot where it normally never goes: frontiers of thousands of vertices whose
    columns all hold 129 ... 1023 entries (set aside for the whole workgroup; the list of 256 per workgroup overflows and the
    wavefronts apply the rest themselves), never handed to the streaming pull (GRAPHLILY_DEBUG bfs_heavy_div=0) nor to the
    bottom-up scan (bfs_bu_div=0), threshold above 1: every slot pushes.  One GPU and every rank of 2 and of 16
    (16 ranks x 8 tally lines: the prologue's loop over more than 64 lines)."""
    set_knob(monkeypatch, "bfs_heavy_div", "0")
    set_knob(monkeypatch, "bfs_bu_div", "0")
    g = datasets.uniform(20480, 200, seed=36)
    src = 5
    ref = O.bfs(to_oracle(_prepared(g)), src, 5)
    whole = _whole(g)
    for rep in range(3):
        assert np.array_equal(whole.pull_push(src, 5, 2.0), ref)
    assert whole.push_iterations_ == 4 and list(whole.bfs_slot_modes_[:3]) == [1, 1, 1]
    for world in (2, 16):
        for k in range(world):
            b = _rank(g, k, world, whole, copy=(k % 2 == 0))
            whole.pull_push(src, 5, 2.0)
            got = b.pull_push(src, 5, 2.0)
            r0, r1 = b.result_range_
            assert np.array_equal(got, ref[r0:r1]), (world, k)
            assert b.push_iterations_ == whole.push_iterations_
            assert np.array_equal(b.bfs_slot_counts_, whole.bfs_slot_counts_)
