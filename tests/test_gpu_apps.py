"""GPU parity: BFS / PageRank / SSSP drivers on the HIP modules vs the oracle's app compositions.
Follows tests/test_app.cpp:51-135 (uniform_10K_10, source 0, 10 iterations, threshold 0.1, PageRank
damping 0.9) and adds power-law graphs and the shipped line_8 / eye_10 fixtures."""
import json
import os

import numpy as np
import pytest

from graphlily_amd import app, datasets, io, module as M
from oracle import oracle as O

from helpers import named_matrix, to_oracle, set_knob

pytestmark = pytest.mark.gpu


def _oracle_prepared(m, kind, damping=0.9):
    om = to_oracle(m)
    if kind == "sssp":
        O.sssp_preprocess(om)
    O.util_round_csr_matrix_dim(om, 128, 128)
    if kind == "pagerank":
        O.util_normalize_csr_matrix_by_outdegree(om)
        om.adj_data = (om.adj_data * np.float32(damping)).astype(np.float32)
    elif kind == "bfs":
        om.adj_data = np.ones(om.nnz, np.float32)
    return om


GRAPHS = ["uniform_10K_10", "rmat_20K", "rmat_sym_50K"]


@pytest.mark.parametrize("name", GRAPHS)
def test_bfs(gpu, name):
    m = named_matrix(name)
    ref = O.bfs(_oracle_prepared(m, "bfs"), 0, 10)
    bfs = app.BFS(M.num_hbm_channels, 1024, 512, 256)
    bfs.set_target("hw")
    bfs.set_up_runtime("unused.xclbin")
    bfs.load_and_format_matrix(m, True)
    bfs.send_matrix_host_to_device()
    for thr in (0.1, 0.001, 1.0):
        assert np.array_equal(bfs.pull_push(0, 10, thr), ref), "pull_push thr %g" % thr
    assert np.array_equal(bfs.pull(0, 10), ref), "pull"
    assert np.array_equal(bfs.push(0, 10), ref), "push"
    assert ref.max() > 2, "source must reach something for the test to mean anything"


@pytest.mark.parametrize("name", GRAPHS)
def test_pagerank(gpu, name):
    m = named_matrix(name)
    om = _oracle_prepared(m, "pagerank")
    ref = O.pagerank(om, 0.9, 10)
    pr = app.PageRank(M.num_hbm_channels, 1024, 256)
    pr.set_up_runtime()
    pr.load_and_format_matrix(m, 0.9, True)
    pr.send_matrix_host_to_device()
    got = pr.pull(0.9, 10)
    # float64 evaluation of the same recurrence on the same (float32) matrix
    import scipy.sparse as sp
    A = sp.csr_matrix((om.adj_data.astype(np.float64), om.adj_indices, om.adj_indptr), shape=(om.num_rows, om.num_cols))
    r = np.full(om.num_rows, np.float64(np.float32(1.0 / om.num_rows)))
    tele = np.float64(np.float32(np.float32(1) - np.float32(0.9)) / np.float32(om.num_rows))
    for _ in range(10):
        r = A @ r + tele
    # north_star tolerance: 1e-5 relative for float PageRank -- against the exact recurrence ...
    assert np.allclose(got, r, rtol=1e-5, atol=0), "vs float64 recurrence"
    # ... and against the fp32 sequential oracle, which itself drifts from it on hub rows
    # (10 iterations x sequential accumulation over rows of 1e3..1e4 entries): 1e-4 relative
    assert np.allclose(got, ref, rtol=1e-4, atol=0), "vs fp32 oracle"
    if name == "uniform_10K_10":
        assert np.allclose(got, ref, rtol=1e-5, atol=0), "short rows: 1e-5 against the oracle as well"


@pytest.mark.parametrize("zero", [255.0, 999999999.0])
@pytest.mark.parametrize("name", GRAPHS)
def test_sssp(gpu, name, zero):
    m = named_matrix(name)
    ref = O.sssp(_oracle_prepared(m, "sssp"), 0, 10, zero)
    s = app.SSSP(M.num_hbm_channels, 1024, 512, 256, semiring=M.SemiringType(M.kAddMin, 0.0, zero))
    s.set_up_runtime()
    s.load_and_format_matrix(m, True)
    s.send_matrix_host_to_device()
    assert np.array_equal(s.pull(0, 10), ref), "pull"
    # the reference checks push and pull_push against the same pull-form composition
    # (tests/test_app.cpp:118-132); with unit weights and the source's self edge both forms agree
    for got, what in ((s.pull_push(0, 10, 0.1), "pull_push"), (s.push(0, 10), "push")):
        assert np.array_equal(got, ref), what


def test_torch_backed_buffers(gpu):
    """The bench path: vectors owned by torch tensors, library on torch's stream."""
    torch = pytest.importorskip("torch")
    m = named_matrix("rmat_20K")
    ref = O.bfs(_oracle_prepared(m, "bfs"), 0, 8)
    bfs = app.BFS(backend=app.HipBackend(0, use_torch=True))
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(m, True)
    bfs.send_matrix_host_to_device()
    try:
        assert np.array_equal(bfs.pull_push(0, 8, 0.01), ref)
        assert np.array_equal(bfs.pull(0, 8), ref)
    finally:
        from graphlily_amd import capi
        capi.reset_stream()


def test_golden_apps(gpu, golden_dir):
    G = json.load(open(os.path.join(golden_dir, "reference_known_answers.json")))
    for a in G["survey_8c"]["apps"]:
        path = os.path.join(golden_dir, a["matrix"] + "_csr_float32.npz")
        exp = np.array(a["first"], dtype=np.float32)
        if a["call"] == "bfs":
            d = app.BFS(16, 1024, 512, 256)
            d.set_up_runtime()
            d.load_and_format_matrix(path, True)
            d.send_matrix_host_to_device()
            for got in (d.pull(a["source"], a["iters"]), d.push(a["source"], a["iters"]),
                        d.pull_push(a["source"], a["iters"], 0.05)):
                assert got.shape[0] == 128 and got[:len(exp)].tolist() == exp.tolist(), a
        elif a["call"] == "pagerank":
            d = app.PageRank(16, 1024, 256)
            d.set_up_runtime()
            d.load_and_format_matrix(path, a["damping"], True)
            d.send_matrix_host_to_device()
            got = d.pull(a["damping"], a["iters"])
            assert np.allclose(got[:len(exp)], exp, rtol=1e-5, atol=0), a
        elif a["call"] == "sssp":
            d = app.SSSP(16, 1024, 512, 256, semiring=M.TropicalSemiringUfixed)
            d.set_up_runtime()
            d.load_and_format_matrix(path, True)
            d.send_matrix_host_to_device()
            assert d.pull(a["source"], a["iters"])[:len(exp)].tolist() == exp.tolist(), a


def test_time_breakdown_variants(gpu, capsys):
    """app/bfs.h:222-347 and app/pagerank.h:93-147: same results as the plain drivers, buckets reported."""
    raw = datasets.rmat(20000, 300000, 11, True)
    bfs = app.BFS(16, 0, 0, 0)
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(raw.copy(), True)
    bfs.send_matrix_host_to_device()
    src = int(np.argmax(np.diff(raw.adj_indptr.astype(np.int64)) > 0))
    d0 = bfs.pull_push(src, 8, 0.001)
    d1 = bfs.pull_push_time_breakdown(src, 8, 0.001)
    assert np.array_equal(d0, d1)
    tb = bfs.time_breakdown_
    assert tb["total"] > 0 and abs(tb["total"] - tb["spmv_spmspv"] - tb["assign"] - tb["data_transfer"] - tb["overhead"]) < 1e-6
    pr = app.PageRank(16, 0, 0)
    pr.set_up_runtime()
    pr.load_and_format_matrix(raw.copy(), 0.9, True)
    pr.send_matrix_host_to_device()
    r0 = pr.pull(0.9, 5)
    r1 = pr.pull_time_breakdown(0.9, 5)
    assert np.array_equal(r0, r1)
    assert pr.time_breakdown_["spmv"] > 0
    out = capsys.readouterr().out
    assert "spmv_spmspv_time_ms" in out and "spmv_time_ms per iteration" in out


def test_fused_bfs_pull_step_equals_the_three_calls(gpu, monkeypatch):
    """gl_bfs_pull_step = masked (||,&&) SpMV + eWiseAdd(+0) + dense assign(level) + packing of the next frontier
    (app/bfs.h:118-123) in one launch.  Whole BFS runs with and without it must give the same levels (and the
    oracle's), and one step through the C ABI must leave the same distance vector and frontier as the module calls."""
    from graphlily_amd import capi
    raw = datasets.rmat(30000, 500000, 17, True)
    src = int(np.argmax(np.diff(raw.adj_indptr.astype(np.int64)) > 0))
    runs = {}
    for host_loop in ("0", "1"):      # the device-resident schedule / the reference's module-call loop (fused pull step inside)
        monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", host_loop)
        bfs = app.BFS(16, 0, 0, 0)
        bfs.set_up_runtime()
        bfs.load_and_format_matrix(raw.copy(), True)
        bfs.send_matrix_host_to_device()
        runs[host_loop] = [bfs.pull(src, 8), bfs.pull_push(src, 8, 0.01)]
        assert bfs.fused_
    for a, b in zip(runs["1"], runs["0"]):
        assert np.array_equal(a, b)
    om = to_oracle(raw)
    O.util_round_csr_matrix_dim(om, 128, 128)
    om.adj_data[:] = 1
    assert np.array_equal(runs["1"][0], O.bfs(om, src, 8))
    # one step, C ABI against the oracle's three operators
    m = raw.copy()
    io.util_round_csr_matrix_dim(m, 128, 128)
    n = m.num_rows
    plan = capi.SpMVPlan(n, n, m.adj_indptr, m.adj_indices, np.ones(m.nnz, np.float32), flags=capi.GL_PLAN_BOOLEAN)
    rng = np.random.default_rng(2)
    x = (rng.random(n) < 0.02).astype(np.float32)
    dist = np.where(rng.random(n) < 0.3, np.float32(2.0), np.float32(0.0)).astype(np.float32)
    words = plan.bits_words()
    dx, dd = capi.DeviceBuffer(4 * n), capi.DeviceBuffer(4 * n)
    b_in, b_out = capi.DeviceBuffer(4 * words), capi.DeviceBuffer(4 * words)
    dx.write(x)
    dd.write(dist)
    b_in.write(np.zeros(words, np.uint32))
    b_out.write(np.full(words, 0xFFFFFFFF, np.uint32))     # every word of the row range must be overwritten
    capi.pack_bits(dx, n, b_in)
    plan.bfs_pull_step(b_in, b_out, dd, 5.0)
    y = O.spmv(to_oracle(m), x, 1, 0.0, dist, O.WRITETOZERO)   # a_ij are 1 in `m`? use unit weights
    om2 = to_oracle(m)
    om2.adj_data[:] = 1
    y = O.spmv(om2, x, 1, 0.0, dist, O.WRITETOZERO)
    want_dist = dist.copy()
    want_dist[y != 0] = 5.0
    assert np.array_equal(dd.read(np.float32, n), want_dist)
    bits = b_out.read(np.uint32, words)
    got_front = ((bits[np.arange(n) >> 5] >> (np.arange(n) & 31).astype(np.uint32)) & 1).astype(np.float32)
    assert np.array_equal(got_front, y)
    assert not bits[(n + 31) // 32:].any() or np.all(bits[(n + 31) // 32:] == 0xFFFFFFFF)   # words past the rows untouched


@pytest.mark.parametrize("name", ["rmat_sym_50K", "uniform_10K_10"])
def test_bfs_pull_push_device_loop_equals_host_loop(gpu, name, monkeypatch):
    """pull_push with the direction decided on the device (no read-back per iteration; the schedule is
    captured as a hipGraph from the second call on) gives the oracle's distances and switches direction after the same
    number of push iterations as the host-driven loop of the reference (app/bfs.h:180-190), for several sources and
    thresholds, eagerly and replayed."""
    m = named_matrix(name)
    om = _oracle_prepared(m, "bfs")
    bfs = app.BFS(M.num_hbm_channels, 0, 0, 0)
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(m, True)
    bfs.send_matrix_host_to_device()
    assert bfs._bits_loop_ok()
    deg = np.diff(m.adj_indptr.astype(np.int64))
    sources = [0, int(np.argmax(deg)), int(np.nonzero(deg > 0)[0][-1])]
    for thr in (0.001, 0.05, 1.0):
        for rep in range(3):                       # eager, capture, replay
            for src in sources:
                monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "0")
                got = bfs.pull_push(src, 8, thr)
                pushes = bfs.push_iterations_
                assert np.array_equal(got, O.bfs(om, src, 8)), "thr %g rep %d src %d" % (thr, rep, src)
                monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "1")
                ref = bfs.pull_push(src, 8, thr)
                assert np.array_equal(ref, got)
                assert bfs.push_iterations_ == pushes, "thr %g src %d: device %d vs host %d push iterations" % (
                    thr, src, pushes, bfs.push_iterations_)
    assert any(bfs.bits_loop_["graphs"].values()), "the schedule was captured as a graph"


@pytest.mark.parametrize("name", ["rmat_sym_50K", "uniform_10K_10"])
def test_bfs_pull_push_returns_to_push(gpu, name, monkeypatch):
    """An extension of the device-resident schedule: once the reference's rule has switched to pulling, every later slot is
    handed back to the push step, which leaves heavy frontiers to the streaming pull (the reference pulls to the end).
    Distances do not depend on the direction: every threshold / source combination must give the oracle's result, eagerly
    and replayed, and the reference's push count (the first push phase) must equal the host-driven loop's."""
    m = named_matrix(name)
    om = _oracle_prepared(m, "bfs")
    bfs = app.BFS(M.num_hbm_channels, 0, 0, 0)
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(m, True)
    bfs.send_matrix_host_to_device()
    assert bfs._bits_loop_ok()
    deg = np.diff(m.adj_indptr.astype(np.int64))
    sources = [0, int(np.argmax(deg)), int(np.nonzero(deg > 0)[0][-1])]
    again = 0
    for thr in (0.001, 0.05):
        for rep in range(3):                       # eager, capture, replay
            for src in sources:
                monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "0")
                got = bfs.pull_push(src, 9, thr)
                pushes, again = bfs.push_iterations_, again + bfs.push_iterations_again_
                assert np.array_equal(got, O.bfs(om, src, 9)), "thr %g rep %d src %d" % (thr, rep, src)
                monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "1")
                assert np.array_equal(bfs.pull_push(src, 9, thr), got)
                assert bfs.push_iterations_ == pushes
    assert again > 0, "some pull step must have handed the loop back to pushing"


def test_bfs_bits_schedule_long_columns_and_heavy_frontiers(gpu, monkeypatch):
    """gl_bfs_bits_*: columns of 4096 entries and more are served from the plan's static chunk list (two hub vertices with
    ~6000 and ~9000 neighbours, one of them the source), and a push whose frontier holds more than 1/PULL_DIV of the
    non-zeros is left to the pull step of its slot (forced on and off through GRAPHLILY_DEBUG bfs_heavy_div).  Distances and
    the reference's push count must not depend on any of it."""
    rng = np.random.default_rng(5)
    n = 40000
    rows = [rng.integers(0, n, size=12 * n), np.full(6000, 7), np.full(9000, 11)]
    cols = [rng.integers(0, n, size=12 * n), rng.choice(n, 6000, replace=False), rng.choice(n, 9000, replace=False)]
    r, c = np.concatenate(rows), np.concatenate(cols)
    r, c = np.concatenate([r, c]), np.concatenate([c, r])       # symmetric
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = (key // n).astype(np.uint32), (key % n).astype(np.uint32)
    indptr = np.zeros(n + 1, np.uint32)
    np.add.at(indptr, r.astype(np.int64) + 1, 1)
    indptr = np.cumsum(indptr).astype(np.uint32)
    m = io.CSRMatrix(n, n, np.ones(len(c), np.float32), c, indptr)
    om = _oracle_prepared(m, "bfs")
    for div in ("128", "1000000", "1"):
        set_knob(monkeypatch, "bfs_heavy_div", div)
        bfs = app.BFS(M.num_hbm_channels, 0, 0, 0)
        bfs.set_up_runtime()
        bfs.load_and_format_matrix(m, True)
        bfs.send_matrix_host_to_device()
        monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "0")
        assert bfs._bits_loop_ok()
        for thr in (0.001, 0.3, 1.0):
            for rep in range(3):
                for src in (7, 11, 0, 12345):
                    monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "0")
                    got = bfs.pull_push(src, 7, thr)
                    pushes = bfs.push_iterations_
                    assert np.array_equal(got, O.bfs(om, src, 7)), "div %s thr %g rep %d src %d" % (div, thr, rep, src)
                    monkeypatch.setenv("GRAPHLILY_BFS_HOST_LOOP", "1")
                    assert np.array_equal(bfs.pull_push(src, 7, thr), got)
                    assert bfs.push_iterations_ == pushes, "div %s thr %g src %d" % (div, thr, src)


def test_bfs_bottom_up_with_an_unreachable_hub(gpu, monkeypatch):
    """The bottom-up pull of the bit-frontier schedule (gl_bfs_bits_push_step with the pull plan's rows) visits only the
    rows not reached yet.  A hub the BFS never reaches (a star of 30 000 leaves in another component) stays in that set for
    the whole run: its row is finished by a whole wavefront, not by one thread.  pull, pull_push with the bottom-up step
    forced on for every non-scattering slot (BU_DIV 1) and switched off (0) must give the oracle's distances."""
    rng = np.random.default_rng(9)
    n = 60000
    r = [rng.integers(0, 30000, size=400000), np.full(30000, 30000)]
    c = [rng.integers(0, 30000, size=400000), np.arange(30001, 60001) % n]
    r, c = np.concatenate(r), np.concatenate(c)
    r, c = np.concatenate([r, c]), np.concatenate([c, r])
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = (key // n).astype(np.uint32), (key % n).astype(np.uint32)
    indptr = np.zeros(n + 1, np.uint32)
    np.add.at(indptr, r.astype(np.int64) + 1, 1)
    indptr = np.cumsum(indptr).astype(np.uint32)
    m = io.CSRMatrix(n, n, np.ones(len(c), np.float32), c, indptr)
    om = _oracle_prepared(m, "bfs")
    for div in ("3", "1", "0"):
        set_knob(monkeypatch, "bfs_bu_div", div)
        bfs = app.BFS(M.num_hbm_channels, 0, 0, 0)
        bfs.set_up_runtime()
        bfs.load_and_format_matrix(m, True)
        bfs.send_matrix_host_to_device()
        assert bfs._bits_loop_ok()
        for src in (0, 5, 30000, 30001):
            ref = O.bfs(om, src, 8)
            for rep in range(3):
                assert np.array_equal(bfs.pull(src, 8), ref), "pull div %s src %d rep %d" % (div, src, rep)
                assert np.array_equal(bfs.pull_push(src, 8, 0.01), ref), "pull_push div %s src %d rep %d" % (div, src, rep)
    counts = bfs.bfs_slot_counts_
    assert counts.shape[0] == 8


@pytest.mark.parametrize("zero", [255.0, 999999999.0])
@pytest.mark.parametrize("name", ["rmat_sym_50K", "gplus_small"])
def test_sssp_pull_push_on_completion_records(gpu, name, zero):
    """SSSP::pull_push (app/sssp.h:197-243): the loop's count comes from the SpMSpV's completion record (gl_spmspv_wait), read
    behind the relax step.  The oracle's distances for thresholds that end the push phase after every possible iteration,
    the push counts a plain re-statement of the loop condition gives, repeated calls on one object."""
    m = named_matrix(name)
    iters = 7
    om = _oracle_prepared(m, "sssp")
    ref = O.sssp(om, 0, iters, zero)
    sem = M.SemiringType(M.kAddMin, 0.0, zero)
    s = app.SSSP(M.num_hbm_channels, 1024, 512, 256, semiring=sem)
    s.set_up_runtime()
    s.load_and_format_matrix(m, True)
    s.send_matrix_host_to_device()
    push = app.SSSP(M.num_hbm_channels, 1024, 512, 256, semiring=sem)     # the same loop with the count copied back: SSSP.push
    push.set_up_runtime()
    push.load_and_format_matrix(m, True)
    push.send_matrix_host_to_device()
    seen = set()
    for thr in (0.0, 1e-4, 1e-3, 1e-2, 0.05, 0.3, 2.0):
        for rep in range(3):
            got = s.pull_push(0, iters, thr)
            assert np.array_equal(got, ref), (name, zero, thr, rep)
        seen.add(s.push_iterations_)
    assert len(seen) >= 3 and 1 in seen and (iters - 1) in seen, seen
    assert np.array_equal(push.push(0, iters), ref)
    assert np.array_equal(s.pull_push(0, 10, 0.01), O.sssp(om, 0, 10, zero))


def test_bfs_byte_read_back_equals_the_float_one(gpu, monkeypatch):
    """BFS levels cross PCIe as bytes when they fit (gl_levels_to_u8 + gl_host_expand_u8_f32) and as floats otherwise
    (GRAPHLILY_BFS_U8=0, or more than 254 iterations): the same vector either way, equal to the oracle."""
    from graphlily_amd import app, datasets, io
    from oracle import oracle as O
    from helpers import to_oracle
    g = datasets.rmat(70000, 1200000, seed=41, symmetric=True)
    m = g.copy()
    io.util_round_csr_matrix_dim(m, 128, 128)
    m.adj_data = np.ones(m.nnz, np.float32)
    src = int(np.argmax(np.diff(g.adj_indptr.astype(np.int64)) > 0))
    bfs = app.BFS(16, 0, 0, 0)
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(g, True)
    bfs.send_matrix_host_to_device()
    ref = O.bfs(to_oracle(m), src, 7)
    for u8 in ("2", "0", "1", "2"):      # packed pinned / floats / whichever measures faster / packed
        monkeypatch.setenv("GRAPHLILY_BFS_U8", u8)
        for _ in range(12 if u8 == "1" else 3):
            got = bfs.pull_push(src, 7, 0.01)
            assert got.dtype == np.float32 and np.array_equal(got, ref)
            assert np.array_equal(bfs.pull(src, 7), ref)
    # a line graph walked for 300 iterations: levels beyond a byte -- the float read-back serves them
    n = 600064
    line = io.CSRMatrix(n, n, np.ones(n - 1, np.float32), np.arange(n - 1, dtype=np.uint32),
                        np.concatenate([[0], np.arange(n, dtype=np.uint32)]).astype(np.uint32))
    monkeypatch.setenv("GRAPHLILY_BFS_U8", "2")
    b2 = app.BFS(16, 0, 0, 0)
    b2.set_up_runtime()
    b2.load_and_format_matrix(line, True)
    b2.send_matrix_host_to_device()
    lm = line.copy()
    io.util_round_csr_matrix_dim(lm, 128, 128)
    ref2 = O.bfs(to_oracle(lm), 0, 300)
    assert ref2.max() > 255
    assert np.array_equal(b2.pull(0, 300), ref2)
    ref200 = O.bfs(to_oracle(lm), 0, 200)
    assert np.array_equal(b2.pull(0, 200), ref200)      # (and with bytes again)
    # the packed levels streamed into the host block chunk by chunk (the default) and the round-5 way: pack, copy, wait, expand
    ref9 = O.bfs(to_oracle(lm), 0, 9)
    for stream in ("0", "1", "0", "1"):
        monkeypatch.setenv("GRAPHLILY_BFS_STREAM", stream)
        for _ in range(4):               # eager, capture, replays
            assert np.array_equal(b2.pull(0, 200), ref200)
            assert np.array_equal(b2.pull_push(0, 9, 0.01), ref9)      # (nibbles)
        assert b2.readback_["way"] == "packed"


def test_bench_line_carries_the_packed_bfs_read_back(gpu):
    """bench.py on a stand-in large enough for the packed read-back (>= 2^19 rows): the line's `bfs` object is complete -- both
    modes, the read-back's choice, the host half timed alone -- and the headline repeats its numbers (round 6: a changed driver
    field made this leg fail into {"error": ...} while every smaller bench test stayed green)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--scale", "0.25", "--bfs-runs", "2",
                        "--no-cpu-baseline", "--no-six-graphs", "--no-spmspv", "--no-pattern"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    bfs = rec["bfs"]
    assert "error" not in bfs, bfs
    assert bfs["pull"]["reached"] == bfs["pull_push"]["reached"] > 0
    assert bfs["host_unpack_ms"] > 0 and bfs["host_unpack"]["levels"] >= 1 << 19 and bfs["host_unpack"]["streamed_in_chunks"] is True
    assert bfs["host_unpack"]["bytes_over_pcie"] < bfs["host_unpack"]["float_bytes"] // 4
    assert rec["headline"]["bfs_pull_push_ms"] == bfs["pull_push"]["ms"] > 0


@pytest.mark.parametrize("name", ["rmat_sym_50K", "gplus_small", "uniform_10K_10"])
def test_chained_spmv_runs_are_bit_identical(gpu, name, monkeypatch):
    """gl_spmv_plan_chain (include/graphlily_hip.h): PageRank.pull and SSSP.pull feed every result straight back as the next
    vector; chained, a run's epilogue leaves y in the next run's packed form (its slots of the packed vector and the hot table,
    times the column's value in pattern plans) and the next run skips its helper launch.  Same bits as the unchained loop,
    for both layouts the apps use, with the streaming helper forced (the planner picks it from a quarter of the columns on);
    and a vector written between two runs is served by the helper again once the chain is off."""
    from graphlily_amd import capi
    set_knob(monkeypatch, "spmv_helper", 1)
    m = named_matrix(name)
    outs = {}
    for chain in ("0", "1"):
        monkeypatch.setenv("GRAPHLILY_SPMV_CHAIN", chain)
        pr = app.PageRank(M.num_hbm_channels, 1024, 256)
        pr.set_up_runtime()
        pr.load_and_format_matrix(m, 0.9, True)
        pr.send_matrix_host_to_device()
        if chain == "1":
            assert pr.SpMV_.chain(True) is True, "the plan must be able to chain for this test to mean anything"
            pr.SpMV_.chain(False)
        s = app.SSSP(M.num_hbm_channels, 1024, 512, 256, semiring=M.SemiringType(M.kAddMin, 0.0, 255.0))
        s.set_up_runtime()
        s.load_and_format_matrix(m, True)
        s.send_matrix_host_to_device()
        outs[chain] = (pr.pull(0.9, 7), pr.pull(0.9, 2), s.pull(0, 9), s.pull_push(0, 9, 0.1))
    for a, b, what in zip(outs["0"], outs["1"], ("pagerank 7", "pagerank 2", "sssp pull", "sssp pull_push")):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), what
    assert np.array_equal(outs["1"][2], O.sssp(_oracle_prepared(m, "sssp"), 0, 9, 255.0))
    # the plan level: x = the previous y is only trusted while the chain is on
    rng = np.random.default_rng(3)
    mm = m.copy()
    io.util_round_csr_matrix_dim(mm, 128, 128)
    n = mm.num_rows
    data = rng.random(mm.nnz, dtype=np.float32)
    plan = capi.SpMVPlan(n, n, mm.adj_indptr, mm.adj_indices, data, 0, n, flags=capi.GL_PLAN_KEEP_VALUES)
    x0 = rng.random(n, dtype=np.float32)
    a, b = capi.DeviceBuffer.from_host(x0), capi.DeviceBuffer(4 * n)
    def three_runs(chained):
        a.write(x0)
        if chained:
            assert plan.chain(True)
        plan.run(a, None, b, 0, 0.0, 0)
        plan.run(b, None, a, 0, 0.0, 0)          # (chained: no helper, x comes packed from the first run's epilogue)
        plan.run(a, None, b, 0, 0.0, 0)
        if chained:
            plan.chain(False)
        return b.read(np.float32, n)
    plain, chained = three_runs(False), three_runs(True)
    assert np.array_equal(plain.view(np.uint32), chained.view(np.uint32))
    # off again: a vector rewritten by the caller is read afresh
    plan.run(a, None, b, 0, 0.0, 0)
    y1 = b.read(np.float32, n)
    b.write(x0)
    plan.run(b, None, a, 0, 0.0, 0)
    a2 = a.read(np.float32, n)
    a.write(x0)
    plan.run(a, None, b, 0, 0.0, 0)
    assert np.array_equal(b.read(np.float32, n).view(np.uint32), a2.view(np.uint32)) and y1 is not None
