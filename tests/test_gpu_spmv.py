"""GPU parity: SpMVModule through the C ABI vs the CPU oracle.
Cases follow the reference's TEST(SpMV, MultipleCases) (tests/test_module_spmv_spmspv.cpp:137-178):
{Arithmetic, Logical} x {NoMask, WriteToZero, WriteToOne} x skip_empty_rows on dense_32, Arithmetic x 3
masks on uniform_10K_10 -- widened with the Tropical semiring and power-law matrices (long rows, empty
rows, non-contiguous tiles)."""
import json
import os

import numpy as np
import pytest

from graphlily_amd import datasets, io, module as M
from oracle import oracle as O

from helpers import (MASKS, SEMIRINGS, arith_exact, assert_arith_parity, assert_parity, rand01, set_knob, spmv_prepare,
                     to_oracle)

pytestmark = pytest.mark.gpu


def _run_spmv(gpu, m, sem, mask_name, x, mask, skip_empty_rows=True, shard=None):
    op, zero = SEMIRINGS[sem]
    mod = M.SpMVModule(M.num_hbm_channels, 1024, 256)
    mod.set_semiring(M.SemiringType(op, 1.0, zero))
    mod.set_mask_type(MASKS[mask_name])
    mod.set_target("hw")
    mod.set_up_runtime("unused.xclbin")
    if shard:
        mod.set_row_shard(*shard)
    mod.load_and_format_matrix(m, skip_empty_rows)
    mod.send_matrix_host_to_device()
    mod.send_vector_host_to_device(x)
    mod.send_mask_host_to_device(mask)
    mod.run()
    return mod.send_results_device_to_host()


def _check(got, m, sem, mask_name, x, mask, what):
    ref = _ref_spmv(m, sem, mask_name, x, mask)
    op = SEMIRINGS[sem][0]
    if op != 0:
        return assert_parity(got, ref, op, what)
    exact, abs_sum, lens = arith_exact(m, x)
    keep = None
    if MASKS[mask_name] == O.WRITETOZERO:
        keep = np.asarray(mask) == 0
    elif MASKS[mask_name] == O.WRITETOONE:
        keep = np.asarray(mask) != 0
    assert_arith_parity(got, ref, exact, abs_sum, lens, what, keep)


def _ref_spmv(m, sem, mask_name, x, mask):
    op, zero = SEMIRINGS[sem]
    om = to_oracle(m)
    if MASKS[mask_name] == O.NOMASK:
        return O.spmv(om, x, op, zero)
    return O.spmv(om, x, op, zero, mask, MASKS[mask_name])


@pytest.mark.parametrize("skip_empty_rows", [False, True])
@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", ["Arithmetic", "Logical", "Tropical"])
def test_dense_32(gpu, sem, mask_name, skip_empty_rows):
    m = spmv_prepare("dense_32")
    x, mask = rand01(m.num_cols, 1), rand01(m.num_rows, 2)
    got = _run_spmv(gpu, m, sem, mask_name, x, mask, skip_empty_rows)
    assert_parity(got, _ref_spmv(m, sem, mask_name, x, mask), SEMIRINGS[sem][0], "dense_32/%s/%s" % (sem, mask_name))


@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", ["Arithmetic", "Logical", "Tropical", "TropicalFloatInf"])
@pytest.mark.parametrize("name", ["uniform_10K_10", "rmat_20K", "rmat_sym_50K"])
def test_matrices(gpu, name, sem, mask_name):
    m = spmv_prepare(name)
    if sem.startswith("Tropical"):
        # distinct, non-trivial weights so (min,+) is not degenerate
        m.adj_data = (np.random.default_rng(5).integers(1, 9, size=m.nnz)).astype(np.float32)
        x = np.where(rand01(m.num_cols, 3) > 0, np.float32(SEMIRINGS[sem][1]),
                     np.random.default_rng(4).integers(0, 50, size=m.num_cols).astype(np.float32)).astype(np.float32)
    else:
        x = rand01(m.num_cols, 3)
    mask = rand01(m.num_rows, 4)
    got = _run_spmv(gpu, m, sem, mask_name, x, mask)
    _check(got, m, sem, mask_name, x, mask, "%s/%s/%s" % (name, sem, mask_name))


def test_float_values_random(gpu):
    """Arbitrary float weights and a dense random x: the float (+,x) tolerance case."""
    m = spmv_prepare("rmat_sym_50K")
    rng = np.random.default_rng(9)
    m.adj_data = rng.random(m.nnz, dtype=np.float32)
    x = rng.random(m.num_cols, dtype=np.float32)
    got = _run_spmv(gpu, m, "Arithmetic", "NoMask", x, rand01(m.num_rows, 1))
    _check(got, m, "Arithmetic", "NoMask", x, None, "random floats")


@pytest.mark.parametrize("blocks,segments", [(1, 1), (3, 1), (2, 5), (7, 3), (64, 4), (300, 1)])
def test_decompositions(gpu, blocks, segments, monkeypatch):
    """The planner's (row blocks x column segments) choice only changes the work decomposition -- and,
    for segments > 1, switches the epilogue from direct stores to init + atomic folds -- never results."""
    set_knob(monkeypatch, "spmv_blocks", str(blocks))
    set_knob(monkeypatch, "spmv_segments", str(segments))
    m = spmv_prepare("rmat_20K")
    x, mask = rand01(m.num_cols, 7), rand01(m.num_rows, 8)
    for sem in ("Logical", "Tropical", "Arithmetic"):
        for mk in ("WriteToZero", "NoMask"):
            got = _run_spmv(gpu, m, sem, mk, x, mask)
            _check(got, m, sem, mk, x, mask, "shape %dx%d %s %s" % (blocks, segments, sem, mk))


@pytest.mark.parametrize("hot,mix", [("0", "-1"), ("1024", "3"), ("1", "1"), ("1", "2"), ("4096", "4")])
def test_hot_column_cache_variants(gpu, hot, mix, monkeypatch):
    """The LDS-cached hot columns and the cold/hot interleave are pure work re-arrangements: any table
    size (0 = disabled) and any interleave must reproduce the same results, with and without segments."""
    set_knob(monkeypatch, "spmv_hot", hot)
    set_knob(monkeypatch, "spmv_mix", mix)
    m = spmv_prepare("rmat_sym_50K")
    x, mask = rand01(m.num_cols, 11), rand01(m.num_rows, 12)
    for shape in ((0, 0), (5, 3)):
        set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
        set_knob(monkeypatch, "spmv_segments", str(shape[1]))
        for sem in ("Arithmetic", "Logical", "Tropical"):
            got = _run_spmv(gpu, m, sem, "WriteToOne", x, mask)
            _check(got, m, sem, "WriteToOne", x, mask, "hot %s mix %s shape %s %s" % (hot, mix, shape, sem))


@pytest.mark.parametrize("kind", ["general", "pattern"])
@pytest.mark.parametrize("hot", ["64", "1024", "1"])
def test_run_coded_hot_stream(gpu, kind, hot, monkeypatch):
    """The hot stream is run-coded (csrc/gl_spmv_plan.h): 16-bit row slots, the column as one bit per entry + a base per group,
    table slots numbered per unit.  Runs longer than a group, runs cut by a column segment's boundary, columns absent from a
    unit and a table of one wavefront's width all give the oracle's results; general (values travel) and pattern layout."""
    set_knob(monkeypatch, "spmv_hot", hot)
    m = spmv_prepare("rmat_sym_50K")
    rng = np.random.default_rng(31)
    m.adj_data = rng.random(m.nnz, dtype=np.float32) if kind == "general" else np.full(m.nnz, np.float32(0.5), np.float32)
    x, mask = rng.random(m.num_cols, dtype=np.float32), rand01(m.num_rows, 4)
    for shape in ((0, 0), (3, 4), (40, 1)):
        set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
        set_knob(monkeypatch, "spmv_segments", str(shape[1]))
        for sem in ("Arithmetic", "Tropical"):
            got = _run_spmv(gpu, m, sem, "WriteToZero", x, mask)
            _check(got, m, sem, "WriteToZero", x, mask, "run-coded hot %s table %s shape %s %s" % (kind, hot, shape, sem))


@pytest.mark.parametrize("kind", ["general", "pattern"])
def test_delta_coded_cold_stream_bridges_wide_gaps(gpu, kind, monkeypatch):
    """The cold stream is delta-coded (csrc/gl_spmv_plan.h): a 16-bit row slot + an 8-bit delta to the previous entry's gather
    index, the index from a prefix sum over the lanes; a gap of more than 255 columns is bridged by dummy entries.  A WIDE matrix
    (20 K rows x 300 K columns, 4 entries per row, 64 row blocks: ~1250 entries per block spread over 300 K columns, mean gap
    240) makes most gaps need one or more dummies; the host and the device formatter must agree on every byte of it, and both
    layouts give the oracle's results for the three semirings."""
    from graphlily_amd import capi
    rng = np.random.default_rng(77)
    rows, cols, deg = 20096, 300032, 4
    indices = np.sort(rng.integers(0, cols, size=(rows, deg)), axis=1).astype(np.uint32).reshape(-1)
    indptr = np.arange(0, rows * deg + 1, deg, dtype=np.uint32)
    data = rng.random(rows * deg, dtype=np.float32) if kind == "general" else np.full(rows * deg, np.float32(0.25), np.float32)
    m = io.CSRMatrix(rows, cols, data, indices, indptr)
    set_knob(monkeypatch, "spmv_blocks", "64")
    x, mask = rng.random(cols, dtype=np.float32), rand01(rows, 3)
    plans = [capi.SpMVPlan(rows, cols, indptr, indices, data, flags=f) for f in (capi.GL_PLAN_HOST_FORMAT, capi.GL_PLAN_DEVICE_FORMAT)]
    assert plans[0].info()["layout"] == kind and plans[0].info()["groups"] == plans[1].info()["groups"]
    for name in ("entries", "bases", "units", "hot", "hot_hdr", "present"):
        assert np.array_equal(plans[0].export(name), plans[1].export(name)), name
    # the stream holds visibly more slots than entries: the dummies (and the padding of 64 units)
    assert plans[0].info()["groups"] * 64 > 1.3 * rows * deg
    om = to_oracle(m)
    for sem, op, zero in (("Arithmetic", 0, 0.0), ("Tropical", 2, 255.0), ("Logical", 1, 0.0)):
        xs = x if op != 2 else np.where(x > 0.5, x, np.float32(zero)).astype(np.float32)
        ref = O.spmv(om, xs, op, zero, mask, MASKS["WriteToZero"])
        for p in plans:
            dx, dm, dy = capi.DeviceBuffer.from_host(xs), capi.DeviceBuffer.from_host(mask), capi.DeviceBuffer(4 * rows)
            p.run(dx, dm, dy, op, zero, MASKS["WriteToZero"])
            got = dy.read(np.float32, rows)
            if op == 0:
                assert np.allclose(got, ref, rtol=1e-5, atol=1e-7)
            else:
                assert np.array_equal(got, ref), sem


@pytest.mark.parametrize("shape", [(0, 0), (5, 3)])
def test_row_packed_hot_stream_record_boundaries(gpu, monkeypatch, shape):
    """Pattern layout, round 6: a row's hot entries travel as records of 7 table slots + the row slot (csrc/gl_spmv_plan.h).  Rows
    with 0 ... 16 hot entries (0, 1, 7, 8, 14, 15 among them: empty, padded, exactly full, one over) next to three cold ones each;
    unsplit and split into column segments (records cut by position); the host and the device formatter agree on every byte, and
    every semiring -- (min,+) with both zeros: the padding fields name the table's identity slot, +inf there -- gives the oracle's
    results."""
    from graphlily_amd import capi
    rng = np.random.default_rng(61)
    n, hotc = 4096, 64
    rows, cols = [], []
    for r in range(n):
        k = r % 17
        hot = (np.arange(k) * 5 + r) % hotc                      # k distinct hot columns (5 is coprime to 64)
        cold = hotc + rng.choice(n - hotc, size=3, replace=False)
        c = np.unique(np.concatenate([hot, cold]))
        rows.append(np.full(c.shape[0], r))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    indptr = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    data = np.full(rows.shape[0], np.float32(0.5), np.float32)
    m = io.CSRMatrix(n, n, data, cols.astype(np.uint32), indptr)
    set_knob(monkeypatch, "spmv_hot", hotc)
    set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
    set_knob(monkeypatch, "spmv_segments", str(shape[1]))
    plans = [capi.SpMVPlan(n, n, indptr, m.adj_indices, data, flags=f) for f in (capi.GL_PLAN_HOST_FORMAT, capi.GL_PLAN_DEVICE_FORMAT)]
    assert plans[0].info()["layout"] == "pattern" and plans[0].info()["hot_columns"] == hotc
    assert plans[0].info()["hot_nnz"] == int(sum(r % 17 for r in range(n)))
    for name in ("entries", "bases", "units", "hub_rows", "hot", "hot_hdr", "present"):
        assert np.array_equal(plans[0].export(name), plans[1].export(name)), name
    x, mask = rng.random(n, dtype=np.float32), rand01(n, 9)
    om = to_oracle(m)
    for sem, op, zero in (("Arithmetic", 0, 0.0), ("Tropical", 2, 255.0), ("TropicalFloatInf", 2, 999999999.0), ("Logical", 1, 0.0)):
        xs = x if op != 2 else np.where(x > 0.5, x, np.float32(zero)).astype(np.float32)
        ref = O.spmv(om, xs, op, zero, mask, MASKS["WriteToOne"])
        for p in plans:
            dx, dm, dy = capi.DeviceBuffer.from_host(xs), capi.DeviceBuffer.from_host(mask), capi.DeviceBuffer(4 * n)
            p.run(dx, dm, dy, op, zero, MASKS["WriteToOne"])
            got = dy.read(np.float32, n)
            if op == 0:
                assert np.allclose(got, ref, rtol=1e-5, atol=1e-7), sem
            else:
                assert np.array_equal(got, ref), sem


@pytest.mark.parametrize("kind", ["general", "pattern"])
def test_run_coded_hot_stream_dense_columns(gpu, monkeypatch, kind):
    """dense_1K: every column is as hot as every other and every run is as long as the block is tall (runs spanning many
    groups and elements); with hub rows off and on (a dense row is a hub row of its block).  Pattern layout (round 6: the
    ROW-PACKED hot stream): every row holds 1024 hot entries = 147 records, far more than a lane's chunk -- the same row sits in
    several lanes of a step (hub rows on: spread over the private slots by lane; off: same-address accumulates)."""
    m = spmv_prepare("dense_1K")
    rng = np.random.default_rng(32)
    m.adj_data = rng.random(m.nnz, dtype=np.float32) if kind == "general" else np.full(m.nnz, np.float32(0.375), np.float32)
    x, mask = rng.random(m.num_cols, dtype=np.float32), rand01(m.num_rows, 5)
    for hub_div in ("48", "1000000"):
        set_knob(monkeypatch, "spmv_hub_div", hub_div)
        for sem in ("Arithmetic", "Tropical", "Logical"):
            got = _run_spmv(gpu, m, sem, "WriteToOne", x, mask)
            _check(got, m, sem, "WriteToOne", x, mask, "dense columns hub_div %s %s" % (hub_div, sem))


@pytest.mark.parametrize("kind", ["general", "pattern"])
def test_hub_row_spreading(gpu, kind):
    """A few rows holding most of a block's entries go through the 16 private LDS slots (pattern layout: their hot entries as
    records of the row-packed stream, private slot by lane)."""
    rng = np.random.default_rng(5)
    n = 4096
    dense_rows = [7, 1000, 4095]
    rows, cols = [], []
    for r in range(n):
        k = n // 2 if r in dense_rows else 3
        rows.append(np.full(k, r))
        cols.append(np.sort(rng.choice(n, size=k, replace=False)))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    indptr = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    data = rng.random(rows.shape[0], dtype=np.float32) if kind == "general" else np.full(rows.shape[0], np.float32(0.25), np.float32)
    m = io.CSRMatrix(n, n, data, cols.astype(np.uint32), indptr)
    x, mask = rng.random(n, dtype=np.float32), rand01(n, 1)
    for sem in ("Arithmetic", "Logical", "Tropical"):
        got = _run_spmv(gpu, m, sem, "NoMask", x, mask)
        _check(got, m, sem, "NoMask", x, None, "hub rows %s %s" % (kind, sem))


def test_wide_column_jumps(gpu):
    """Columns further apart than the 18-bit in-group offset force early group cuts and padding."""
    n_cols = 3_000_000
    rng = np.random.default_rng(3)
    rows = 300
    cols = np.sort(rng.choice(n_cols, size=(rows, 5), replace=False), axis=1).astype(np.uint32)
    m = io.CSRMatrix(rows, n_cols, rng.random(rows * 5, dtype=np.float32), cols.reshape(-1),
                     np.arange(0, rows * 5 + 1, 5, dtype=np.uint32))
    x = rng.random(n_cols, dtype=np.float32)
    for sem in ("Arithmetic", "Tropical"):
        got = _run_spmv(gpu, m, sem, "NoMask", x, np.zeros(rows, np.float32))
        _check(got, m, sem, "NoMask", x, None, "wide jumps " + sem)


def test_tall_matrix_many_blocks(gpu, monkeypatch):
    """More rows than 256 full-height blocks can hold is not needed to hit the row cap: force it."""
    m = spmv_prepare("uniform_10K_10")
    x, mask = rand01(m.num_cols, 1), rand01(m.num_rows, 2)
    set_knob(monkeypatch, "spmv_blocks", 1)      # one planned block, but 10112 rows fit; fine
    got = _run_spmv(gpu, m, "Arithmetic", "WriteToOne", x, mask)
    _check(got, m, "Arithmetic", "WriteToOne", x, mask, "single block")
    big = datasets.uniform(40000, 3, seed=2)        # 40000 rows > 16383: the row cap must split blocks
    got = _run_spmv(gpu, big, "Tropical", "NoMask", rand01(big.num_cols, 5), rand01(big.num_rows, 6))
    set_knob(monkeypatch, "spmv_blocks", None)
    assert_parity(got, _ref_spmv(big, "Tropical", "NoMask", rand01(big.num_cols, 5), None), 2, "row cap")


def test_edge_shapes(gpu):
    # all rows empty
    m = io.CSRMatrix(256, 64, [], [], np.zeros(257, np.uint32))
    x, mask = rand01(64, 1), rand01(256, 2)
    for sem in ("Arithmetic", "Tropical"):
        for mk in MASKS:
            got = _run_spmv(gpu, m, sem, mk, x, mask)
            assert_parity(got, _ref_spmv(m, sem, mk, x, mask), 1, "empty %s %s" % (sem, mk))
    # one row holds everything (a single very long row), ragged size
    n = 100003
    m = io.CSRMatrix(3, n, np.ones(n, np.float32), np.arange(n, dtype=np.uint32), [0, 0, n, n])
    x = rand01(n, 3)
    got = _run_spmv(gpu, m, "Arithmetic", "NoMask", x, np.zeros(3, np.float32))
    _check(got, m, "Arithmetic", "NoMask", x, None, "one long row")
    got = _run_spmv(gpu, m, "Logical", "WriteToOne", x, np.array([1, 1, 0], np.float32))
    assert_parity(got, _ref_spmv(m, "Logical", "WriteToOne", x, np.array([1, 1, 0], np.float32)), 1, "one long row logical")


def test_boolean_plan_phases_and_odd_values(gpu, monkeypatch):
    """(||,&&) runs on the pattern-only layout (GL_PLAN_BOOLEAN): x packed to bits, 1 179 648-column phases
    kept in LDS.  Three phases, explicit zeros / -0.0 / NaN / negative weights (a && b on floats), an x
    with the same oddities, every mask, split and unsplit blocks, and a non-zero semiring `zero`."""
    from graphlily_amd import capi
    rng = np.random.default_rng(21)
    n_rows, n_cols, per_row = 20000, 2_600_000, 24
    cols = np.sort(rng.integers(0, n_cols, size=(n_rows, per_row)), axis=1).astype(np.uint32)
    cols[:, -1] = n_cols - 1 - (np.arange(n_rows) % 7)          # touch the last phase's tail
    vals = rng.choice(np.array([1.0, 0.0, -0.0, -3.5, np.nan, 2.0], np.float32), size=n_rows * per_row)
    m = io.CSRMatrix(n_rows, n_cols, vals, cols.reshape(-1), np.arange(0, n_rows * per_row + 1, per_row, dtype=np.uint32))
    x = rng.choice(np.array([0.0, 0.0, 0.0, 1.0, -0.0, np.nan, -2.0], np.float32), size=n_cols)
    mask = rand01(n_rows, 4)
    for shape in ((0, 0), (3, 4)):
        set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
        set_knob(monkeypatch, "spmv_segments", str(shape[1]))
        for mk in MASKS:
            got = _run_spmv(gpu, m, "Logical", mk, x, mask)
            assert_parity(got, _ref_spmv(m, "Logical", mk, x, mask), 1, "boolean plan %s %s" % (shape, mk))
    set_knob(monkeypatch, "spmv_blocks", None)
    set_knob(monkeypatch, "spmv_segments", None)
    # the same through the C ABI: layout is boolean, zero = 1 turns every allowed row on, (+,x) is refused
    plan = capi.SpMVPlan(n_rows, n_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=capi.GL_PLAN_BOOLEAN)
    dx, dm, dy = capi.DeviceBuffer(4 * n_cols), capi.DeviceBuffer(4 * n_rows), capi.DeviceBuffer(4 * n_rows)
    dx.write(x)
    dm.write(mask)
    plan.run(dx, dm, dy, 1, 1.0, O.WRITETOZERO)
    got = dy.read(np.float32, n_rows)
    assert_parity(got, O.spmv(to_oracle(m), x, 1, 1.0, mask, O.WRITETOZERO), 1, "boolean plan, zero = 1")
    with pytest.raises(capi.GraphLilyError) as ei:
        plan.run(dx, dm, dy, 0, 0.0, O.NOMASK)
    assert ei.value.code == capi.GL_ERR_UNSUPPORTED
    # and the general layout gives the same answer for the same semiring
    general = capi.SpMVPlan(n_rows, n_cols, m.adj_indptr, m.adj_indices, m.adj_data)
    dy2 = capi.DeviceBuffer(4 * n_rows)
    plan.run(dx, dm, dy, 1, 0.0, O.WRITETOONE)
    general.run(dx, dm, dy2, 1, 0.0, O.WRITETOONE)
    assert np.array_equal(dy.read(np.float32, n_rows), dy2.read(np.float32, n_rows))


@pytest.mark.parametrize("kind", ["uniform", "per_column", "per_column_with_zeros", "sssp_self_edges", "diagonal_only_columns"])
def test_pattern_plan_matches_general_layout(gpu, kind, monkeypatch):
    """Matrices whose values are equal within every column are kept as 4-byte pattern entries and
    z = colval (x) x is formed once per run.  The products are the same floats as in the general layout
    (GL_PLAN_KEEP_VALUES), so (min,+) and (||,&&) must agree bit for bit and (+,x) within the usual
    accumulation-order tolerance; all of them must match the oracle."""
    from graphlily_amd import capi
    m = spmv_prepare("rmat_sym_50K")
    rng = np.random.default_rng(17)
    if kind == "uniform":
        m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), np.float32)
    elif kind == "sssp_self_edges":
        # unit weights + weight-0 self edges (app/sssp.h:16-62): column-constant apart from the diagonal, which the
        # plan keeps per row and folds in after the sweep
        io.sssp_add_self_edges(m)
        assert (m.adj_data == 0).any() and (m.adj_data == 1).any()
    elif kind == "diagonal_only_columns":
        # an identity block (columns whose only entry is the diagonal) next to constant columns
        import scipy.sparse as sp
        n = m.num_rows
        a = sp.csr_matrix((np.full(m.nnz, np.float32(0.25)), m.adj_indices[:m.nnz].astype(np.int64), m.adj_indptr.astype(np.int64)), shape=(n, m.num_cols)).tolil()
        a.setdiag(np.float32(3.0))
        a = a.tocsr()
        a.sort_indices()
        m = io.CSRMatrix(n, m.num_cols, a.data.astype(np.float32), a.indices.astype(np.uint32), a.indptr.astype(np.uint32))
    else:
        colval = (rng.integers(1, 40, size=m.num_cols) / np.float32(7)).astype(np.float32)
        if kind == "per_column_with_zeros":
            colval[rng.integers(0, m.num_cols, size=m.num_cols // 10)] = 0.0
        m.adj_data = colval[m.adj_indices[:m.nnz]]
    x = np.where(rand01(m.num_cols, 2) > 0, rng.random(m.num_cols, dtype=np.float32) + np.float32(0.5), 0).astype(np.float32)
    mask = rand01(m.num_rows, 3)
    dx, dm = capi.DeviceBuffer(4 * m.num_cols), capi.DeviceBuffer(4 * m.num_rows)
    dx.write(x)
    dm.write(mask)
    for shape in ((0, 0), (6, 3)):
        set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
        set_knob(monkeypatch, "spmv_segments", str(shape[1]))
        pat = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data)
        gen = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=capi.GL_PLAN_KEEP_VALUES)
        assert pat.info()["layout"] == "pattern" and gen.info()["layout"] == "general"
        assert pat.info()["device_bytes"] < 0.8 * gen.info()["device_bytes"]
        for sem in ("Arithmetic", "Logical", "Tropical"):
            op, zero = SEMIRINGS[sem]
            for mk in MASKS:
                a, b = capi.DeviceBuffer(4 * m.num_rows), capi.DeviceBuffer(4 * m.num_rows)
                pat.run(dx, dm, a, op, zero, MASKS[mk])
                gen.run(dx, dm, b, op, zero, MASKS[mk])
                ya, yb = a.read(np.float32, m.num_rows), b.read(np.float32, m.num_rows)
                what = "pattern %s %s %s %s" % (kind, shape, sem, mk)
                if op != 0:
                    assert np.array_equal(ya, yb), what
                else:
                    np.testing.assert_allclose(ya, yb, rtol=2e-6, atol=0, err_msg=what)
                _check(ya, m, sem, mk, x, mask, what)
    # a single differing entry inside one column turns the detection off
    m.adj_data = m.adj_data.copy()
    m.adj_data[m.nnz // 2] += np.float32(1.0)
    assert capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data).info()["layout"] == "general"


def test_boolean_flag_on_a_very_wide_matrix_uses_the_general_layout(gpu):
    """More than 8 x 1 179 648 columns: the bit layout would copy too many x phases per workgroup, so a
    GL_PLAN_BOOLEAN request is served by the general layout (4-byte tiles); results unchanged, no bit vector."""
    from graphlily_amd import capi
    rng = np.random.default_rng(4)
    n_rows, n_cols, per_row = 3000, 10_000_000, 12
    cols = np.sort(rng.integers(0, n_cols, size=(n_rows, per_row)), axis=1).astype(np.uint32)
    m = io.CSRMatrix(n_rows, n_cols, np.ones(n_rows * per_row, np.float32), cols.reshape(-1),
                     np.arange(0, n_rows * per_row + 1, per_row, dtype=np.uint32))
    x = (rng.random(n_cols) < 0.2).astype(np.float32)
    mask = rand01(n_rows, 2)
    plan = capi.SpMVPlan(n_rows, n_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=capi.GL_PLAN_BOOLEAN)
    assert plan.info()["layout"] != "boolean" and plan.bits_words() == 0
    got = _run_spmv(gpu, m, "Logical", "WriteToOne", x, mask)
    assert_parity(got, _ref_spmv(m, "Logical", "WriteToOne", x, mask), 1, "very wide boolean request")


def test_row_shards_compose(gpu):
    """Two row shards write disjoint slices of one y: the multi-GPU decomposition on one device."""
    m = spmv_prepare("rmat_20K")
    x, mask = rand01(m.num_cols, 1), rand01(m.num_rows, 2)
    ref = _ref_spmv(m, "Tropical", "WriteToZero", x, mask)
    cut = 7040
    a = _run_spmv(gpu, m, "Tropical", "WriteToZero", x, mask, shard=(0, cut))
    b = _run_spmv(gpu, m, "Tropical", "WriteToZero", x, mask, shard=(cut, m.num_rows))
    assert_parity(a[:cut], ref[:cut], 2, "shard 0")
    assert_parity(b[cut:], ref[cut:], 2, "shard 1")
    assert not a[cut:].any() and not b[:cut].any()   # untouched slices keep the zero fill


def test_golden_known_answers(gpu, golden_dir):
    """The reference outputs recorded in SURVEY 8(c), reproduced by the HIP path."""
    G = json.load(open(os.path.join(golden_dir, "reference_known_answers.json")))
    S = G["survey_8c"]["semiring_mask"]
    m = io.load_csr_matrix_from_float_npz(os.path.join(golden_dir, "line_8_csr_float32.npz"))
    io.util_round_csr_matrix_dim(m, 128, 128)
    m.adj_data[:] = 1
    mask = (np.arange(128) % 2).astype(np.float32)
    x = ((3 * np.arange(128)) % 5).astype(np.float32)
    for on, sem in (("arith", "Arithmetic"), ("logical", "Logical"), ("tropical", "Tropical")):
        for mn, mk in (("nomask", "NoMask"), ("wzero", "WriteToZero"), ("wone", "WriteToOne")):
            got = _run_spmv(gpu, m, sem, mk, x, mask)
            assert got[:10].tolist() == S["spmv"][on][mn], (on, mn)
    for a in G["survey_8c"]["apps"]:
        if a["call"] != "spmv_arith_nomask":
            continue
        m = io.load_csr_matrix_from_float_npz(os.path.join(golden_dir, a["matrix"] + "_csr_float32.npz"))
        io.util_round_csr_matrix_dim(m, 128, 128)
        got = _run_spmv(gpu, m, "Arithmetic", "NoMask", (np.arange(m.num_cols) % 7).astype(np.float32),
                        np.zeros(m.num_rows, np.float32))
        assert got[:len(a["first"])].tolist() == a["first"]


@pytest.mark.parametrize("keep_values", [True, False])
def test_packed_gather_vector_changes_nothing(gpu, keep_values, monkeypatch):
    """The cold entries gather from a packed copy of x (never-gathered columns dropped, the rest in degree-class
    order, refilled per run); GRAPHLILY_DEBUG spmv_compact=0 gathers from x itself, =2 packs without classes.  The
    products and their order per row are the same, so all three must agree bit for bit for (min,+) / (||,&&) and
    to accumulation-order tolerance for (+,x), on unsplit and split plans, with half of the columns empty."""
    from graphlily_amd import capi
    m = spmv_prepare("rmat_sym_50K")
    rng = np.random.default_rng(23)
    # empty every second column: relabel the entries into the even columns only
    m.adj_indices = (m.adj_indices // 2 * 2).astype(np.uint32)
    m.adj_data = (rng.integers(1, 9, size=m.nnz) / np.float32(8)).astype(np.float32) if keep_values else \
        np.full(m.nnz, np.float32(0.5), np.float32)
    x = np.where(rand01(m.num_cols, 5) > 0, rng.random(m.num_cols, dtype=np.float32) + np.float32(0.25), 0).astype(np.float32)
    mask = rand01(m.num_rows, 6)
    dx, dm, dy = capi.DeviceBuffer(4 * m.num_cols), capi.DeviceBuffer(4 * m.num_rows), capi.DeviceBuffer(4 * m.num_rows)
    dx.write(x)
    dm.write(mask)
    flags = capi.GL_PLAN_KEEP_VALUES if keep_values else 0
    for shape in ((0, 0), (5, 3)):
        set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
        set_knob(monkeypatch, "spmv_segments", str(shape[1]))
        outs = {}
        for mode in ("1", "0", "2"):
            set_knob(monkeypatch, "spmv_compact", mode)
            plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=flags)
            assert plan.info()["layout"] == ("general" if keep_values else "pattern")
            for sem, (op, zero) in SEMIRINGS.items():
                for mk, mt in MASKS.items():
                    plan.run(dx, dm if mt else None, dy, op, zero, mt)
                    outs[(mode, sem, mk)] = dy.read(np.float32, m.num_rows).copy()
            plan.destroy()
        for sem, (op, zero) in SEMIRINGS.items():
            for mk, mt in MASKS.items():
                _check(outs[("1", sem, mk)], m, sem, mk, x, mask, "%s/%s packed" % (sem, mk))
                for mode in ("0", "2"):
                    if op == O.MULADD:
                        np.testing.assert_allclose(outs[(mode, sem, mk)], outs[("1", sem, mk)], rtol=2e-6, atol=0)
                    else:
                        assert np.array_equal(outs[(mode, sem, mk)], outs[("1", sem, mk)]), (mode, sem, mk, shape)


@pytest.mark.parametrize("keep_values", [True, False])
def test_helper_modes_agree(gpu, keep_values, monkeypatch):
    """How a plan refills its hot table / packed gather vector per run -- gathering helper kernel, one streaming pass
    over x, or (general layout, small hot table) no helper launch at all -- is chosen per plan (GRAPHLILY_DEBUG spmv_helper
    forces it).  The three produce the same vectors, so every semiring x mask must agree bit for bit between them and
    match the oracle, on unsplit and split plans."""
    from graphlily_amd import capi
    m = spmv_prepare("rmat_sym_50K")
    rng = np.random.default_rng(29)
    m.adj_data = (rng.integers(1, 9, size=m.nnz) / np.float32(8)).astype(np.float32) if keep_values else \
        np.full(m.nnz, np.float32(0.5), np.float32)
    x = np.where(rand01(m.num_cols, 5) > 0, rng.random(m.num_cols, dtype=np.float32) + np.float32(0.25), 0).astype(np.float32)
    mask = rand01(m.num_rows, 6)
    dx, dm, dy = capi.DeviceBuffer(4 * m.num_cols), capi.DeviceBuffer(4 * m.num_rows), capi.DeviceBuffer(4 * m.num_rows)
    dx.write(x)
    dm.write(mask)
    flags = capi.GL_PLAN_KEEP_VALUES if keep_values else 0
    set_knob(monkeypatch, "spmv_hot", "2048")     # a table small enough for the self-hot mode
    seen = set()
    for shape in ((0, 0), (5, 3)):
        set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
        set_knob(monkeypatch, "spmv_segments", str(shape[1]))
        outs = {}
        for mode in ("0", "1", "2"):
            set_knob(monkeypatch, "spmv_helper", mode)
            plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=flags)
            info = plan.info()
            seen.add((info["helper"], info["packed_columns"] > 0))
            for sem, (op, zero) in SEMIRINGS.items():
                for mk, mt in MASKS.items():
                    plan.run(dx, dm if mt else None, dy, op, zero, mt)
                    outs[(mode, sem, mk)] = dy.read(np.float32, m.num_rows).copy()
        for sem, (op, zero) in SEMIRINGS.items():
            for mk in MASKS:
                _check(outs[("0", sem, mk)], m, sem, mk, x, mask, "helper 0 shape %s %s %s" % (shape, sem, mk))
                for mode in ("1", "2"):
                    a, b = outs[("0", sem, mk)], outs[(mode, sem, mk)]
                    if op == 0:   # (+,x): f64 sums of the same products; the packed order may differ between modes
                        assert np.allclose(a, b, rtol=1e-6, atol=0), (mode, shape, sem, mk)
                    else:
                        assert np.array_equal(a, b), (mode, shape, sem, mk)
    assert ("gather", True) in seen and ("spread", True) in seen
    if keep_values:
        assert ("self-hot", False) in seen       # pattern plans always need their helper (it forms z)


@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", ["Arithmetic", "Logical", "Tropical", "TropicalFloatInf"])
def test_reference_order_layout_is_bit_equal_to_the_oracle(gpu, sem, mask_name):
    """GL_PLAN_REFERENCE_ORDER evaluates compute_reference_results' own loop (module/spmv_module.h:478-532) on the
    device: thread per row, CSR order, separately rounded fp32 multiply and add.  Random float weights and a dense
    random x -- the case in which the fast (+,x) layouts differ from the oracle in the last bits -- whole matrix and row
    shards: every word equal."""
    m = spmv_prepare("rmat_sym_50K")
    rng = np.random.default_rng(19)
    op, zero = SEMIRINGS[sem]
    if op == 0:
        m.adj_data = rng.random(m.nnz, dtype=np.float32)
        x = rng.random(m.num_cols, dtype=np.float32)
    elif op == 1:
        m.adj_data = rng.integers(0, 2, size=m.nnz).astype(np.float32)
        x = rand01(m.num_cols, 3)
    else:
        m.adj_data = rng.integers(1, 9, size=m.nnz).astype(np.float32) + rng.random(m.nnz, dtype=np.float32)
        x = np.where(rand01(m.num_cols, 3) > 0, np.float32(zero), rng.random(m.num_cols, dtype=np.float32) * 50).astype(np.float32)
    mask = rand01(m.num_rows, 4)
    ref = _ref_spmv(m, sem, mask_name, x, mask)
    dx, dm = M.capi.DeviceBuffer.from_host(x), M.capi.DeviceBuffer.from_host(mask)
    for r0, r1 in ((0, m.num_rows), (0, 20032), (20032, m.num_rows)):
        plan = M.capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, r0, r1,
                               flags=M.capi.GL_PLAN_REFERENCE_ORDER)
        assert plan.info()["layout"] == "reference-order" and plan.info()["nnz"] == int(m.adj_indptr[r1]) - int(m.adj_indptr[r0])
        dy = M.capi.DeviceBuffer(4 * m.num_rows)
        plan.run(dx, dm if MASKS[mask_name] else None, dy, op, zero, MASKS[mask_name])
        got = dy.read(np.float32, m.num_rows)[r0:r1]
        assert np.array_equal(got.view(np.uint32), ref[r0:r1].view(np.uint32)), (sem, mask_name, r0, r1)
    if op == 0 and mask_name == "NoMask":
        # ... while the fast layout, on the same inputs, is NOT bit-equal (f64 accumulation in another order) -- and is the
        # one that is closer to the exactly evaluated product
        fast = _run_spmv(gpu, m, sem, mask_name, x, mask)
        exact, _, _ = arith_exact(m, x)
        nz = exact != 0
        assert (fast.view(np.uint32) != ref.view(np.uint32)).sum() > 0
        err_fast = np.abs(fast[nz] - exact[nz]) / np.abs(exact[nz])
        err_ref = np.abs(ref[nz] - exact[nz]) / np.abs(exact[nz])
        assert err_fast.max() <= err_ref.max() and err_fast.max() <= 1e-6


def test_reference_order_through_the_module(gpu):
    m = spmv_prepare("uniform_10K_10")
    rng = np.random.default_rng(23)
    m.adj_data = rng.random(m.nnz, dtype=np.float32)
    x = rng.random(m.num_cols, dtype=np.float32)
    mod = M.SpMVModule(M.num_hbm_channels, 1024, 256)
    mod.set_semiring(M.ArithmeticSemiring)
    mod.set_mask_type(M.kNoMask)
    mod.set_plan_flags(M.capi.GL_PLAN_REFERENCE_ORDER)
    mod.set_up_runtime()
    mod.load_and_format_matrix(m, True)
    mod.send_matrix_host_to_device()
    mod.send_vector_host_to_device(x)
    mod.run()
    got = mod.send_results_device_to_host()
    ref = O.spmv(to_oracle(m), x, O.MULADD, 0.0)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    mod.set_semiring(M.TropicalSemiring)      # any semiring runs on this layout: no re-format
    plan = mod.plan_
    mod.run()
    assert mod.plan_ is plan
    ref = O.spmv(to_oracle(m), x, O.ADDMIN, M.FLOAT_INF)
    assert np.array_equal(mod.send_results_device_to_host().view(np.uint32), ref.view(np.uint32))
