"""GPU: plan creation on the device (csrc/gl_format.hip) produces the SAME device layout as the host
formatter -- entries, bases, unit descriptors, hub rows (and phase spans for the boolean layout; the run-coded hot stream,
its headers and the units' present lists for the other two) compared byte for
byte through gl_spmv_plan_export -- for every layout, for row shards and split plans, with hot columns and hub rows,
and the plans it makes compute the oracle's results.  The full-size case also reports the two creation times."""
import json
import os
import time

import numpy as np
import pytest

from graphlily_amd import capi, datasets, io
from oracle import oracle as O

from helpers import named_matrix, to_oracle, set_knob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOST, DEV = capi.GL_PLAN_HOST_FORMAT, capi.GL_PLAN_DEVICE_FORMAT


def _both(m, flags=0, r0=0, r1=None, data=None):
    data = m.adj_data if data is None else data
    r1 = m.num_rows if r1 is None else r1
    a = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, data, r0, r1, flags=flags | HOST)
    b = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, data, r0, r1, flags=flags | DEV)
    return a, b


def _assert_same_layout(a, b, what):
    ia, ib = a.info(), b.info()
    for k in ("nnz", "num_units", "blocks", "segments", "max_block_rows", "groups", "hot_columns", "hot_nnz", "mix", "layout",
              "device_bytes"):
        assert ia[k] == ib[k], "%s: plan.%s host %r device %r" % (what, k, ia[k], ib[k])
    names = ["entries", "bases", "units", "hub_rows"] + (["spans"] if ia["layout"] == "boolean" else ["hot", "hot_hdr", "present"])
    for name in names:
        ha, hb = a.export(name), b.export(name)
        assert ha.shape == hb.shape, "%s: %s has %d words on the host, %d on the device" % (what, name, ha.size, hb.size)
        if not np.array_equal(ha, hb):
            bad = np.nonzero(ha != hb)[0]
            raise AssertionError("%s: %s differs in %d of %d words, first at %d: host %#x device %#x" %
                                 (what, name, bad.size, ha.size, bad[0], ha[bad[0]], hb[bad[0]]))
    return ia


CASES = [
    ("uniform_10K_10", 0), ("rmat_20K", 0), ("rmat_sym_50K", 0), ("gplus_small", 0), ("dense_1K", 0),
]


@pytest.mark.parametrize("name,_", CASES)
@pytest.mark.parametrize("kind", ["general", "pattern", "pattern_diag", "boolean"])
def test_device_formatter_matches_host(gpu, name, _, kind):
    m = named_matrix(name)
    rng = np.random.default_rng(3)
    if kind == "pattern_diag":
        io.sssp_add_self_edges(m)            # unit weights + zero self edges: column-constant apart from the diagonal
    io.util_round_csr_matrix_dim(m, 128, 128)
    if kind == "general":
        data, flags = rng.random(m.nnz, dtype=np.float32), 0
    elif kind == "boolean":
        data, flags = (rng.random(m.nnz) < 0.9).astype(np.float32), capi.GL_PLAN_BOOLEAN   # zero-valued entries are dropped
    else:
        data, flags = m.adj_data, 0
    a, b = _both(m, flags, data=data)
    info = _assert_same_layout(a, b, "%s/%s" % (name, kind))
    assert info["layout"] == {"general": "general", "pattern": "pattern", "pattern_diag": "pattern", "boolean": "boolean"}[kind]
    # and the device-made plan computes the oracle's result
    om = to_oracle(m)
    om.adj_data = np.ascontiguousarray(data, np.float32)
    x = rng.integers(0, 3, size=m.num_cols).astype(np.float32)
    op, zero = (1, 0.0) if kind == "boolean" else (2, 255.0)
    if op == 2:
        x = np.where(x > 0, x, np.float32(zero)).astype(np.float32)
    dx, dy = capi.DeviceBuffer.from_host(x), capi.DeviceBuffer(4 * m.num_rows)
    b.run(dx, None, dy, op, zero, 0)
    assert np.array_equal(dy.read(np.float32, m.num_rows), O.spmv(om, x, op, zero))


@pytest.mark.parametrize("kind", ["general", "pattern", "boolean"])
def test_device_formatter_matches_host_on_shards_and_split_plans(gpu, kind, monkeypatch):
    m = named_matrix("rmat_sym_50K")
    io.util_round_csr_matrix_dim(m, 128, 128)
    rng = np.random.default_rng(4)
    data = rng.random(m.nnz, dtype=np.float32) if kind == "general" else m.adj_data
    flags = capi.GL_PLAN_BOOLEAN if kind == "boolean" else 0
    n = m.num_rows
    for r0, r1 in ((0, n // 2), (n // 2, n), (n // 4 // 64 * 64, n // 4 // 64 * 64 + 4096), (0, 0)):
        a, b = _both(m, flags, r0, r1, data=data)
        _assert_same_layout(a, b, "%s shard [%d,%d)" % (kind, r0, r1))
    set_knob(monkeypatch, "spmv_blocks", "16")
    set_knob(monkeypatch, "spmv_segments", "5")
    a, b = _both(m, flags, data=data)
    info = _assert_same_layout(a, b, kind + " split 16 x 5")
    assert info["segments"] > 1
    set_knob(monkeypatch, "spmv_compact", "0")
    set_knob(monkeypatch, "spmv_hot", "0")
    a, b = _both(m, flags, data=data)
    _assert_same_layout(a, b, kind + " split, no hot table, no packed gather vector")


def test_device_formatter_rejects_bad_columns_like_the_host(gpu):
    m = named_matrix("uniform_10K_10")
    bad = m.adj_indices.copy()
    bad[12345] = m.num_cols + 7
    for f in (HOST, DEV):
        with pytest.raises(capi.GraphLilyError) as e:
            capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, bad, m.adj_data, flags=f)
        assert e.value.code == capi.GL_ERR_INVALID_ARG


@pytest.mark.parametrize("graph", ["pokec", "orkut"])
def test_device_formatter_full_size(gpu, graph):
    """The bench's matrix (and a short-row one): identical layouts, and how long each formatter takes."""
    import torch
    m = datasets.paper_graph(graph, 1.0, device=torch.device("cuda:0"))
    m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32)
    io.util_round_csr_matrix_dim(m, 128, 8)
    rec = {"graph": graph, "n": int(m.num_rows), "nnz": int(m.nnz)}
    for kind, flags in (("general", capi.GL_PLAN_KEEP_VALUES), ("pattern", 0), ("boolean", capi.GL_PLAN_BOOLEAN)):
        ts = {}
        plans = {}
        for where, f in (("device", DEV), ("host", HOST), ("device_again", DEV)):
            t0 = time.perf_counter()
            plans[where] = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=flags | f)
            ts[where] = round(time.perf_counter() - t0, 3)
        _assert_same_layout(plans["host"], plans["device"], "%s/%s" % (graph, kind))
        rec[kind] = ts
        for p in plans.values():
            p.destroy()
    print("FORMAT_SECONDS " + json.dumps(rec))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "format_seconds.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def test_csr2csc_and_normalise_on_device_equal_host(gpu, monkeypatch):
    """gl_csr2csc / gl_csr_normalize_by_outdegree: the GPU versions (forced for a small matrix) against the host ones
    and against the oracle's restatement of io/data_loader.h:108-144 and io/data_formatter.h:36-51."""
    m = named_matrix("rmat_sym_50K")
    m.adj_data = np.random.default_rng(8).random(m.nnz, dtype=np.float32)
    ref = O.csr2csc(to_oracle(m))
    outs = {}
    for where in ("0", "1"):
        monkeypatch.setenv("GRAPHLILY_PLAN_DEVICE", where)
        c = io.csr2csc(m)
        outs[where] = c
        assert np.array_equal(c.adj_indptr, ref.adj_indptr) and np.array_equal(c.adj_indices, ref.adj_indices)
        assert np.array_equal(c.adj_data, ref.adj_data)
        n = m.copy()
        io.util_normalize_csr_matrix_by_outdegree(n)
        on = to_oracle(m.copy())     # (the oracle normalises in place and shares the arrays it is given)
        O.util_normalize_csr_matrix_by_outdegree(on)
        assert np.array_equal(n.adj_data, on.adj_data)


@pytest.mark.parametrize("shard", [None, (0, 25088), (25088, 50048)])
def test_spmspv_plan_built_on_device(gpu, monkeypatch, shard):
    """The SpMSpV plan's {row, value} stream built by the device path (forced) gives the oracle's results, whole and sharded."""
    from graphlily_amd import module as M
    m = named_matrix("rmat_sym_50K")
    io.util_round_csr_matrix_dim(m, 128, 128)
    m.adj_data = np.random.default_rng(9).integers(1, 4, size=m.nnz).astype(np.float32)
    csc = io.csr2csc(m)
    rng = np.random.default_rng(10)
    idx = np.sort(rng.choice(m.num_cols, size=700, replace=False)).astype(np.uint32)
    v = M.make_sparse_vec(idx, rng.integers(1, 5, size=idx.shape[0]).astype(np.float32))
    mask = rng.integers(0, 2, size=m.num_rows).astype(np.float32)
    monkeypatch.setenv("GRAPHLILY_PLAN_DEVICE", "1")
    for sem, op, zero in ((M.ArithmeticSemiring, 0, 0.0), (M.LogicalSemiring, 1, 0.0), (M.TropicalSemiringUfixed, 2, 255.0)):
        mod = M.SpMSpVModule(0)
        mod.set_semiring(sem)
        mod.set_mask_type(M.kMaskWriteToZero if op != 2 else M.kNoMask)
        mod.set_up_runtime()
        if shard:
            mod.set_row_shard(*shard)
        mod.load_and_format_matrix(csc)
        mod.send_matrix_host_to_device()
        mod.send_vector_host_to_device(v)
        mod.send_mask_host_to_device(mask)
        mod.run()
        got = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), m.num_rows, zero)
        ref = O.spmspv(to_oracle(csc), v, op, zero, mask, O.WRITETOZERO if op != 2 else O.NOMASK)
        lo, hi = shard if shard else (0, m.num_rows)
        assert np.array_equal(got[lo:hi], ref[lo:hi])
        assert np.all(got[:lo] == zero) and np.all(got[hi:] == zero)


def test_device_csr2csc_rejects_a_malformed_row_pointer_array(gpu, monkeypatch):
    """A row pointer array that decreases (or does not start at 0) would send the device kernels out of bounds: refused."""
    from graphlily_amd import capi
    monkeypatch.setenv("GRAPHLILY_PLAN_DEVICE", "1")
    m = named_matrix("uniform_10K_10")
    bad = m.adj_indptr.copy()
    bad[5000], bad[5001] = bad[5001], bad[5000]
    with pytest.raises(capi.GraphLilyError):
        capi.host_csr2csc(m.num_rows, m.num_cols, bad, m.adj_indices, m.adj_data)
    off = m.adj_indptr.copy()
    off[0] = 3
    with pytest.raises(capi.GraphLilyError):
        capi.host_csr2csc(m.num_rows, m.num_cols, off, m.adj_indices, m.adj_data)
    c = io.csr2csc(m)                                  # the library is fine afterwards
    assert int(c.adj_indptr[-1]) == m.nnz


@pytest.mark.parametrize("where", ["0", "1"])
def test_plan_creation_refuses_a_malformed_matrix(gpu, where, monkeypatch):
    """gl_spmv_plan_create_ex / gl_spmspv_plan_create on a row (column) pointer array that decreases, and on an index out of
    range, with the host and with the device formatter: an error code, no crash, no plan -- and the library formats the
    well-formed matrix afterwards."""
    from graphlily_amd import capi
    monkeypatch.setenv("GRAPHLILY_PLAN_DEVICE", where)
    m = named_matrix("uniform_10K_10")
    bad_ptr = m.adj_indptr.copy()
    bad_ptr[5000], bad_ptr[5001] = bad_ptr[5001], bad_ptr[5000]
    bad_idx = m.adj_indices.copy()
    bad_idx[1234] = m.num_cols + 7
    for flags in (0, capi.GL_PLAN_KEEP_VALUES, capi.GL_PLAN_BOOLEAN):
        with pytest.raises(capi.GraphLilyError):
            capi.SpMVPlan(m.num_rows, m.num_cols, bad_ptr, m.adj_indices, m.adj_data, flags=flags)
        with pytest.raises(capi.GraphLilyError):
            capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, bad_idx, m.adj_data, flags=flags)
    c = io.csr2csc(m)
    bad_cptr = c.adj_indptr.copy()
    bad_cptr[100], bad_cptr[101] = bad_cptr[101], bad_cptr[100]
    bad_rows = c.adj_indices.copy()
    bad_rows[77] = c.num_rows
    with pytest.raises(capi.GraphLilyError):
        capi.SpMSpVPlan(c.num_rows, c.num_cols, bad_cptr, c.adj_indices, c.adj_data)
    with pytest.raises(capi.GraphLilyError):
        capi.SpMSpVPlan(c.num_rows, c.num_cols, c.adj_indptr, bad_rows, c.adj_data)
    plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data)
    assert plan.info()["nnz"] == m.nnz
    assert capi.SpMSpVPlan(c.num_rows, c.num_cols, c.adj_indptr, c.adj_indices, c.adj_data).info()["nnz"] == m.nnz


def _bool_plan(m, data, knob, monkeypatch):
    set_knob(monkeypatch, "bool_compress", knob)
    return capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, data, 0, m.num_rows, flags=capi.GL_PLAN_BOOLEAN)


@pytest.mark.parametrize("name", ["rmat_sym_50K", "gplus_small", "dense_1K", "uniform_10K_10"])
@pytest.mark.parametrize("where", [HOST, DEV])
def test_boolean_stream_delta_coding_decodes_to_the_4_byte_entries(gpu, name, where, monkeypatch):
    """csrc/gl_spmv_bool.hip bool_plan_compress: the 3-byte stream (16-bit row slots + 8-bit column deltas, 768 bytes per group of
    256) names exactly the entries of the 4-byte stream it was made from -- padding entries keep the ghost row slot and repeat
    their predecessor's column -- and both plans compute the oracle's result."""
    m = named_matrix(name)
    io.util_round_csr_matrix_dim(m, 128, 128)
    rng = np.random.default_rng(11)
    data = (rng.random(m.nnz) < 0.9).astype(np.float32)
    set_knob(monkeypatch, "bool_compress", 0)
    raw = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, data, 0, m.num_rows, flags=capi.GL_PLAN_BOOLEAN | where)
    set_knob(monkeypatch, "bool_compress", 1)
    cod = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, data, 0, m.num_rows, flags=capi.GL_PLAN_BOOLEAN | where)
    groups = raw.info()["groups"]
    assert cod.info()["groups"] == groups
    er, ec = raw.export("entries"), cod.export("entries")
    assert er.nbytes == groups * 1024
    for nm in ("bases", "units", "hub_rows", "spans"):
        assert np.array_equal(raw.export(nm), cod.export(nm)), nm
    om = to_oracle(m)
    om.adj_data = data
    x = (rng.random(m.num_cols) < 0.05).astype(np.float32)
    dx = capi.DeviceBuffer.from_host(x)
    want = O.spmv(om, x, 1, 0.0)
    for p in (raw, cod):
        dy = capi.DeviceBuffer(4 * m.num_rows)
        p.run(dx, None, dy, 1, 0.0, 0)
        assert np.array_equal(dy.read(np.float32, m.num_rows), want)
    if ec.nbytes == er.nbytes:       # a gap of more than 255 columns somewhere: the plan keeps the 4-byte stream
        assert np.array_equal(er, ec)
        return
    assert ec.nbytes == groups * 768 and cod.info()["device_bytes"] == raw.info()["device_bytes"] - groups * 256
    g = ec.view(np.uint8).reshape(groups, 64, 12)                   # lane l: 8 bytes of row slots, 4 of deltas: entries 4 l .. 4 l + 3
    slots = g[:, :, :8].copy().view(np.uint16).reshape(groups, 256).astype(np.uint32)
    delta = g[:, :, 8:].reshape(groups, 256).astype(np.uint32) | ((slots >> 14) << 8)      # bits 8..9 of a delta ride on its row slot
    slots &= 0x3fff
    idx = np.cumsum(delta, axis=1)
    r = er.reshape(groups, 256)
    r_row, r_idx = (r >> 5) & 0x3fff, ((r >> 19) << 5) | (r & 31)
    assert np.array_equal(slots, r_row)
    real = r_row != 0x3fff
    assert np.array_equal(idx[real], r_idx[real])
    # padding: the predecessor's column again
    prev = np.concatenate([np.zeros((groups, 1), np.uint32), idx[:, :-1].astype(np.uint32)], axis=1)
    assert np.array_equal(idx[~real], prev[~real])


def test_boolean_stream_with_wide_column_gaps_keeps_4_byte_entries(gpu, monkeypatch):
    """A matrix with neighbouring entries more than 1023 columns apart cannot be delta-coded: the plan stays on the 4-byte
    stream (one plan, one format) and computes the oracle's result.  With gaps of 256..1023 columns the 10-bit decoder is needed
    (bits 8..9 of a delta on top of its row slot): a stream this small stays as it is (the Infinity Cache holds it; the planner's
    rule, csrc/gl_spmv_bool.hip bool_plan_compress) unless the knob forces the coding, and gaps below 256 are coded with 8 bits."""
    rng = np.random.default_rng(5)
    rows, per = 512, 4
    # row r holds columns stride * (4 r .. 4 r + 3): whatever rows a block takes, its column-sorted entries are `stride` apart
    for stride, knob, coded, hi_bits in ((1100, 1, False, False), (300, 1, False, False), (300, 3, True, True), (200, 1, True, False)):
        cols = stride * rows * per
        c = (np.arange(rows * per, dtype=np.uint32) * stride).astype(np.uint32)
        m = io.CSRMatrix(rows, cols, np.ones(rows * per, np.float32), c, np.arange(0, rows * per + 1, per, dtype=np.uint32))
        p = _bool_plan(m, m.adj_data, knob, monkeypatch)
        groups = p.info()["groups"]
        ent = p.export("entries")
        assert ent.nbytes == groups * (768 if coded else 1024), (stride, knob)
        if coded:
            hi = ent.view(np.uint8).reshape(groups, 64, 12)[:, :, :8].copy().view(np.uint16) >> 14
            assert bool(hi.any()) == hi_bits, (stride, knob)
        x = (rng.random(cols) < 0.3).astype(np.float32)
        dx, dy = capi.DeviceBuffer.from_host(x), capi.DeviceBuffer(4 * rows)
        p.run(dx, None, dy, 1, 0.0, 0)
        assert np.array_equal(dy.read(np.float32, rows), O.spmv(to_oracle(m), x, 1, 0.0)), (stride, knob)
