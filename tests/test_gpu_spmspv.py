"""GPU parity: SpMSpVModule through the C ABI vs the CPU oracle.
Cases follow TEST(SpMSpV, MultipleCases) (tests/test_module_spmv_spmspv.cpp:244-314): the bank-conflict
CSC, dense_1K, uniform_10K_10 and a google+ stand-in at the reference's vector sparsities, all three
semirings x all three masks.  As in the reference (:236-240) the sparse result is compared after
convert_sparse_vec_to_dense_vec; on top, this build's documented ordering (ascending, unique) is checked."""
import json
import os

import numpy as np
import pytest

from graphlily_amd import datasets, io, module as M
from oracle import oracle as O

from helpers import MASKS, SEMIRINGS, assert_parity, named_matrix, rand01, to_oracle, set_knob

pytestmark = pytest.mark.gpu


def _strided_vector(num_cols, sparsity, seed):
    """tests/test_module_spmv_spmspv.cpp:196-209: every k-th column, values (rand()%10)/10."""
    nnz = int(np.floor(num_cols * (1 - sparsity)))
    inc = num_cols // nnz
    vals = (np.random.default_rng(seed).integers(0, 10, size=nnz) / 10.0).astype(np.float32)
    return M.make_sparse_vec(np.arange(nnz, dtype=np.uint32) * inc, vals)


def _run(gpu, csc, sem, mask_name, v, mask, shard=None):
    op, zero = SEMIRINGS[sem]
    mod = M.SpMSpVModule(512)
    mod.set_semiring(M.SemiringType(op, 1.0, zero))
    mod.set_mask_type(MASKS[mask_name])
    mod.set_up_runtime()
    if shard:
        mod.set_row_shard(*shard)
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    mod.send_mask_host_to_device(mask)
    mod.send_vector_host_to_device(v)
    mod.run()
    res = mod.send_results_device_to_host()
    nnz = mod.get_results_nnz()
    assert nnz == int(res["index"][0])
    idx = res["index"][1:nnz + 1].astype(np.int64)
    assert np.all(np.diff(idx) > 0), "result indices must be ascending and unique"
    assert res["val"][0] == np.float32(zero)          # head {nnz, Zero} (kernel_spmspv_impl.h:551-555)
    assert not np.any(res["val"][1:nnz + 1] == np.float32(zero)), "entries equal to zero must not be emitted"
    return M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, zero), mod


def _case(gpu, csc, sem, mask_name, sparsity, seed=0):
    v = _strided_vector(csc.num_cols, sparsity, seed)
    mask = rand01(csc.num_rows, seed + 100)
    got, _ = _run(gpu, csc, sem, mask_name, v, mask)
    op, zero = SEMIRINGS[sem]
    ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS[mask_name])
    assert_parity(got, ref, op, "%s/%s/%.4f" % (sem, mask_name, sparsity))


def _csc(name):
    if name == "conflict1024":
        return datasets.conflict(1024)
    c = io.csr2csc(named_matrix(name))
    c.adj_data = np.full(c.nnz, np.float32(1.0 / c.num_rows), dtype=np.float32)   # :252-266
    return c


@pytest.mark.parametrize("name,sparsity", [("conflict1024", 0.0), ("dense_1K", 0.5), ("uniform_10K_10", 0.5),
                                            ("gplus_small", 0.99), ("dense_1K", 0.99)])
def test_reference_arithmetic_cases(gpu, name, sparsity):
    _case(gpu, _csc(name), "Arithmetic", "NoMask", sparsity)


@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", ["Arithmetic", "Logical", "Tropical"])
def test_dense_1K_all_semirings_masks(gpu, sem, mask_name):
    _case(gpu, _csc("dense_1K"), sem, mask_name, 0.99)


@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", ["Logical", "Tropical", "TropicalFloatInf", "Arithmetic"])
@pytest.mark.parametrize("sparsity", [0.9, 0.999])
def test_power_law(gpu, sem, mask_name, sparsity):
    """gplus stand-in: hub columns exceed the hub threshold, short columns dominate the count."""
    c = io.csr2csc(named_matrix("gplus_small"))
    if sem.startswith("Tropical"):
        c.adj_data = np.random.default_rng(2).integers(1, 5, size=c.nnz).astype(np.float32)
    op, zero = SEMIRINGS[sem]
    v = _strided_vector(c.num_cols, sparsity, 3)
    if sem.startswith("Tropical"):
        # mask values drawn from {zero, 0}: the SpMSpV mask is compared with semiring.zero
        mask = np.where(rand01(c.num_rows, 5) > 0, np.float32(zero), np.float32(0)).astype(np.float32)
    else:
        mask = rand01(c.num_rows, 5)
    got, _ = _run(gpu, c, sem, mask_name, v, mask)
    ref = O.spmspv(to_oracle(c), v, op, zero, mask, MASKS[mask_name])
    assert_parity(got, ref, op, "gplus/%s/%s" % (sem, mask_name))


def _column_constant_csc(kind, seed=4):
    """rmat_20K as a CSC whose columns are constant apart from the diagonal: `sssp` = unit weights + weight-0 self edges
    (app/sssp.h:16-62), `pagerank` = 0.9 / out-degree per column (app/pagerank.h:66-67) with the diagonal entries set to other
    values (exceptions), `unit` = all ones."""
    m = named_matrix("rmat_20K")
    if kind == "sssp":
        io.sssp_add_self_edges(m)
    io.util_round_csr_matrix_dim(m, 128, 128)
    c = io.csr2csc(m)
    if kind == "pagerank":
        deg = np.maximum(np.diff(c.adj_indptr.astype(np.int64)), 1)
        col_of = np.repeat(np.arange(c.num_cols), np.diff(c.adj_indptr.astype(np.int64)))
        c.adj_data = (np.float32(0.9) / deg[col_of].astype(np.float32)).astype(np.float32)
        diag = c.adj_indices == col_of
        c.adj_data[diag] = np.random.default_rng(seed).random(int(diag.sum()), dtype=np.float32) + 2.0   # exceptions
    elif kind == "unit":
        c.adj_data = np.ones(c.nnz, np.float32)
    return c


@pytest.mark.parametrize("kind", ["sssp", "pagerank", "unit"])
@pytest.mark.parametrize("cnt", [300, 1025, 2500, 4097, 9000, 16384, 16385, 20000])
def test_column_constant_matrices(gpu, kind, cnt):
    """The matrices the apps build -- unit weights, SSSP's zero diagonal (app/sssp.h:16-62), PageRank-like 0.9 / out-degree
    columns, here with diagonal exceptions -- through every path of the bin kernel (one window / 2, 4, 8, 16 entries per thread
    without a rendezvous, up to 16 windows / the rendezvous / the whole vector), three semirings, a mask compared with `zero`.  (Round 5 tried 4-byte bin records for such
    matrices -- 12 bytes moved per product instead of 24 -- and reverted them: the operator is bound by its phases, not its
    bytes; EXPERIMENTS R5.4.  The cases stay.)"""
    c = _column_constant_csc(kind)
    rng = np.random.default_rng(cnt)
    cols = np.sort(rng.choice(c.num_cols, size=min(cnt, c.num_cols), replace=False)).astype(np.uint32)
    for sem in ("Arithmetic", "Tropical", "Logical"):
        op, zero = SEMIRINGS[sem]
        vals = (rng.integers(1, 9, size=cols.shape[0]) / 4.0).astype(np.float32)
        if sem == "Logical":
            vals[::7] = 0.0                     # (entries whose products cannot change a result)
        v = M.make_sparse_vec(cols, vals)
        mask = np.where(rand01(c.num_rows, 5) > 0, np.float32(zero), np.float32(3)).astype(np.float32)
        got, mod = _run(gpu, c, sem, "WriteToZero", v, mask)
        ref = O.spmspv(to_oracle(c), v, op, zero, mask, MASKS["WriteToZero"])
        assert_parity(got, ref, op, "column-constant %s %s %d" % (kind, sem, cnt))


@pytest.mark.parametrize("cnt", [4097, 9000, 20000])
def test_cut_by_entries_and_cut_by_products_agree(gpu, cnt, monkeypatch):
    """A mid-size vector over a matrix without long columns is cut into equal slices of ENTRIES (no rendezvous); with long
    columns, or forced by the knob, into equal ranges of PRODUCTS behind the rendezvous.  Both against the oracle, and
    bit-identical lists for the order-free semirings."""
    c = _column_constant_csc("sssp")
    assert int(np.diff(c.adj_indptr.astype(np.int64)).max()) <= 12288      # (else the default would not take the first path)
    rng = np.random.default_rng(cnt + 1)
    cols = np.sort(rng.choice(c.num_cols, size=cnt, replace=False)).astype(np.uint32)
    vals = (rng.integers(1, 9, size=cnt) / 4.0).astype(np.float32)
    v = M.make_sparse_vec(cols, vals)
    mask = np.zeros(c.num_rows, np.float32)
    for sem in ("Arithmetic", "Tropical", "Logical"):
        op, zero = SEMIRINGS[sem]
        ref = O.spmspv(to_oracle(c), v, op, zero, mask, MASKS["NoMask"])
        lists = []
        for knob in (None, "spmspv_by_entries_maxcol=0"):
            if knob:
                monkeypatch.setenv("GRAPHLILY_DEBUG", knob)
            else:
                monkeypatch.delenv("GRAPHLILY_DEBUG", raising=False)
            got, mod = _run(gpu, c, sem, "NoMask", v, mask)
            assert_parity(got, ref, op, "%s %d %s" % (sem, cnt, knob))
            lists.append(got.copy())
        monkeypatch.delenv("GRAPHLILY_DEBUG", raising=False)
        if sem != "Arithmetic":                      # (dense forms of the two lists)
            assert np.array_equal(lists[0], lists[1])
        else:
            assert np.array_equal(lists[0] != 0, lists[1] != 0)


def test_saturating_min_plus(gpu):
    """(min,+) products saturate at FLOAT_INF (hw/float_pe.h:24-33, spmspv_module.h:482-491)."""
    c = _csc("dense_1K")
    c.adj_data[::3] = np.float32(2e9)
    v = M.make_sparse_vec([0, 5, 9], [np.float32(3e9), 1.0, np.float32(9.9e8)])
    zero = 999999999.0
    got, _ = _run(gpu, c, "TropicalFloatInf", "NoMask", v, np.zeros(c.num_rows, np.float32))
    ref = O.spmspv(to_oracle(c), v, O.ADDMIN, zero)
    assert_parity(got, ref, 2, "saturation")


def test_repeated_runs_and_empty_frontier(gpu):
    """The accumulator is restored by every run; an empty frontier gives an empty result; changing the
    semiring zero between runs re-initialises it."""
    c = io.csr2csc(named_matrix("uniform_10K_10"))
    oc = to_oracle(c)
    mod = M.SpMSpVModule(512)
    mod.set_up_runtime()
    mod.load_and_format_matrix(c)
    mod.send_matrix_host_to_device()
    mask = rand01(c.num_rows, 1)
    mod.send_mask_host_to_device(mask)
    for it, (sem, mk, sp) in enumerate([("Logical", "WriteToZero", 0.9), ("Tropical", "NoMask", 0.99),
                                        ("Arithmetic", "WriteToOne", 0.5), ("Logical", "NoMask", 0.999)]):
        op, zero = SEMIRINGS[sem]
        mod.set_semiring(M.SemiringType(op, 1.0, zero))
        mod.set_mask_type(MASKS[mk])
        v = _strided_vector(c.num_cols, sp, it)
        mod.send_vector_host_to_device(v)
        mod.run()
        got = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), c.num_rows, zero)
        assert_parity(got, O.spmspv(oc, v, op, zero, mask, MASKS[mk]), op, "run %d" % it)
    mod.send_vector_host_to_device(M.make_sparse_vec([], []))
    mod.run()
    assert mod.get_results_nnz() == 0


def test_row_shards_concatenate(gpu):
    """Row-sharded plans produce disjoint ascending slices whose concatenation is the full result."""
    c = io.csr2csc(named_matrix("rmat_20K"))
    v = _strided_vector(c.num_cols, 0.95, 1)
    mask = rand01(c.num_rows, 2)
    full, _ = _run(gpu, c, "Logical", "WriteToZero", v, mask)
    cut = 9984
    a, _ = _run(gpu, c, "Logical", "WriteToZero", v, mask, shard=(0, cut))
    b, _ = _run(gpu, c, "Logical", "WriteToZero", v, mask, shard=(cut, c.num_rows))
    assert not a[cut:].any() and not b[:cut].any()
    assert np.array_equal(a + b, full)


def test_golden_known_answers(gpu, golden_dir):
    G = json.load(open(os.path.join(golden_dir, "reference_known_answers.json")))
    S = G["survey_8c"]["semiring_mask"]
    m = io.load_csr_matrix_from_float_npz(os.path.join(golden_dir, "line_8_csr_float32.npz"))
    io.util_round_csr_matrix_dim(m, 128, 128)
    m.adj_data[:] = 1
    c = io.csr2csc(m)
    mask = (np.arange(128) % 2).astype(np.float32)
    v = M.make_sparse_vec([p[0] for p in S["spmspv_v"]], [p[1] for p in S["spmspv_v"]])
    for on, sem in (("arith", "Arithmetic"), ("logical", "Logical"), ("tropical", "Tropical")):
        for mn, mk in (("nomask", "NoMask"), ("wzero", "WriteToZero"), ("wone", "WriteToOne")):
            got, _ = _run(gpu, c, sem, mk, v, mask)
            assert got[:10].tolist() == S["spmspv"][on][mn], (on, mn)


@pytest.mark.parametrize("shape", [(0, 0), (4, 3)])
def test_direction_switch_inside_the_operator(gpu, shape, monkeypatch):
    """With a boolean SpMV plan of the same matrix attached, a (||,&&) run whose frontier columns hold more
    than 1/32 of the non-zeros goes row-wise (frontier -> bits -> boolean SpMV -> compaction) instead of being
    scattered.  Both directions must give the oracle's result for every mask; zero-valued frontier entries
    and zero-valued matrix entries take part in neither."""
    set_knob(monkeypatch, "spmv_blocks", str(shape[0]))
    set_knob(monkeypatch, "spmv_segments", str(shape[1]))
    csr = named_matrix("rmat_sym_50K")
    rng = np.random.default_rng(3)
    csr.adj_data = rng.choice(np.array([1.0, 1.0, 0.0, 2.5], np.float32), size=csr.nnz)
    csc = io.csr2csc(csr)
    spmv = M.SpMVModule(16, 0, 0)
    spmv.set_semiring(M.LogicalSemiring)
    spmv.set_up_runtime()
    spmv.load_and_format_matrix(csr, True)
    spmv.send_matrix_host_to_device()
    mask = rand01(csc.num_rows, 5)
    for density, expect in ((0.0005, "scatter"), (0.3, "row-wise")):
        idx = np.flatnonzero(rng.random(csc.num_cols) < density).astype(np.uint32)
        vals = rng.choice(np.array([1.0, 0.0, 3.0], np.float32), size=idx.size)
        v = M.make_sparse_vec(idx, vals)
        for mk in MASKS:
            mod = M.SpMSpVModule(0)
            mod.set_semiring(M.LogicalSemiring)
            mod.set_mask_type(MASKS[mk])
            mod.set_up_runtime()
            mod.load_and_format_matrix(csc)
            mod.send_matrix_host_to_device()
            mod.attach_pull(spmv)
            mod.send_mask_host_to_device(mask)
            mod.send_vector_host_to_device(v)
            for rep in range(2):   # the second run finds the accumulator reset by the first
                mod.run()
                assert mod.plan_.last_direction() == expect
                res = mod.send_results_device_to_host()
                nnz = mod.get_results_nnz()
                got = M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, 0.0)
                assert np.all(np.diff(res["index"][1:nnz + 1].astype(np.int64)) > 0)
                ref = O.spmspv(to_oracle(csc), v, 1, 0.0, mask, MASKS[mk])
                assert_parity(got, ref, 1, "direction %s %s run %d" % (expect, mk, rep))


def test_arithmetic_direction_switch(gpu):
    """(+,x) with a general / pattern SpMV plan of the same matrix attached: heavy frontiers are applied row-wise
    (frontier -> dense x -> SpMV into the accumulator), light ones scattered; both match the oracle within the
    float tolerance, for every mask, on a weighted (general layout) and a constant (pattern layout) matrix."""
    csr0 = named_matrix("rmat_sym_50K")
    rng = np.random.default_rng(8)
    for layout in ("general", "pattern"):
        csr = csr0.copy()
        csr.adj_data = (rng.random(csr.nnz, dtype=np.float32) + np.float32(0.25)) if layout == "general" else np.full(csr.nnz, np.float32(0.125), np.float32)
        csc = io.csr2csc(csr)
        mask = rand01(csc.num_rows, 6)
        for density, expect in ((0.0005, "scatter"), (0.25, "row-wise")):
            idx = np.flatnonzero(rng.random(csc.num_cols) < density).astype(np.uint32)
            vals = (rng.integers(1, 9, size=idx.size) / np.float32(8)).astype(np.float32)
            v = M.make_sparse_vec(idx, vals)
            for mk in MASKS:
                mod = M.SpMSpVModule(0)
                mod.set_semiring(M.ArithmeticSemiring)
                mod.set_mask_type(MASKS[mk])
                mod.set_up_runtime()
                mod.load_and_format_matrix(csc)
                mod.send_matrix_host_to_device()
                mod.enable_own_pull()
                assert mod.own_pull_.plan_.info()["layout"] == layout
                mod.send_mask_host_to_device(mask)
                mod.send_vector_host_to_device(v)
                for rep in range(2):
                    mod.run()
                    assert mod.plan_.last_direction() == expect
                    res = mod.send_results_device_to_host()
                    got = M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, 0.0)
                    ref = O.spmspv(to_oracle(csc), v, 0, 0.0, mask, MASKS[mk])
                    assert_parity(got, ref, 0, "arith direction %s %s %s run %d" % (layout, expect, mk, rep))


@pytest.mark.parametrize("sem", ["Tropical", "TropicalFloatInf"])
def test_tropical_direction_switch(gpu, sem):
    """(min,+) goes row-wise on heavy frontiers too (x = +inf off the frontier); bit-equal to the oracle, with
    negative weights, a diagonal that differs from its columns (pattern layout with exceptions) and every mask."""
    op, zero = SEMIRINGS[sem]
    csr = named_matrix("rmat_sym_50K")
    rng = np.random.default_rng(12)
    for layout in ("general", "pattern"):
        m = csr.copy()
        if layout == "general":
            m.adj_data = rng.integers(-2, 9, size=m.nnz).astype(np.float32)
        else:
            io.sssp_add_self_edges(m)
        csc = io.csr2csc(m)
        spmv = M.SpMVModule(16, 0, 0)
        spmv.set_semiring(M.SemiringType(op, 0.0, zero))
        spmv.set_up_runtime()
        spmv.load_and_format_matrix(m, True)
        spmv.send_matrix_host_to_device()
        assert spmv.plan_.info()["layout"] == layout
        mask = np.where(rand01(csc.num_rows, 3) > 0, np.float32(zero), np.float32(4.0)).astype(np.float32)
        for density, expect in ((0.0005, "scatter"), (0.3, "row-wise")):
            idx = np.flatnonzero(rng.random(csc.num_cols) < density).astype(np.uint32)
            v = M.make_sparse_vec(idx, rng.integers(0, 6, size=idx.size).astype(np.float32))
            for mk in MASKS:
                mod = M.SpMSpVModule(0)
                mod.set_semiring(M.SemiringType(op, 0.0, zero))
                mod.set_mask_type(MASKS[mk])
                mod.set_up_runtime()
                mod.load_and_format_matrix(csc)
                mod.send_matrix_host_to_device()
                mod.attach_pull(spmv)
                mod.send_mask_host_to_device(mask)
                mod.send_vector_host_to_device(v)
                mod.run()
                assert mod.plan_.last_direction() == expect
                got = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), csc.num_rows, zero)
                ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS[mk])
                assert_parity(got, ref, op, "tropical direction %s %s %s %s" % (sem, layout, expect, mk))


@pytest.mark.parametrize("sparsity", [0.5, 0.999])
def test_run_assign_equals_run_then_assign(gpu, sparsity):
    """gl_spmspv_run_assign = SpMSpV + AssignVectorSparse::run(val) (app/bfs.h:146-148), with the BFS aliasing:
    the mask IS the distance vector the assign writes."""
    csr = named_matrix("gplus_small")
    csc = io.csr2csc(csr)
    csc.adj_data = np.ones(csc.nnz, dtype=np.float32)
    n = csc.num_rows
    v = _strided_vector(csc.num_cols, sparsity, 3)
    dist0 = (np.random.default_rng(5).random(n) < 0.4).astype(np.float32) * 2.0   # 0 = unvisited
    outs = []
    for fused in (False, True):
        mod = M.SpMSpVModule(512)
        mod.set_semiring(M.LogicalSemiring)
        mod.set_mask_type(M.kMaskWriteToZero)
        mod.set_up_runtime()
        mod.load_and_format_matrix(csc)
        mod.send_matrix_host_to_device()
        mod.send_mask_host_to_device(dist0.copy())
        mod.send_vector_host_to_device(v)
        if fused:
            mod.run_assign(mod.mask_buf, 7.0)
        else:
            mod.run()
            asg = M.AssignVectorSparseModule(False)
            asg.set_up_runtime()
            asg.bind_mask_buf(mod.results_buf)
            asg.bind_inout_buf(mod.mask_buf)
            asg.run(7.0)
        res = mod.send_results_device_to_host()
        nnz = mod.get_results_nnz()
        outs.append((res[:nnz + 1].copy(), mod.send_mask_device_to_host()[:n].copy()))
    (r0, d0), (r1, d1) = outs
    assert r0.tobytes() == r1.tobytes()
    assert np.array_equal(d0, d1)
    ref = O.spmspv(to_oracle(csc), v, 1, 0.0, dist0, M.kMaskWriteToZero)
    want = dist0.copy()
    want[ref != 0] = 7.0
    assert np.array_equal(d1, want)
    assert int(r1["index"][0]) == int((ref != 0).sum())


@pytest.mark.parametrize("sem", ["Arithmetic", "Logical", "Tropical"])
def test_many_long_columns_fill_the_chunk_queue(gpu, sem):
    """Columns just above a multiple of the 4096-entry chunk size need ceil(deg / 4096) queue slots each; nnz / 4096
    undercounts that (two columns of 4097 entries: 3 slots for 4 chunks -- round 1 dropped the surplus silently).
    24 columns of 4097..8191 entries, all in the frontier; then the same frontier with every column listed twice,
    which exceeds even the exact capacity and must take the in-kernel overflow path."""
    rng = np.random.default_rng(17)
    n = 16384
    degs = rng.integers(4097, 8192, size=24)
    degs[:4] = 4097
    cols = np.sort(rng.choice(n, size=24, replace=False))
    indptr = np.zeros(n + 1, np.int64)
    indptr[cols + 1] = degs
    indptr = np.cumsum(indptr)
    rows = np.concatenate([np.sort(rng.choice(n, size=int(d), replace=False)) for d in degs]).astype(np.uint32)
    vals = rng.integers(1, 4, size=rows.shape[0]).astype(np.float32)
    csc = io.CSCMatrix(n, n, vals, rows, indptr.astype(np.uint32))
    op, zero = SEMIRINGS[sem]
    mask = rand01(n, 9)
    xv = rng.integers(1, 5, size=24).astype(np.float32)
    v = M.make_sparse_vec(cols, xv)
    got, _ = _run(gpu, csc, sem, "WriteToZero" if sem != "Tropical" else "NoMask", v, mask)
    ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS["WriteToZero" if sem != "Tropical" else "NoMask"])
    assert_parity(got, ref, op, "long columns " + sem)
    assert (ref != zero).sum() > n // 4
    # duplicates: the reference loop simply applies a column twice (spmspv_module.h:463-497)
    v2 = M.make_sparse_vec(np.concatenate([cols, cols]), np.concatenate([xv, xv + 1]))
    mod = M.SpMSpVModule(512)
    mod.set_semiring(M.SemiringType(op, 1.0, zero))
    mod.set_mask_type(M.kNoMask)
    mod.set_up_runtime()
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    mod.send_mask_host_to_device(mask)
    mod.send_vector_host_to_device(v2)
    mod.run()
    got2 = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), n, zero)
    ref2 = O.spmspv(to_oracle(csc), v2, op, zero, mask, O.NOMASK)
    assert_parity(got2, ref2, op, "long columns listed twice " + sem)
    # and a clean run afterwards: the queue counter was reset although it overflowed
    mod.send_vector_host_to_device(v)
    mod.run()
    got3 = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), n, zero)
    assert_parity(got3, O.spmspv(to_oracle(csc), v, op, zero, mask, O.NOMASK), op, "run after overflow " + sem)


@pytest.mark.parametrize("sem", list(SEMIRINGS))
@pytest.mark.parametrize("mask_name", list(MASKS))
def test_tiny_runs_are_one_launch_with_the_same_result(gpu, sem, mask_name, monkeypatch):
    """gl_spmspv_plan_hint_work: a module that has just uploaded a vector of <= 1024 entries whose columns hold <= 2048
    non-zeros runs it as ONE launch (scatter with first-touch detection, sort of the rows reached, ordered emission).  The
    result list must be the general path's, entry for entry ((+,x): same rows, values within the float tolerance -- atomics
    add in arrival order either way), and the oracle's; repeated runs keep the hint, another module's write drops it."""
    from graphlily_amd import capi
    csc = _csc("rmat_20K")
    op, zero = SEMIRINGS[sem]
    coldeg = np.diff(csc.adj_indptr.astype(np.int64))
    rng = np.random.default_rng(3)
    order = np.argsort(coldeg)
    light = order[coldeg[order] > 0][:4000]                      # short columns: many entries stay tiny
    picks = {"one": light[:1], "ten": light[5:15], "many": np.sort(rng.choice(light, 900, replace=False)),
             "hub": np.array([order[-1]]),                         # one long column: > 2048 non-zeros -> general path
             "dups": np.array([light[3], light[3], light[7]])}     # a column named twice
    mid, tot = [], 0                                              # 6000 .. 8000 products in few columns: the two-launch path, hinted
    for c in order[::-1][5:]:
        if tot + coldeg[c] <= 8000:
            mid.append(c)
            tot += int(coldeg[c])
        if tot > 6000 or len(mid) == 1000:
            break
    picks["mid"] = np.sort(np.array(mid))
    assert 6000 < tot <= 8000
    mask = rand01(csc.num_rows, 11)
    for name, cols in picks.items():
        vals = (rng.integers(1, 10, size=len(cols)) / 10.0).astype(np.float32)
        v = M.make_sparse_vec(cols.astype(np.uint32), vals)
        out = {}
        for tiny in ("1", "0"):
            set_knob(monkeypatch, "spmspv_tiny", tiny)
            got, mod = _run(gpu, csc, sem, mask_name, v, mask)
            res1 = mod.send_results_device_to_host()
            mod.run()                                             # the hint survives the module's own runs
            res2 = mod.send_results_device_to_host()
            n1 = int(res1["index"][0])
            assert n1 == int(res2["index"][0]) and np.array_equal(res1["index"][1:n1 + 1], res2["index"][1:n1 + 1])
            out[tiny] = (got, res1, n1)
            work = int(coldeg[cols].sum())
            assert (mod.tiny_ is not None) == (len(cols) <= 1024), name
            if tiny == "1" and name == "ten":
                assert work <= 2048
        (g1, r1, n1), (g0, r0, n0) = out["1"], out["0"]
        assert n1 == n0 and np.array_equal(r1["index"][1:n1 + 1], r0["index"][1:n0 + 1]), "%s: rows differ" % name
        if op == 0:
            assert np.allclose(r1["val"][1:n1 + 1], r0["val"][1:n0 + 1], rtol=1e-5, atol=0)
        else:
            assert np.array_equal(r1["val"][1:n1 + 1], r0["val"][1:n0 + 1]), "%s: values differ" % name
        ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS[mask_name])
        assert_parity(g1, ref, op, "tiny %s/%s/%s" % (sem, mask_name, name))
    # a stale hint: the plan is told "tiny", the vector is not -- the one workgroup still computes it (slowly)
    set_knob(monkeypatch, "spmspv_tiny", "1")
    cols = np.sort(rng.choice(order[coldeg[order] > 0], 3000, replace=False))
    v = M.make_sparse_vec(cols.astype(np.uint32), np.full(len(cols), 0.5, np.float32))
    got, mod = _run(gpu, csc, sem, mask_name, v, mask)            # general path (3000 entries: no hint from the module)
    ref_res = mod.send_results_device_to_host()
    mod.plan_.hint_work(10, 100, 50)
    mod.run()
    res = mod.send_results_device_to_host()
    n = int(res["index"][0])
    assert n == int(ref_res["index"][0]) and np.array_equal(res["index"][1:n + 1], ref_res["index"][1:n + 1])
    if op == 0:
        assert np.allclose(res["val"][1:n + 1], ref_res["val"][1:n + 1], rtol=1e-5, atol=0)
    else:
        assert np.array_equal(res["val"][1:n + 1], ref_res["val"][1:n + 1])


@pytest.mark.parametrize("sem", list(SEMIRINGS))
def test_work_hint_never_changes_the_result(gpu, sem, monkeypatch):
    """gl_spmspv_plan_hint_work: a module that uploaded the vector from the host tells the plan the non-zeros of its columns
    and the longest of them; light vectors then skip the direction switch's decision kernels and, without a long column, the
    chunk-queue pass.  With the switch attached (enable_own_pull), for light, long-column and heavy (row-wise) vectors: the
    hinted run, the unhinted run (GRAPHLILY_DEBUG spmspv_work_hint=0) and the oracle agree; and a STALE hint -- "light, no long
    column" for a vector that is heavy and names the hubs -- only costs time."""
    csc = _csc("rmat_20K")
    op, zero = SEMIRINGS[sem]
    coldeg = np.diff(csc.adj_indptr.astype(np.int64))
    order = np.argsort(coldeg)
    rng = np.random.default_rng(5)
    nz = order[coldeg[order] > 0]
    picks = {"light": np.sort(rng.choice(nz[:len(nz) // 2], 300, replace=False)),
             "hubs": np.sort(order[-6:]),                                     # long columns (>= 4096 entries on this graph?)
             "heavy": np.sort(rng.choice(nz, len(nz) // 3, replace=False))}   # far beyond non-zeros / 32: row-wise
    mask = rand01(csc.num_rows, 12)
    mods = {}
    for hint in ("1", "0"):
        set_knob(monkeypatch, "spmspv_work_hint", hint)
        mod = M.SpMSpVModule(512)
        mod.set_semiring(M.SemiringType(op, 1.0, zero))
        mod.set_mask_type(MASKS["WriteToZero"])
        mod.set_up_runtime()
        mod.load_and_format_matrix(csc)
        mod.send_matrix_host_to_device()
        mod.enable_own_pull()
        mod.send_mask_host_to_device(mask)
        mods[hint] = mod
    for name, cols in picks.items():
        vals = (rng.integers(1, 10, size=len(cols)) / 10.0).astype(np.float32)
        v = M.make_sparse_vec(cols.astype(np.uint32), vals)
        ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS["WriteToZero"])
        dirs = {}
        for hint in ("1", "0"):
            set_knob(monkeypatch, "spmspv_work_hint", hint)
            mod = mods[hint]
            mod.send_vector_host_to_device(v)
            for rep in range(2):                       # (the module repeats the hint while the vector is its own upload)
                mod.run()
                got = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), csc.num_rows, zero)
                assert_parity(got, ref, op, "work hint %s %s/%s rep %d" % (hint, sem, name, rep))
            dirs[hint] = mod.plan_.last_direction()
        assert dirs["1"] == dirs["0"], (name, dirs)
        if name == "heavy":
            assert dirs["1"] == "row-wise"
    # stale: the plan is told "300 entries, 900 non-zeros, longest column 5" and finds the heavy vector with the hubs in it
    set_knob(monkeypatch, "spmspv_work_hint", "1")
    cols = np.union1d(picks["heavy"], picks["hubs"])
    v = M.make_sparse_vec(cols.astype(np.uint32), np.full(len(cols), 0.5, np.float32))
    ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS["WriteToZero"])
    mod = mods["1"]
    mod.send_vector_host_to_device(v)
    mod.tiny_ = None                                   # (the module's own, correct hint is dropped ...)
    mod.plan_.hint_work(300, 900, 5)                    # (... for a wrong one)
    mod.run()
    got = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), csc.num_rows, zero)
    assert_parity(got, ref, op, "stale work hint " + sem)
    assert mod.plan_.last_direction() == "scatter"


def test_vector_head_larger_than_the_vector_is_clamped(gpu):
    """send_vector_host_to_device of a vector whose head count exceeds the entries it holds: the reference would read its
    zero-initialised mirror (spmspv_module.h:280); here the device block is recycled memory, so the head is clamped to what
    was uploaded and the run uses exactly those entries."""
    from graphlily_amd import capi, io, module as M
    from helpers import named_matrix, to_oracle
    m = named_matrix("uniform_10K_10")
    io.util_round_csr_matrix_dim(m, 128, 128)
    csc = io.csr2csc(m)
    # dirty the pool's blocks of this size first
    junk = capi.DeviceBuffer(8 * (csc.num_cols + 1))
    junk.write(np.full(2 * (csc.num_cols + 1), 0x7fffffff, np.uint32))
    junk.free()
    mod = M.SpMSpVModule(1024)
    mod.set_semiring(M.ArithmeticSemiring)
    mod.set_mask_type(M.kNoMask)
    mod.set_up_runtime()
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    v = M.make_sparse_vec([3, 77, 4000], [1.0, 2.0, 3.0])
    v["index"][0] = 500                                # claims 500 entries, holds 3
    mod.send_vector_host_to_device(v)
    mod.run()
    res = mod.send_results_device_to_host()
    got = M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, 0.0)
    good = M.make_sparse_vec([3, 77, 4000], [1.0, 2.0, 3.0])
    ref = O.spmspv(to_oracle(csc), good, O.MULADD, 0.0)
    assert np.allclose(got, ref, rtol=1e-6, atol=0)


# ------------------------------------------------------------------ round 4: the bin / fold kernels' other paths
def _random_csc(n, avg, seed, hub=None):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 2 * avg + 1, size=n)
    if hub:
        deg[rng.choice(n, size=hub[0], replace=False)] = hub[1]
    indptr = np.concatenate([[0], np.cumsum(deg)])
    rows = np.concatenate([np.sort(rng.choice(n, size=int(d), replace=False)) for d in deg]).astype(np.uint32)
    vals = rng.integers(1, 5, size=rows.shape[0]).astype(np.float32)
    return io.CSCMatrix(n, n, vals, rows, indptr.astype(np.uint32))


@pytest.mark.parametrize("sem", ["Arithmetic", "Logical", "Tropical"])
@pytest.mark.parametrize("tile_rows,n", [(64, 40000), (64, 200000), (192, 40000)])
def test_many_row_tiles(gpu, sem, tile_rows, n, monkeypatch):
    """Tiles forced small (GRAPHLILY_DEBUG spmspv_tile_rows, read at plan creation): 625 tiles -- more than compute units, so the fold
    hands them out by ticket and looks back over more tiles than one wavefront pass --, 3125 tiles -- more than the bin kernel
    has counters for, so every product goes through the dense accumulator and the fold merges it --, and a tile height that is
    not a power of two (row -> tile by multiplication).  Vectors: a few hundred entries (every workgroup cuts the products
    itself) and tens of thousands (the rendezvous)."""
    set_knob(monkeypatch, "spmspv_tile_rows", str(tile_rows))
    csc = _random_csc(n, 6, 5, hub=(3, 5000))
    op, zero = SEMIRINGS[sem]
    mask = rand01(n, 3)
    rng = np.random.default_rng(11)
    for cnt in (300, 3000, n // 3):
        cols = np.sort(rng.choice(n, size=cnt, replace=False)).astype(np.uint32)
        v = M.make_sparse_vec(cols, rng.integers(1, 4, size=cnt).astype(np.float32))
        for mask_name in ("NoMask", "WriteToZero"):
            got, mod = _run(gpu, csc, sem, mask_name, v, mask)
            ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS[mask_name])
            assert_parity(got, ref, op, "tiles of %d rows, n %d, %s %s %d" % (tile_rows, n, sem, mask_name, cnt))
            # and once more on the same plan: every between-runs invariant (cursors, tags, accumulator) was restored
            mod.run()
            again = M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), n, zero)
            assert np.array_equal(again, got)


def test_vector_longer_than_one_rendezvous_round(gpu):
    """2.3 M vector entries: more than the 2048 slices of 1024 entries the bin kernel's rendezvous holds, so it runs two rounds
    (two generations of tags).  One entry per column at row (7 c) mod n: the result is a permutation of the vector."""
    n = 2300000
    indptr = np.arange(n + 1, dtype=np.uint32)
    rows = ((np.arange(n, dtype=np.int64) * 7) % n).astype(np.uint32)
    csc = io.CSCMatrix(n, n, np.full(n, 2.0, np.float32), rows, indptr)
    vals = (np.arange(n) % 5 + 1).astype(np.float32)
    v = M.make_sparse_vec(np.arange(n, dtype=np.uint32), vals)
    got, mod = _run(gpu, csc, "Arithmetic", "NoMask", v, np.zeros(n, np.float32))
    ref = np.zeros(n, np.float32)
    ref[rows] = 2.0 * vals
    assert np.array_equal(got, ref)
    mod.run()       # the generation moved on by two: the next run's tags are fresh
    assert np.array_equal(M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), n, 0.0), ref)


def test_rendezvous_rounds_with_unequal_work(gpu):
    """4.5 M vector entries = three rendezvous rounds, and the first two hold almost no products (their columns are empty but
    for a handful): most workgroups leave rounds 0 and 1 at once (`lo >= P`) and publish the next round's slice sums while the
    few with work still poll the current one -- the slice words are double-buffered by round parity so that those pollers
    still find their round's tags (with one buffer they spun out and used the next round's sums).  Repeated so
    that the generations of several runs pass through both halves."""
    n = 4500000
    deg = np.zeros(n, np.uint32)
    deg[::100003] = 3            # ~45 short columns spread over the first two rounds
    deg[4300000:] = 1            # the last round: one entry per column
    indptr = np.zeros(n + 1, np.uint32)
    np.cumsum(deg, out=indptr[1:])
    nnz = int(indptr[-1])
    rows = ((np.arange(nnz, dtype=np.int64) * 11 + 5) % n).astype(np.uint32)
    csc = io.CSCMatrix(n, n, np.full(nnz, 2.0, np.float32), rows, indptr)
    vals = (np.arange(n) % 3 + 1).astype(np.float32)
    v = M.make_sparse_vec(np.arange(n, dtype=np.uint32), vals)
    got, mod = _run(gpu, csc, "Arithmetic", "NoMask", v, np.zeros(n, np.float32))
    ref = np.zeros(n, np.float64)
    np.add.at(ref, rows, 2.0 * np.repeat(vals, deg).astype(np.float64))
    assert np.array_equal(got, ref.astype(np.float32))
    for _ in range(3):
        mod.run()
        assert np.array_equal(M.convert_sparse_vec_to_dense_vec(mod.send_results_device_to_host(), n, 0.0), ref.astype(np.float32))


def test_duplicates_beyond_32_bits_of_products(gpu):
    """A vector that names one dense column 70 000 times: 4.6e9 products, more than the kernel's 32-bit product numbers
    hold -- it must notice (the slice sums saturate) and take the column-by-column path; the bins overflow into the dense
    accumulator on the way.  (||,&&): the result is simply every row."""
    n = 65536
    indptr = np.zeros(n + 1, np.uint32)
    indptr[1:] = n                                      # column 0 holds every row, the others are empty
    csc = io.CSCMatrix(n, n, np.ones(n, np.float32), np.arange(n, dtype=np.uint32), indptr)
    v = M.make_sparse_vec(np.zeros(70000, np.uint32), np.ones(70000, np.float32))
    mod = M.SpMSpVModule(512)
    mod.set_semiring(M.LogicalSemiring)
    mod.set_mask_type(M.kNoMask)
    mod.set_up_runtime()
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    mod.send_mask_host_to_device(np.zeros(n, np.float32))
    mod.vector_buf = None
    from graphlily_amd import capi
    mod.vector_buf = capi.DeviceBuffer(8 * v.shape[0])   # (send_vector_host_to_device truncates to num_cols + 1 entries)
    mod.vector_buf.write(v)
    mod.run()
    res = mod.send_results_device_to_host()
    assert int(res["index"][0]) == n
    assert np.array_equal(res["index"][1:n + 1], np.arange(n, dtype=np.uint32)) and np.all(res["val"][1:n + 1] == 1.0)
    # a normal run afterwards
    mod.send_vector_host_to_device(M.make_sparse_vec([0], [1.0]))
    mod.run()
    assert mod.get_results_nnz() == n


def test_wait_returns_the_count_without_a_copy(gpu):
    """gl_spmspv_wait: the operator's own completion record.  A non-blocking run, then wait(): the count equals the head
    element; a run recorded into a graph keeps no record (wait() is gl_sync and says so)."""
    from graphlily_amd import capi
    csc = _random_csc(30000, 8, 21)
    op, zero = SEMIRINGS["Arithmetic"]
    rng = np.random.default_rng(2)
    mod = M.SpMSpVModule(512)
    mod.set_semiring(M.ArithmeticSemiring)
    mod.set_mask_type(M.kNoMask)
    mod.set_up_runtime()
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    mod.blocking = False
    for cnt in (5, 700, 2500, 9000):     # one workgroup / one window / three, nine windows (no rendezvous)
        cols = np.sort(rng.choice(30000, size=cnt, replace=False)).astype(np.uint32)
        mod.send_vector_host_to_device(M.make_sparse_vec(cols, np.ones(cnt, np.float32)))
        for _ in range(3):
            mod.run()
            n = mod.plan_.wait()
            assert n is not None and n == int(mod.send_results_device_to_host()["index"][0]) and n > 0
            assert mod.get_results_nnz() == n
    # a run recorded into a graph keeps no record: wait() is gl_sync and says so; the replay computes the same list
    want = mod.send_results_device_to_host().copy()
    with capi.Graph.capture() as g:
        mod.run()
    g.launch()
    assert mod.plan_.wait() is None
    assert np.array_equal(mod.send_results_device_to_host(), want)
    mod.run()                                  # and an eager run after the replay reports again
    assert mod.plan_.wait() == int(want["index"][0])


def test_a_timed_out_run_fails_alone(gpu, monkeypatch):
    """A workgroup rendezvous that times out marks ITS run failed (gl_spmspv_bin.h kSyncErr holds the run's generation): the
    list is emptied, a blocking caller gets GL_ERR_HIP -- and nothing else happens.  Round 5 kept a boolean that only a
    host-reporting run cleared, so one timeout inside a recorded (non-reporting) run emptied every later run and replay on the
    plan.  The timeout is injected (GRAPHLILY_DEBUG spmspv_inject_timeout, read when the run is enqueued)."""
    from graphlily_amd import capi
    csc = _random_csc(30000, 8, 22)
    rng = np.random.default_rng(3)
    mod = M.SpMSpVModule(512)
    mod.set_semiring(M.ArithmeticSemiring)
    mod.set_mask_type(M.kNoMask)
    mod.set_up_runtime()
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    mod.blocking = False
    cols = np.sort(rng.choice(30000, size=2500, replace=False)).astype(np.uint32)
    v = M.make_sparse_vec(cols, np.ones(2500, np.float32))
    mod.send_vector_host_to_device(v)
    mod.run()
    n_good = mod.plan_.wait()
    want = mod.send_results_device_to_host().copy()
    assert n_good is not None and n_good > 0 and mod.plan_.failed_runs() == 0
    ref = O.spmspv(to_oracle(csc), v, O.MULADD, 0.0, np.zeros(30000, np.float32), O.NOMASK)
    assert np.allclose(M.convert_sparse_vec_to_dense_vec(want, 30000, 0.0), ref, rtol=1e-5, atol=1e-6)
    # 1. a reporting run that times out: GL_ERR_HIP from the wait, an empty list, one failed run on the books
    set_knob(monkeypatch, "spmspv_inject_timeout", 1)
    mod.run()
    with pytest.raises(capi.GraphLilyError) as ei:
        mod.plan_.wait()
    assert ei.value.code == capi.GL_ERR_HIP
    assert int(mod.send_results_device_to_host()["index"][0]) == 0 and mod.plan_.failed_runs() == 1
    # 2. a recorded run that times out on every replay: no record, empty lists, the counter says so
    with capi.Graph.capture() as g:
        mod.run()
    set_knob(monkeypatch, "spmspv_inject_timeout", None)
    for k in range(3):
        g.launch()
        assert mod.plan_.wait() is None
        assert int(mod.send_results_device_to_host()["index"][0]) == 0
        assert mod.plan_.failed_runs() == 2 + k
    # 3. ... and the runs after it are healthy again, eager and recorded alike (the stale mark belongs to another generation)
    mod.run()
    assert mod.plan_.wait() == n_good
    assert np.array_equal(mod.send_results_device_to_host(), want)
    with capi.Graph.capture() as g2:
        mod.run()
    g.launch()                                  # the failing graph in between
    g2.launch()
    assert mod.plan_.wait() is None
    assert np.array_equal(mod.send_results_device_to_host(), want) and mod.plan_.failed_runs() == 5
    g.destroy()
    g2.destroy()


@pytest.mark.parametrize("sem", ["Arithmetic", "Tropical"])
def test_tiny_run_whose_rows_are_adjacent(gpu, sem):
    """The one-launch kernel orders the rows it reached with a counting pass over 2048 buckets of consecutive rows + a rank
    inside each bucket.  600 K rows make a bucket 512 rows wide; three columns that cover 1300 ADJACENT rows (two of them the
    same rows: sums / minima of two products) put ~500 keys into each of three buckets -- the case the buckets do not help,
    which must still come out in ascending row order -- next to a column at the far end of the matrix."""
    n, ncols = 600000, 64
    rows = [np.arange(200000, 200700), np.arange(200300, 201300), np.arange(200300, 201300)[::3], np.array([5, 599999])]
    indptr = np.zeros(ncols + 1, np.uint32)
    for c, r in enumerate(rows):
        indptr[c + 1] = indptr[c] + len(r)
    indptr[len(rows) + 1:] = indptr[len(rows)]
    idx = np.concatenate(rows).astype(np.uint32)
    rng = np.random.default_rng(5)
    data = (rng.integers(1, 9, size=len(idx)) / np.float32(8)).astype(np.float32)
    csc = io.CSCMatrix(n, ncols, data, idx, indptr)
    v = M.make_sparse_vec(np.arange(4, dtype=np.uint32), np.array([0.5, 0.25, 0.75, 1.0], np.float32))
    mask = np.zeros(n, np.float32)
    op, zero = SEMIRINGS[sem]
    got, mod = _run(gpu, csc, sem, "NoMask", v, mask)
    assert mod.tiny_ is not None                     # 4 entries, 2036 products: one launch
    res = mod.send_results_device_to_host()
    cnt = int(res["index"][0])
    assert np.all(np.diff(res["index"][1:cnt + 1].astype(np.int64)) > 0)
    ref = O.spmspv(to_oracle(csc), v, op, zero, mask, MASKS["NoMask"])
    assert_parity(got, ref, op, "adjacent rows %s" % sem)
