"""The reference's own module-test cases whose inputs its test sources determine completely
(tests/test_module_apply.cpp, tests/test_module_spmv_spmspv.cpp), with expectations derived by closed forms in
tests/golden/make_reference_test_cases.py -- independently of oracle/ and of the HIP library.

  * CPU (-m "not gpu"): the oracle must reproduce every expectation BIT FOR BIT (same float32 order as the reference loops);
  * GPU (-m gpu): the HIP modules, driven through the reference's module API, must match them -- bit-exact for
    (||,&&), (min,+) and the element-wise modules, 1e-5 relative for (+,x) (the device accumulates in f64)."""
import json
import os

import numpy as np
import pytest

from graphlily_amd import datasets, io, module as M
from oracle import oracle as O

F = np.float32
MASK = {"kNoMask": 0, "kMaskWriteToZero": 1, "kMaskWriteToOne": 2}
OP = {"Arithmetic": 0, "Logical": 1, "Tropical": 2}


@pytest.fixture(scope="module")
def C(golden_dir):
    with open(os.path.join(golden_dir, "reference_test_cases.json")) as f:
        return json.load(f)


def f32(a):
    return np.asarray(a, dtype=F)


def _dense_csr(live, n):
    """dense_32 as the reference test prepares it (:144-152): padded to n x n, values 1/n"""
    m = datasets.dense(live)
    io.util_round_csr_matrix_dim(m, n, n)
    m.adj_data = np.full(m.nnz, F(1.0 / m.num_rows), dtype=F)
    return m


def _dense_csc(n):
    c = io.csr2csc(datasets.dense(n))
    c.adj_data = np.full(c.nnz, F(1.0 / c.num_rows), dtype=F)
    return c


def _sv(index, val):
    return M.make_sparse_vec(np.asarray(index, dtype=np.uint32), f32(val))


def test_fixture_is_reproducible(C, golden_dir, tmp_path):
    """the committed JSON is what the committed generator writes (glibc rand(), default seed)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cases", os.path.join(golden_dir, "make_reference_test_cases.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert g.apply_cases() == C["apply"]
    assert g.spmv_cases() == C["spmv_dense_32"]
    assert g.spmspv_cases() == C["spmspv"]


# ------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_apply_cases(C):
    a = C["apply"]
    c = a["ewise_add"]
    assert np.array_equal(O.ewise_add(f32(c["in"]), len(c["in"]), c["val"]), f32(c["expected"]))
    c = a["assign_dense_write_to_one"]
    inout = f32(c["inout"]).copy()
    O.assign_dense(O.WRITETOONE, f32(c["mask"]), inout, len(inout), c["val"])
    assert np.array_equal(inout, f32(c["expected"]))
    c = a["assign_sparse"]
    inout = f32(c["inout"]).copy()
    O.assign_sparse(O.make_sparse_vec(c["mask_index"], c["mask_val"]), inout, c["val"])
    assert np.array_equal(inout, f32(c["expected"]))
    c = a["assign_sparse_new_frontier"]
    inout = f32(c["inout"]).copy()
    nf = O.assign_sparse_new_frontier(O.make_sparse_vec(c["mask_index"], c["mask_val"]), inout)
    assert np.array_equal(inout, f32(c["expected_inout"]))
    assert int(nf["index"][0]) == len(c["expected_frontier_index"]) and nf["val"][0] == 0.0
    assert nf["index"][1:].tolist() == c["expected_frontier_index"]
    assert np.array_equal(nf["val"][1:], f32(c["expected_frontier_val"]))


def test_oracle_spmv_dense_32(C):
    S = C["spmv_dense_32"]
    m = _dense_csr(S["live"], S["n"])
    om = O.CSR(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)
    assert float(m.adj_data[0]) == S["value"]
    for c in S["cases"]:
        mt = MASK[c["mask_type"]]
        y = O.spmv(om, f32(c["x"]), OP[c["semiring"]], 0.0) if mt == 0 else O.spmv(om, f32(c["x"]), OP[c["semiring"]], 0.0, f32(c["mask"]), mt)
        assert np.array_equal(y, f32(c["expected"])), (c["semiring"], c["mask_type"])


def test_oracle_spmspv_cases(C):
    S = C["spmspv"]
    c = S["conflict1024"]
    m = datasets.conflict(c["n"])
    om = O.CSC(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)
    y = O.spmspv(om, O.make_sparse_vec(c["vector_index"], c["vector_val"]), O.MULADD, 0.0, f32(c["mask"]), O.NOMASK)
    assert np.array_equal(y, f32(c["expected"]))
    D = S["dense1K"]
    m = _dense_csc(D["n"])
    om = O.CSC(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)
    for c in D["cases"]:
        y = O.spmspv(om, O.make_sparse_vec(c["vector_index"], c["vector_val"]), OP[c["semiring"]], c["zero"], f32(c["mask"]),
                     MASK[c["mask_type"]])
        assert np.array_equal(y, f32(c["expected"])), (c["semiring"], c["mask_type"])


# ------------------------------------------------------------------------------------------- HIP modules (GPU)
def _close(got, exp, op):
    exp = f32(exp)
    if op == 0:   # (+,x): f64 accumulation on the device vs the float32 sequential sum
        err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
        assert np.all(err <= 1e-5 * np.abs(exp.astype(np.float64))), float(err.max())
    else:
        assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_gpu_apply_cases(gpu, C):
    a = C["apply"]
    c = a["ewise_add"]
    mod = M.eWiseAddModule()
    mod.set_up_runtime()
    mod.send_in_host_to_device(f32(c["in"]))
    mod.allocate_out_buf(len(c["in"]))
    mod.run(len(c["in"]), c["val"])
    assert np.array_equal(mod.send_out_device_to_host(), f32(c["expected"]))
    c = a["assign_dense_write_to_one"]
    mod = M.AssignVectorDenseModule()
    mod.set_up_runtime()
    mod.set_mask_type(M.kMaskWriteToOne)
    mod.send_mask_host_to_device(f32(c["mask"]))
    mod.send_inout_host_to_device(f32(c["inout"]))
    mod.run(len(c["inout"]), c["val"])
    assert np.array_equal(mod.send_inout_device_to_host(), f32(c["expected"]))
    c = a["assign_sparse"]
    mod = M.AssignVectorSparseModule(False)
    mod.set_up_runtime()
    mod.send_mask_host_to_device(_sv(c["mask_index"], c["mask_val"]))
    mod.send_inout_host_to_device(f32(c["inout"]))
    mod.run(c["val"])
    assert np.array_equal(mod.send_inout_device_to_host(), f32(c["expected"]))
    c = a["assign_sparse_new_frontier"]
    mod = M.AssignVectorSparseModule(True)
    mod.set_up_runtime()
    mod.send_mask_host_to_device(_sv(c["mask_index"], c["mask_val"]))
    mod.send_inout_host_to_device(f32(c["inout"]))
    mod.run()
    assert np.array_equal(mod.send_inout_device_to_host(), f32(c["expected_inout"]))
    nf = mod.send_new_frontier_device_to_host()
    n = int(nf["index"][0])
    assert n == len(c["expected_frontier_index"]) and nf["val"][0] == 0.0
    assert nf["index"][1:n + 1].tolist() == c["expected_frontier_index"]
    assert np.array_equal(nf["val"][1:n + 1], f32(c["expected_frontier_val"]))


@pytest.mark.gpu
def test_gpu_spmv_dense_32(gpu, C):
    S = C["spmv_dense_32"]
    m = _dense_csr(S["live"], S["n"])
    for c in S["cases"]:
        mod = M.SpMVModule(16, 1024, 256)
        mod.set_semiring(M.ArithmeticSemiring if c["semiring"] == "Arithmetic" else M.LogicalSemiring)
        mod.set_mask_type(MASK[c["mask_type"]])
        mod.set_up_runtime()
        mod.load_and_format_matrix(m, c["skip_empty_rows"])
        mod.send_matrix_host_to_device()
        mod.send_vector_host_to_device(f32(c["x"]))
        mod.send_mask_host_to_device(f32(c["mask"]))
        mod.run()
        _close(mod.send_results_device_to_host(), c["expected"], OP[c["semiring"]])


def _gpu_spmspv(csc, c, sem, zero, mask_type):
    mod = M.SpMSpVModule(512)
    mod.set_semiring(M.SemiringType(OP[sem], 1.0, zero))
    mod.set_mask_type(MASK[mask_type])
    mod.set_up_runtime()
    mod.load_and_format_matrix(csc)
    mod.send_matrix_host_to_device()
    mod.send_mask_host_to_device(f32(c["mask"]))
    mod.send_vector_host_to_device(_sv(c["vector_index"], c["vector_val"]))
    mod.run()
    res = mod.send_results_device_to_host()
    return M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, zero)


@pytest.mark.gpu
def test_gpu_spmspv_cases(gpu, C):
    S = C["spmspv"]
    c = S["conflict1024"]
    _close(_gpu_spmspv(datasets.conflict(c["n"]), c, "Arithmetic", 0.0, "kNoMask"), c["expected"], 0)
    D = S["dense1K"]
    csc = _dense_csc(D["n"])
    for c in D["cases"]:
        _close(_gpu_spmspv(csc, c, c["semiring"], c["zero"], c["mask_type"]), c["expected"], OP[c["semiring"]])
