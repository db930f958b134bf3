"""CPU suite, part 1: pin the oracle against every golden vector available for this path, and check
the product's host-side formatters (graphlily_amd/io.py) against the oracle."""
import json
import os

import numpy as np
import pytest

from graphlily_amd import datasets, io
from oracle import oracle as O

from helpers import to_oracle


@pytest.fixture(scope="module")
def G(golden_dir):
    with open(os.path.join(golden_dir, "reference_known_answers.json")) as f:
        return json.load(f)


def _load(golden_dir, name, pad=True, sssp=False):
    m = O.load_csr_matrix_from_float_npz(os.path.join(golden_dir, name + "_csr_float32.npz"))
    if sssp:
        O.sssp_preprocess(m)
    if pad:
        O.util_round_csr_matrix_dim(m, 128, 128)   # apps pad both dims to num_channels*pack_size
    return m


# ---------------------------------------------------------------- reference tests/test_io.cpp KATs
def test_io_load_npz(G, golden_dir):
    e = G["test_io"]["eye_10"]
    m = _load(golden_dir, "eye_10", pad=False)
    assert (m.num_rows, m.num_cols) == (e["num_rows"], e["num_cols"])
    assert m.adj_data.tolist() == e["adj_data"]
    assert m.adj_indices.tolist() == e["adj_indices"]
    assert m.adj_indptr.tolist() == e["adj_indptr"]


def _csr1(G):
    c = G["test_io"]["csr_matrix_1"]
    return O.CSR(c["num_rows"], c["num_cols"], c["adj_data"], c["adj_indices"], c["adj_indptr"])


def test_io_csr2csc(G):
    e = G["test_io"]["csr2csc"]
    csc = O.csr2csc(_csr1(G))
    assert csc.adj_data.tolist() == e["adj_data"]
    assert csc.adj_indices.tolist() == e["adj_indices"]
    assert csc.adj_indptr.tolist() == e["adj_indptr"]


def test_io_round_dim(G):
    e = G["test_io"]["round_dim"]
    m = _csr1(G)
    O.util_round_csr_matrix_dim(m, e["row_divisor"], e["col_divisor"])
    assert (m.num_rows, m.num_cols) == (e["num_rows"], e["num_cols"])
    assert m.adj_indptr.tolist() == [0, 4, 6, 7, 8, 8, 8]


def test_io_normalize(G):
    m = _csr1(G)
    O.util_normalize_csr_matrix_by_outdegree(m)
    assert m.adj_data[:4].tolist() == G["test_io"]["normalize_first4"]


# ---------------------------------------------------------------- SURVEY 8(c) reference outputs
def test_app_known_answers(G, golden_dir):
    for a in G["survey_8c"]["apps"]:
        call = a["call"]
        if call == "spmv_arith_nomask":
            m = _load(golden_dir, a["matrix"])
            y = O.spmv(m, (np.arange(m.num_cols) % 7).astype(np.float32), O.MULADD, 0)
        elif call == "bfs":
            m = _load(golden_dir, a["matrix"])
            m.adj_data[:] = 1
            y = O.bfs(m, a["source"], a["iters"])
        elif call == "pagerank":
            m = _load(golden_dir, a["matrix"])
            O.util_normalize_csr_matrix_by_outdegree(m)
            m.adj_data *= np.float32(a["damping"])
            y = O.pagerank(m, a["damping"], a["iters"])
        else:
            m = _load(golden_dir, a["matrix"], sssp=True)
            y = O.sssp(m, a["source"], a["iters"], 255)
        exp = np.array(a["first"], dtype=np.float32)
        # recorded with 6 significant digits (relative rounding up to 5e-6)
        assert np.allclose(y[:len(exp)], exp, rtol=1e-5, atol=1e-9), (a["matrix"], call, y[:len(exp)])


def test_semiring_mask_known_answers(G, golden_dir):
    S = G["survey_8c"]["semiring_mask"]
    m = _load(golden_dir, "line_8")
    m.adj_data[:] = 1
    csc = O.csr2csc(m)
    mask = (np.arange(128) % 2).astype(np.float32)
    x = ((3 * np.arange(128)) % 5).astype(np.float32)
    v = O.make_sparse_vec([p[0] for p in S["spmspv_v"]], [p[1] for p in S["spmspv_v"]])
    ops = {"arith": (O.MULADD, 0), "logical": (O.ANDOR, 0), "tropical": (O.ADDMIN, S["tropical_zero"])}
    mts = {"nomask": O.NOMASK, "wzero": O.WRITETOZERO, "wone": O.WRITETOONE}
    for on, (op, z) in ops.items():
        for mn, mt in mts.items():
            y = O.spmv(m, x, op, z) if mt == O.NOMASK else O.spmv(m, x, op, z, mask, mt)
            assert y[:10].tolist() == S["spmv"][on][mn], ("spmv", on, mn)
            y = O.spmspv(csc, v, op, z, mask, mt)
            assert y[:10].tolist() == S["spmspv"][on][mn], ("spmspv", on, mn)


# ---------------------------------------------------------------- sssp preprocess vs a literal replay
def _literal_sssp_preprocess(indptr, indices):
    """Line-by-line replay of app/sssp.h:16-62 on Python lists (real in-place inserts)."""
    indptr, indices = list(map(int, indptr)), list(map(int, indices))
    data = [1.0] * len(indices)
    n = len(indptr) - 1
    nnz_each_row = [indptr[i + 1] - indptr[i] for i in range(n)]
    for r in range(n):
        start, end = indptr[r], indptr[r + 1]
        if start == end:
            data.insert(start, 0.0)
            indices.insert(start, r)
            nnz_each_row[r] += 1
        else:
            add = False
            i = start
            while i < end:
                c = indices[i]
                if c == r:
                    data[i] = 0.0
                    break
                elif c > r:
                    add = True
                    data.insert(i, 0.0)
                    indices.insert(i, r)
                    break
                elif i == end - 1:
                    add = True
                    data.insert(i, 0.0)
                    indices.insert(i, r)
                    break
                i += 1
            if add:
                nnz_each_row[r] += 1
        indptr[r + 1] = indptr[r] + nnz_each_row[r]
    return indptr, indices, data


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_sssp_preprocess_matches_literal_replay(seed):
    rng = np.random.default_rng(seed)
    n = 60
    dens = [0.02, 0.1, 0.3, 0.6][seed]
    A = rng.random((n, n)) < dens
    if seed == 1:
        A[np.arange(0, n, 3), np.arange(0, n, 3)] = True   # some existing diagonals
    if seed == 2:
        A[5] = True                                        # one dense row
        A[40:] = False                                     # trailing empty rows
    rows, cols = np.nonzero(A)
    indptr = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    ip, ix, dv = _literal_sssp_preprocess(indptr, cols)
    for impl in ("oracle", "product"):
        if impl == "oracle":
            m = O.CSR(n, n, np.ones(len(cols), np.float32), cols, indptr)
            O.sssp_preprocess(m)
        else:
            m = io.CSRMatrix(n, n, np.ones(len(cols), np.float32), cols, indptr)
            io.sssp_add_self_edges(m)
        assert m.adj_indptr.tolist() == ip, impl
        assert m.adj_indices.tolist() == ix, impl
        assert m.adj_data.tolist() == dv, impl


# ---------------------------------------------------------------- product host formatters vs oracle
@pytest.mark.parametrize("gen", ["uniform", "rmat", "rmat_2M"])
def test_product_formatters_match_oracle(gen):
    if gen == "rmat_2M":      # > 1 Mi non-zeros: the multi-threaded path of gl_csr2csc
        m = datasets.rmat(100000, 2200000, seed=8)
    else:
        m = datasets.uniform(500, 7, seed=3) if gen == "uniform" else datasets.rmat(3000, 40000, seed=5)
    m.adj_data = np.random.default_rng(1).random(m.nnz, dtype=np.float32)
    om = to_oracle(m)
    io.util_round_csr_matrix_dim(m, 128, 128)
    O.util_round_csr_matrix_dim(om, 128, 128)
    assert (m.num_rows, m.num_cols) == (om.num_rows, om.num_cols)
    assert np.array_equal(m.adj_indptr, om.adj_indptr)
    c, oc = io.csr2csc(m), O.csr2csc(om)
    assert np.array_equal(c.adj_indptr, oc.adj_indptr)
    assert np.array_equal(c.adj_indices, oc.adj_indices)
    assert np.array_equal(c.adj_data, oc.adj_data)
    io.util_normalize_csr_matrix_by_outdegree(m)
    O.util_normalize_csr_matrix_by_outdegree(om)
    assert np.array_equal(m.adj_data, om.adj_data)


def test_sparse_dense_roundtrip():
    from graphlily_amd import module as M
    v = M.make_sparse_vec([3, 9, 1], [1.5, 2.5, 3.5])
    d = M.convert_sparse_vec_to_dense_vec(v, 12, 255.0)
    assert np.array_equal(d, O.convert_sparse_vec_to_dense_vec(v, 12, 255.0))


def test_oracle_edge_cases():
    # empty matrix, empty frontier, all-masked
    m = O.CSR(4, 4, [], [], [0, 0, 0, 0, 0])
    assert O.spmv(m, np.ones(4, np.float32), O.ADDMIN, 255).tolist() == [255] * 4
    csc = O.csr2csc(m)
    v = O.make_sparse_vec([], [])
    assert O.spmspv(csc, v, O.MULADD, 0).tolist() == [0] * 4
    inout = np.zeros(4, np.float32)
    with pytest.raises(ValueError):
        O.assign_dense(O.NOMASK, np.zeros(4, np.float32), inout, 4, 1.0)
    nf = O.assign_sparse_new_frontier(O.make_sparse_vec([1, 2], [5.0, 0.5]), np.array([9, 9, 0.25, 9], np.float32))
    assert nf["index"].tolist() == [1, 1] and nf["val"].tolist() == [0.0, 5.0]
