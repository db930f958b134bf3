"""Shared helpers for the parity tests (the oracle is the checker, never the thing under test)."""
import numpy as np

from graphlily_amd import datasets, io
from oracle import oracle as O


def to_oracle(m):
    cls = O.CSC if isinstance(m, io.CSCMatrix) else O.CSR
    return cls(m.num_rows, m.num_cols, m.adj_data, m.adj_indices, m.adj_indptr)


def rand01(n, seed):
    """The reference draws x and masks as rand() % 2 (tests/test_module_spmv_spmspv.cpp:105-111);
    seeded here."""
    return np.random.default_rng(seed).integers(0, 2, size=n).astype(np.float32)


def named_matrix(name):
    if name == "dense_32":
        return datasets.dense(32)
    if name == "dense_1K":
        return datasets.dense(1024)
    if name == "uniform_10K_10":
        return datasets.uniform(10000, 10, seed=7)
    if name == "rmat_20K":          # power-law: a few rows far longer than a tile, many empty rows
        return datasets.rmat(20000, 400000, seed=11, symmetric=False)
    if name == "rmat_sym_50K":
        return datasets.rmat(50000, 1500000, seed=12, symmetric=True)
    if name == "gplus_small":       # googleplus stand-in at 1/8 scale
        return datasets.paper_graph("googleplus", scale=0.125)
    raise KeyError(name)


def spmv_prepare(name, row_div=128, col_div=8):
    """Matrix preparation of the reference SpMV test (tests/test_module_spmv_spmspv.cpp:144-151):
    pad rows to num_hbm_channels*pack_size and cols to pack_size, values = 1/num_rows."""
    m = named_matrix(name)
    io.util_round_csr_matrix_dim(m, row_div, col_div)
    m.adj_data = np.full(m.adj_data.shape[0], np.float32(1.0 / m.num_rows), dtype=np.float32)
    return m


SEMIRINGS = {"Arithmetic": (0, 0.0), "Logical": (1, 0.0), "Tropical": (2, 255.0), "TropicalFloatInf": (2, 999999999.0)}
MASKS = {"NoMask": 0, "WriteToZero": 1, "WriteToOne": 2}


U32 = 2.0 ** -24   # fp32 unit roundoff


def arith_exact(m, x, rows=None):
    """float64 evaluation of the (+,x) product and the per-row data the float tolerance needs:
    exact[r] = sum a_i x_i, abs_sum[r] = sum |a_i x_i|, length[r]."""
    n = m.nnz
    prod = m.adj_data[:n].astype(np.float64) * np.asarray(x, np.float64)[m.adj_indices[:n]]
    lens = np.diff(m.adj_indptr.astype(np.int64))
    row_of = np.repeat(np.arange(m.num_rows), lens)
    exact = np.bincount(row_of, weights=prod, minlength=m.num_rows)
    abs_sum = np.bincount(row_of, weights=np.abs(prod), minlength=m.num_rows)
    return exact, abs_sum, lens


def assert_arith_parity(got, ref, exact, abs_sum, lens, what="", keep=None):
    """Float (+,x) tolerance, stated in full:
      (1) |got - exact| <= 1e-5 |exact| (+ 4u abs_sum for cancellation): the HIP result is within the
          north_star's 1e-5 relative of the exactly evaluated product;
      (2) |got - ref| <= 1e-5 |ref| + L u abs_sum: against the fp32 oracle, whose sequential
          accumulation (spmv_module.h:495) itself carries the classical forward error (L-1) u sum|a_i x_i|
          -- on rows with thousands of entries that alone exceeds 1e-5, whatever order the device uses.
    `keep` selects the rows that were not masked off (masked rows must be exactly 0)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if keep is None:
        keep = np.ones(got.shape[0], bool)
    if np.any(got[~keep] != 0):
        raise AssertionError("%s: masked-off rows must be exactly 0" % what)
    e1 = np.abs(got - exact) - (1e-5 * np.abs(exact) + 4 * U32 * abs_sum)
    e2 = np.abs(got - ref) - (1e-5 * np.abs(ref) + np.maximum(lens, 1) * U32 * abs_sum)
    for name, e in (("vs exact", e1), ("vs oracle", e2)):
        bad = np.nonzero((e > 0) & keep)[0]
        if bad.size:
            i = bad[0]
            raise AssertionError("%s %s: %d rows out of tolerance, first row %d (len %d): got %r ref %r exact %r" %
                                 (what, name, bad.size, i, lens[i], got[i], ref[i], exact[i]))


def assert_parity(got, ref, op, what=""):
    """Bit-exact for the boolean and (min,+) semirings.  For float (+,x) this short form is the
    1e-5 relative bar of the north_star and is only used where rows are short; long-row cases go
    through assert_arith_parity."""
    got = np.asarray(got, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float32)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    if op == 0:
        ok = np.allclose(got, ref, rtol=1e-5, atol=1e-9)
        if not ok:
            bad = np.nonzero(~np.isclose(got, ref, rtol=1e-5, atol=1e-9))[0]
            raise AssertionError("%s: %d mismatches, first at %d: got %r ref %r" %
                                 (what, bad.size, bad[0], got[bad[0]], ref[bad[0]]))
    else:
        if not np.array_equal(got, ref):
            bad = np.nonzero(got != ref)[0]
            raise AssertionError("%s: %d mismatches (bit-exact required), first at %d: got %r ref %r" %
                                 (what, bad.size, bad[0], got[bad[0]], ref[bad[0]]))


def set_knob(monkeypatch, key, value):
    """One planner override of GRAPHLILY_DEBUG="key=value,..." (csrc/gl_spmv_plan.h debug_knob: test hooks that force a decision
    the planner would take from the matrix -- plan shape, hot table, helper mode, tile height ...); value None removes the key."""
    import os
    cur = dict(kv.split("=", 1) for kv in os.environ.get("GRAPHLILY_DEBUG", "").split(",") if kv)
    if value is None:
        cur.pop(key, None)
    else:
        cur[key] = str(value)
    if cur:
        monkeypatch.setenv("GRAPHLILY_DEBUG", ",".join("%s=%s" % kv for kv in cur.items()))
    else:
        monkeypatch.delenv("GRAPHLILY_DEBUG", raising=False)
