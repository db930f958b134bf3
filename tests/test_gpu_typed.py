"""GPU parity for the reference's other value types (graphlily/global.h:62-64): `unsigned` and
ap_ufixed<32, 8, AP_RND, AP_SAT>, the shipped default.  The words run through the same plans and kernels as float
(template instantiations over op + 3 * value type) and must equal the oracle's integer restatement of the device ALUs
(hw/ufixed_pe_fwd.h:23-65) BIT FOR BIT: all three semirings for both types.  (+,x) is exact in any order for `unsigned`
(modular sums) and for the fixed point too: the product is rounded and saturated once, the terms are non-negative, and a
clamped running sum of non-negative terms is min(sum, 2^32 - 1) whatever the order -- the reference's shipped PageRank
arithmetic."""
import numpy as np
import pytest

from graphlily_amd import capi, io
from oracle import oracle as O

from helpers import named_matrix

pytestmark = pytest.mark.gpu

VT = {"unsigned": capi.GL_VAL_UNSIGNED, "ufixed": capi.GL_VAL_UFIXED_32_8}
OPS = {"Arithmetic": 0, "Logical": 1, "Tropical": 2}
MASKS = {"NoMask": 0, "WriteToZero": 1, "WriteToOne": 2}


def _zero_word(vt, op):
    if op != 2:
        return 0
    return 0xffffffff if vt == capi.GL_VAL_UNSIGNED else 255 << 24      # UINT_INF / UFIXED_INF (global.h:78-79)


def _values(vt, op, rng, n, kind):
    """Words with the corner cases the types have: zeros (dropped by &&), words near saturation, 0x80000000 (a float -0.0)."""
    if vt == capi.GL_VAL_UNSIGNED:
        w = rng.integers(0, 50 if op != 0 else 1 << 20, size=n, dtype=np.uint64).astype(np.uint32)
    elif op == 0:   # (+,x): products around 15, so that short rows stay below 256.0 and long ones saturate
        w = capi.words_from_float(vt, rng.random(n, dtype=np.float32) * (2.0 if kind == "matrix" else 30.0))
    else:
        w = capi.words_from_float(vt, rng.random(n, dtype=np.float32) * (8.0 if kind == "matrix" else 120.0))
    special = rng.random(n)
    w[special < 0.05] = 0
    w[(special > 0.05) & (special < 0.07)] = 0x80000000
    if op == 2:
        w[(special > 0.07) & (special < 0.10)] = 0xfffffff0            # saturates (ufixed) / wraps (unsigned) when added
    return w


@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", list(OPS))
@pytest.mark.parametrize("vt_name", list(VT))
@pytest.mark.parametrize("name", ["uniform_10K_10", "rmat_sym_50K"])
def test_spmv_words(gpu, name, vt_name, sem, mask_name):
    vt, op, mt = VT[vt_name], OPS[sem], MASKS[mask_name]
    m = named_matrix(name)
    io.util_round_csr_matrix_dim(m, 128, 128)
    rng = np.random.default_rng(hash((name, vt_name, sem)) % (1 << 31))
    zero = _zero_word(vt, op)
    x = _values(vt, op, rng, m.num_cols, "vector")
    if op == 2:
        x[rng.random(m.num_cols) < 0.5] = zero
    mask = rng.integers(0, 3, size=m.num_rows).astype(np.uint32)
    mask[mask == 2] = 0x80000000                                        # non-zero as a word, -0.0 as a float
    dx, dm, dy = capi.DeviceBuffer.from_host(x), capi.DeviceBuffer.from_host(mask), capi.DeviceBuffer(4 * m.num_rows)
    for layout, data in (("general", _values(vt, op, rng, m.nnz, "matrix")),
                         ("pattern", np.full(m.nnz, 3 if vt == capi.GL_VAL_UNSIGNED else 3 << 23, dtype=np.uint32))):
        plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, data.view(np.float32))
        assert plan.info()["layout"] == layout
        plan.run_typed(dx, dm if mt else None, dy, op, zero, mt, vt)
        got = dy.read(np.uint32, m.num_rows)
        ref = O.spmv_words(m.adj_indptr, m.adj_indices, data, x, op, vt, zero, mask if mt else None, mt)
        assert np.array_equal(got, ref), "%s %s %s %s %s: %d rows differ" % (name, vt_name, sem, mask_name, layout, int((got != ref).sum()))
        assert len(np.unique(ref)) >= 2
        if vt == capi.GL_VAL_UFIXED_32_8 and op == 0 and layout == "general" and mt == 0:
            # the saturating case: row sums above 256.0 clamp to 2^32 - 1, and plenty of rows stay below
            sat = int((ref == 0xffffffff).sum())
            assert sat > 0 and int((ref != 0xffffffff).sum()) > m.num_rows // 10, sat


def test_spmv_ufixed_muladd_saturates_and_rounds(gpu):
    """(+,x) over ap_ufixed<32,8,AP_RND,AP_SAT> on rows built to hit the corners (hw/ufixed_pe_fwd.h:29-31, :53-55): a product
    that rounds half up, a product that saturates on its own, a row whose SUM crosses 256.0 only with its last term, a hub
    row of 20 000 terms (spread over private LDS slots by the plan), and a split plan (row shard)."""
    n = 4096 * 8
    rng = np.random.default_rng(11)
    rows, cols, vals = [], [], []
    rows += [0, 0]; cols += [0, 1]; vals += [1, 3]                       # x = 0.5 ulp-ish words: (1 * x + 2^23) >> 24
    rows += [1]; cols += [2]; vals += [0xffffffff]                       # 255.99.. * 200 saturates in the product
    rows += [2] * 5; cols += [3, 4, 5, 6, 7]; vals += [64 << 24] * 5     # 4 x 64 = 256 > max: clamps at the 4th term
    hub = rng.choice(n, size=20000, replace=False)
    rows += [3] * 20000; cols += hub.tolist(); vals += rng.integers(1, 1 << 18, size=20000).tolist()
    body = rng.integers(8, n, size=200000)
    rows += body.tolist(); cols += rng.integers(0, n, size=200000).tolist(); vals += rng.integers(0, 1 << 25, size=200000).tolist()
    order = np.lexsort((np.array(cols), np.array(rows)))
    r, c, v = np.array(rows)[order], np.array(cols, dtype=np.uint32)[order], np.array(vals, dtype=np.uint64).astype(np.uint32)[order]
    keep = np.ones(r.shape[0], bool)
    keep[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    r, c, v = r[keep], c[keep], v[keep]
    indptr = np.zeros(n + 1, np.uint32)
    np.add.at(indptr, r + 1, 1)
    indptr = np.cumsum(indptr, dtype=np.uint64).astype(np.uint32)
    x = rng.integers(0, 1 << 26, size=n, dtype=np.uint64).astype(np.uint32)
    x[0], x[1] = (1 << 23), (1 << 23) - 1                                 # 1 * 2^23 rounds up to 1; 3 * (2^23 - 1) rounds to 1
    x[2] = 200 << 24
    x[3:8] = 1 << 24
    vt = capi.GL_VAL_UFIXED_32_8
    ref = O.spmv_words(indptr, c, v, x, 0, vt, 0)
    assert ref[0] == 2 and ref[1] == 0xffffffff and ref[2] == 0xffffffff
    dx = capi.DeviceBuffer.from_host(x)
    for r0, r1 in ((0, n), (0, 4096), (4096, n)):
        plan = capi.SpMVPlan(n, n, indptr, c, v.view(np.float32), r0, r1, flags=capi.GL_PLAN_KEEP_VALUES)
        dy = capi.DeviceBuffer(4 * n)
        dy.write(np.full(n, 0xdeadbeef, np.uint32))
        plan.run_typed(dx, None, dy, 0, 0, 0, vt)
        got = dy.read(np.uint32, n)
        assert np.array_equal(got[r0:r1], ref[r0:r1]), (r0, r1, int((got[r0:r1] != ref[r0:r1]).sum()))
        # a non-zero `zero` is added once, saturating (the accumulator starts from it, spmv_module.h:487)
        plan.run_typed(dx, None, dy, 0, 250 << 24, 0, vt)
        ref_z = O.spmv_words(indptr, c, v, x, 0, vt, 250 << 24)
        assert np.array_equal(dy.read(np.uint32, n)[r0:r1], ref_z[r0:r1])


def test_pagerank_in_ufixed_matches_the_integer_oracle(gpu):
    """PageRank::pull (app/pagerank.h:80-90) in the reference's shipped arithmetic, val_t = ap_ufixed<32,8,AP_RND,AP_SAT>
    (global.h:63): the out-degree-normalised matrix times the damping factor converted to value words
    (csr_matrix_convert_from_float, io/data_loader.h:75-84), rank = 1/n, 10 x { SpMV (+,x); eWiseAdd teleport } -- every word
    of every iteration equal to the integer oracle's."""
    m = named_matrix("uniform_10K_10")
    io.util_round_csr_matrix_dim(m, 128, 128)
    io.util_normalize_csr_matrix_by_outdegree(m)
    vt, n, damping = capi.GL_VAL_UFIXED_32_8, m.num_rows, np.float32(0.9)
    data = capi.words_from_float(vt, (m.adj_data * damping).astype(np.float32))
    rank = capi.words_from_float(vt, np.full(n, np.float32(1.0 / n), np.float32))
    teleport = int(capi.words_from_float(vt, np.array([np.float32(np.float32(1) - damping) / np.float32(n)], np.float32))[0])
    assert teleport > 0 and rank[0] > 100                                   # the words carry the values (24 fraction bits)
    plan = capi.SpMVPlan(n, n, m.adj_indptr, m.adj_indices, data.view(np.float32))
    dv, dr = capi.DeviceBuffer.from_host(rank), capi.DeviceBuffer(4 * n)
    ref = rank.copy()
    for it in range(10):
        plan.run_typed(dv, None, dr, 0, 0, 0, vt)
        capi.ewise_add_typed(dr, dv, n, teleport, vt)
        ref = O.ewise_add_words(vt, O.spmv_words(m.adj_indptr, m.adj_indices, data, ref, 0, vt, 0), teleport)
        got = dv.read(np.uint32, n)
        assert np.array_equal(got, ref), "iteration %d: %d words differ" % (it, int((got != ref).sum()))
    assert len(np.unique(ref)) > 50


@pytest.mark.parametrize("mask_name", list(MASKS))
@pytest.mark.parametrize("sem", list(OPS))
@pytest.mark.parametrize("vt_name", list(VT))
def test_spmspv_words(gpu, vt_name, sem, mask_name):
    vt, op, mt = VT[vt_name], OPS[sem], MASKS[mask_name]
    m = named_matrix("gplus_small")
    io.util_round_csr_matrix_dim(m, 128, 128)
    rng = np.random.default_rng(hash((vt_name, sem)) % (1 << 31))
    csc = io.csr2csc(m)
    data = _values(vt, op, rng, csc.nnz, "matrix")
    zero = _zero_word(vt, op)
    idx = np.sort(rng.choice(m.num_cols, size=m.num_cols // 50, replace=False)).astype(np.uint32)
    v = np.zeros(idx.shape[0] + 1, dtype=capi.IDX_WORD)
    v["index"][0] = idx.shape[0]
    v["index"][1:] = idx
    v["val"][1:] = _values(vt, op, rng, idx.shape[0], "vector")
    mask = np.where(rng.random(m.num_rows) < 0.5, np.uint32(zero), np.uint32(7)).astype(np.uint32)   # compared with `zero`
    plan = capi.SpMSpVPlan(csc.num_rows, csc.num_cols, csc.adj_indptr, csc.adj_indices, data.view(np.float32))
    dv = capi.DeviceBuffer(8 * (m.num_cols + 1))
    dv.write(v)
    dm, dr = capi.DeviceBuffer.from_host(mask), capi.DeviceBuffer(8 * (m.num_rows + 1))
    for rep in range(2):                                                # the second run starts from the reset accumulator
        plan.run_typed(dv, dm if mt else None, dr, op, zero, mt, vt)
        res = dr.read(capi.IDX_WORD, m.num_rows + 1)
        nnz = int(res["index"][0])
        assert res["val"][0] == zero
        got = np.full(m.num_rows, zero, dtype=np.uint32)
        got[res["index"][1:nnz + 1]] = res["val"][1:nnz + 1]
        assert np.all(np.diff(res["index"][1:nnz + 1].astype(np.int64)) > 0)
        ref = O.spmspv_words(csc.adj_indptr, csc.adj_indices, data, v, m.num_rows, op, vt, zero, mask if mt else None, mt)
        assert np.array_equal(got, ref), "%s %s %s run %d: %d rows differ" % (vt_name, sem, mask_name, rep, int((got != ref).sum()))
        assert not np.any(res["val"][1:nnz + 1] == zero)
    if mt != 1:      # (with a random mask half of the rows survive; with none, most)
        assert (ref != zero).sum() > 100


@pytest.mark.parametrize("vt_name", list(VT))
def test_apply_ops_words(gpu, vt_name):
    vt = VT[vt_name]
    rng = np.random.default_rng(5)
    n = 100000
    a = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    val = 0x40000000
    da, db = capi.DeviceBuffer.from_host(a), capi.DeviceBuffer(4 * n)
    capi.ewise_add_typed(da, db, n, val, vt)
    ref = O.ewise_add_words(vt, a, val)
    assert np.array_equal(db.read(np.uint32, n), ref)
    if vt == capi.GL_VAL_UFIXED_32_8:
        assert (ref == 0xffffffff).sum() > n // 8                       # saturated
    else:
        assert (ref < a).sum() > n // 8                                 # wrapped
    mask = rng.integers(0, 3, size=n).astype(np.uint32)
    mask[mask == 2] = 0x80000000
    for mt in (1, 2):
        inout = a.copy()
        dm, di = capi.DeviceBuffer.from_host(mask), capi.DeviceBuffer.from_host(inout)
        capi.assign_dense_typed(dm, di, n, 0xfffffffe, mt, vt)
        O.assign_dense_words(mt, mask, inout, 0xfffffffe)
        assert np.array_equal(di.read(np.uint32, n), inout)
    # SSSP-mode sparse assign: relax where the stored word is larger (unsigned order, incl. words above 2^31)
    idx = np.sort(rng.choice(n, size=n // 10, replace=False)).astype(np.uint32)
    sv = np.zeros(idx.shape[0] + 1, dtype=capi.IDX_WORD)
    sv["index"][0] = idx.shape[0]
    sv["index"][1:] = idx
    sv["val"][1:] = rng.integers(0, 1 << 32, size=idx.shape[0], dtype=np.uint64).astype(np.uint32)
    inout = a.copy()
    dsv, di, dn = capi.DeviceBuffer(8 * (n + 1)), capi.DeviceBuffer.from_host(inout), capi.DeviceBuffer(8 * (n + 1))
    dsv.write(sv)
    capi.assign_sparse_new_frontier_typed(dsv, di, dn, n, vt)
    nf_ref = O.assign_sparse_new_frontier_words(sv, inout)
    nf = dn.read(capi.IDX_WORD, n + 1)
    k = int(nf["index"][0])
    assert k == int(nf_ref["index"][0]) and k > n // 40
    assert np.array_equal(nf[1:k + 1], nf_ref[1:k + 1]) and np.array_equal(di.read(np.uint32, n), inout)
    # BFS-mode sparse assign and sparse -> dense move words untouched
    capi.assign_sparse_typed(dsv, di, 0xffffffff, n)
    inout[idx] = 0xffffffff
    assert np.array_equal(di.read(np.uint32, n), inout)
    dd = capi.DeviceBuffer(4 * n)
    capi.sparse_to_dense_typed(dsv, dd, n, 0xffffffff, n)
    dense = np.full(n, 0xffffffff, dtype=np.uint32)
    dense[idx] = sv["val"][1:]
    assert np.array_equal(dd.read(np.uint32, n), dense)


def test_word_conversions_follow_the_value_types():
    """csr_matrix_convert_from_float<val_t> (io/data_loader.h:75-84): AP_RND rounds half up, AP_SAT clamps; unsigned truncates."""
    f = np.array([0.0, 1.0, 1.5, 2.0 ** -24, 2.0 ** -25, 0.75 * 2.0 ** -24, 255.0, 256.0, 1e9, -3.0, 7.9], dtype=np.float32)
    w = capi.words_from_float(capi.GL_VAL_UFIXED_32_8, f)
    assert w.tolist() == [0, 1 << 24, 3 << 23, 1, 1, 1, 255 << 24, 0xffffffff, 0xffffffff, 0, 132540008]
    assert np.array_equal(w, O.words_from_float(O.VAL_UFIXED_32_8, f))
    u = capi.words_from_float(capi.GL_VAL_UNSIGNED, f)
    assert u.tolist() == [0, 1, 1, 0, 0, 0, 255, 256, 1000000000, 0, 7]
    assert np.array_equal(u, O.words_from_float(O.VAL_UNSIGNED, f))
    assert capi.words_to_float(capi.GL_VAL_UFIXED_32_8, [3 << 23, 0xffffffff]).tolist() == [1.5, 256.0]
