"""CPU suite, part 2: the C-ABI library loads, exports every symbol include/graphlily_hip.h declares,
its host-only entry points work, and compute entry points fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from graphlily_amd import capi, io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "graphlily_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(L, sym), "libgraphlily_hip.so does not export %s" % sym
    assert sorted(capi.EXPORTS) == declared


def test_npz_loader_on_reference_fixtures(golden_dir):
    m = io.load_csr_matrix_from_float_npz(os.path.join(golden_dir, "eye_10_csr_float32.npz"))
    assert (m.num_rows, m.num_cols) == (10, 10)
    assert m.adj_data.tolist() == [1.0] * 10
    assert m.adj_indices.tolist() == list(range(10))
    assert m.adj_indptr.tolist() == list(range(11))
    m = io.load_csr_matrix_from_float_npz(os.path.join(golden_dir, "line_8_csr_float32.npz"))
    assert (m.num_rows, m.num_cols) == (8, 8)
    assert m.adj_indptr.tolist() == [0, 0, 1, 2, 3, 4, 5, 6, 7]
    assert m.adj_indices.tolist() == list(range(7))


def test_npz_loader_roundtrip_scipy(tmp_path):
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    A = sp.random(300, 211, density=0.05, format="csr", dtype=np.float32, random_state=rng)
    for compressed in (True, False):
        p = str(tmp_path / ("m%d.npz" % compressed))
        sp.save_npz(p, A, compressed=compressed)
        m = io.load_csr_matrix_from_float_npz(p)
        assert (m.num_rows, m.num_cols) == A.shape
        assert np.array_equal(m.adj_data, A.data)
        assert np.array_equal(m.adj_indices, A.indices.astype(np.uint32))
        assert np.array_equal(m.adj_indptr, A.indptr.astype(np.uint32))


def test_npz_loader_errors(tmp_path):
    with pytest.raises(capi.GraphLilyError) as e:
        io.load_csr_matrix_from_float_npz(str(tmp_path / "missing.npz"))
    assert e.value.code == capi.GL_ERR_IO
    bad = tmp_path / "bad.npz"
    bad.write_bytes(b"not a zip archive at all, definitely" * 4)
    with pytest.raises(capi.GraphLilyError):
        io.load_csr_matrix_from_float_npz(str(bad))


@pytest.mark.skipif(capi.device_count() > 0, reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(capi.GraphLilyError):
        capi.init(0)
    L = capi.lib()
    rc = L.gl_ewise_add(None, None, 8, 1.0)
    assert rc == capi.GL_ERR_NOT_INITIALIZED
    assert b"gl_init" in L.gl_last_error()
    h = ctypes.c_void_p(0)
    ip = np.zeros(2, np.uint32)
    rc = L.gl_spmv_plan_create(ctypes.byref(h), 1, 1, ip.ctypes.data, None, None, 0, 1)
    assert rc == capi.GL_ERR_NOT_INITIALIZED
