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


def test_npz_loader_survives_truncated_and_corrupted_archives(tmp_path, golden_dir):
    """Every length field of the zip / npy containers comes from the file; a damaged archive must be
    rejected with GL_ERR_IO (or load, if the damage missed everything that matters) -- never read past
    the file or allocate by a forged size.  Truncations at every 7th byte + seeded byte flips."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    A = sp.random(64, 64, density=0.1, format="csr", dtype=np.float32, random_state=rng)
    blobs = []
    for compressed in (True, False):
        q = str(tmp_path / ("src%d.npz" % compressed))
        sp.save_npz(q, A, compressed=compressed)
        blobs.append(open(q, "rb").read())
    blobs.append(open(os.path.join(golden_dir, "eye_10_csr_float32.npz"), "rb").read())
    victim = tmp_path / "victim.npz"
    outcomes = {"ok": 0, "rejected": 0}

    def attempt(data):
        victim.write_bytes(data)
        try:
            io.load_csr_matrix_from_float_npz(str(victim))
            outcomes["ok"] += 1
        except capi.GraphLilyError as e:
            assert e.code in (capi.GL_ERR_IO, capi.GL_ERR_INVALID_ARG)
            outcomes["rejected"] += 1

    for blob in blobs:
        for cut in range(0, len(blob), 7):
            attempt(blob[:cut])
        for _ in range(300):
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            attempt(bytes(b))
        # forged sizes: every 32-bit field of the central directory / end record set to 0xffffffff in turn
        cd = blob.rfind(b"PK\x01\x02")
        for off in range(cd, len(blob) - 3, 2):
            b = bytearray(blob)
            b[off:off + 4] = b"\xff\xff\xff\xff"
            attempt(bytes(b))
    assert outcomes["rejected"] > 500


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


@pytest.mark.skipif(capi.device_count() > 0, reason="checks the no-GPU failure mode")
def test_bfs_schedule_entry_points_fail_loudly_without_gpu():
    """The bit-frontier BFS steps and the asynchronous read-back are compute entry points: without a device they must
    refuse (GL_ERR_NOT_INITIALIZED), not pretend; the hints validate their handle; the host-side patch helper is gone."""
    L = capi.lib()
    assert L.gl_bfs_bits_begin(None, 32, None, 8, None, 4, 3, 0) == capi.GL_ERR_NOT_INITIALIZED
    assert L.gl_bfs_bits_push_step(None, None, None, None, None, 4, None, 2.0, None, 1, 0.5, 3) == capi.GL_ERR_NOT_INITIALIZED
    assert L.gl_bfs_bits_pull_step(None, None, None, None, None, 2.0, None, 1, 0.5, 3, 1.0) == capi.GL_ERR_NOT_INITIALIZED
    assert L.gl_buf_d2h_async(None, None, 16) == capi.GL_ERR_NOT_INITIALIZED
    assert L.gl_spmspv_plan_hint_work(None, 1, 1, 1) == capi.GL_ERR_INVALID_ARG
    assert not hasattr(L, "gl_host_patch_bits") and not hasattr(L, "gl_side_copy_d2h")


def test_host_helpers_need_no_gpu():
    """gl_host_sparse_to_dense (convert_sparse_vec_to_dense_vec, graphlily/global.h:153-164, on a few host threads), gl_host_fill_u32
    and gl_host_pool_reserve work without a device."""
    import ctypes
    L = capi.lib()
    rng = np.random.default_rng(3)
    n = 3_000_017
    for kind in ("ascending", "unordered_with_repeats", "empty"):
        if kind == "ascending":
            idx = np.sort(rng.choice(n, size=400_000, replace=False)).astype(np.uint32)
        elif kind == "unordered_with_repeats":
            idx = rng.integers(0, n, size=300_000).astype(np.uint32)
        else:
            idx = np.zeros(0, np.uint32)
        sv = np.zeros(idx.shape[0] + 1, dtype=capi.IDX_VAL)
        sv["index"][0] = idx.shape[0]
        sv["index"][1:] = idx
        sv["val"][1:] = rng.random(idx.shape[0], dtype=np.float32) + 1.0
        out = np.empty(n, np.float32)
        zero = np.float32(255.0)
        capi.check(L.gl_host_sparse_to_dense(sv.ctypes.data, n, int(zero.view(np.uint32)), out.ctypes.data))
        ref = np.full(n, zero, np.float32)
        for k in range(1, idx.shape[0] + 1) if kind != "ascending" else ():
            ref[sv["index"][k]] = sv["val"][k]          # sequential: the last store to an index wins
        if kind == "ascending":
            ref[idx] = sv["val"][1:]
        assert np.array_equal(out, ref), kind
    buf = np.empty(2_500_001, np.uint32)
    capi.check(L.gl_host_fill_u32(buf.ctypes.data, 0xdeadbeef, buf.shape[0]))
    assert np.all(buf == 0xdeadbeef)
    capi.check(L.gl_host_pool_reserve(5 << 20, 2))
    p = ctypes.c_void_p(0)
    capi.check(L.gl_host_pool_alloc(ctypes.byref(p), 5 << 20))
    assert p.value and p.value % (2 << 20) == 0, "large host blocks are 2 MB-aligned (huge-page advice)"
    capi.check(L.gl_host_pool_free(p))


def test_levels_stream_block_layout():
    """gl_levels_stream_bytes (needs no GPU): packed words, the tail on the next 16-byte boundary, one 64-byte flag line per chunk of
    GL_LEVELS_CHUNK_WORDS words plus one for the tail."""
    for n, bits, tw in ((8, 4, 0), (8, 8, 48), (2048 * 8, 4, 48), (2048 * 8 + 8, 4, 48), (3072448, 4, 64), (3072448, 8, 64)):
        words = n // (32 // bits)
        tail_at = (words + 3) & ~3
        flags_at = (tail_at + tw + 15) & ~15
        chunks = -(-words // 2048) + 1
        assert capi.levels_stream_bytes(n, bits, tw) == 4 * (flags_at + 16 * chunks), (n, bits, tw)
        assert tail_at == capi.levels_packed_words(n, bits)


def test_host_levels_unpack():
    """gl_host_levels_unpack (the host half of the BFS packed read-back): every byte / nibble value, sizes on both sides of
    the thread-count steps, nothing written past n."""
    rng = np.random.default_rng(9)
    for n in (0, 8, 256, 70000, (1 << 18) + 16, 3 << 20):
        for bits in (8, 4):
            lev = rng.integers(0, 1 << bits, size=n, dtype=np.uint8)
            if n >= 256:
                lev[:256] = np.arange(256, dtype=np.uint16) % (1 << bits)
            src = lev if bits == 8 else (lev[0::2] | (lev[1::2] << 4)).astype(np.uint8)
            dst = np.full(n + 3, -1.0, np.float32)
            capi.host_levels_unpack(dst, np.ascontiguousarray(src), n, bits)
            assert np.array_equal(dst[:n], lev.astype(np.float32)), (n, bits)
            assert np.all(dst[n:] == -1.0)
