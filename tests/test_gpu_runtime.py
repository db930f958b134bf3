"""GPU tests of the runtime plumbing (csrc/gl_runtime.hip): the device block pool behind gl_buf_alloc / gl_buf_free, which
replaces the per-call cl::Buffer construction of the reference's modules (module/spmv_module.h:424-439,
app/bfs.h:107-113), and the drivers' behaviour when a matrix is sent again (BaseModule::set_up_runtime +
send_matrix_host_to_device may be called more than once, app/bfs.h:100-103)."""
import numpy as np
import pytest

from graphlily_amd import app, capi, datasets, io
from oracle import oracle as O

from helpers import to_oracle

pytestmark = pytest.mark.gpu


def _pattern(n, seed):
    return np.random.default_rng(seed).integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)


def test_pool_recycled_block_survives_trim(gpu):
    """alloc, free, alloc the same size (a cache hit on a slab-carved block), gl_pool_trim, then use the buffer: the
    round-2 pool lost count of the re-issued block and released the slab under it."""
    import gc
    gc.collect()                           # buffers of earlier tests that nobody holds any more
    capi.pool_trim()
    a = capi.DeviceBuffer(1 << 20)
    pa = a.ptr
    a.free()
    b = capi.DeviceBuffer(1 << 20)
    # (in a fresh process b IS the freed block; after other tests of the process the pool may serve another parked block of
    # the size, or carve the slab again from its start once it has gone idle -- what matters is what follows)
    assert b.ptr != 0 and (b.ptr == pa or capi.pool_stats()[2] >= 1)
    keep = _pattern(1 << 18, 1)
    b.write(keep)
    capi.pool_trim()                       # b is live: its slab must stay
    live, cached, slabs = capi.pool_stats()
    assert slabs >= 1 and live >= 1
    # allocations after the trim must not land on b
    others = [capi.DeviceBuffer(1 << 20) for _ in range(8)]
    for i, o in enumerate(others):
        assert o.ptr != b.ptr
        o.write(_pattern(1 << 18, 100 + i))
    assert np.array_equal(b.read(np.uint32, 1 << 18), keep)
    for o in others:
        o.free()
    b.free()
    capi.pool_trim()


def test_pool_best_fit_and_slab_reset(gpu):
    import gc
    gc.collect()                           # buffers of earlier tests that nobody holds any more
    capi.pool_trim()
    live0, _, _ = capi.pool_stats()
    # a 1.00 MB request is served by a parked 1.125 MB block (<= 25 % larger), not by a 2 MB one
    big, mid = capi.DeviceBuffer(2 << 20), capi.DeviceBuffer((1 << 20) + (1 << 17))
    pmid, pbig = mid.ptr, big.ptr
    anchor = capi.DeviceBuffer(4096)       # keeps the slab alive, so the two stay parked
    big.free()
    mid.free()
    c = capi.DeviceBuffer(1 << 20)
    assert c.ptr == pmid
    d = capi.DeviceBuffer(1 << 20)
    assert d.ptr not in (pmid, pbig), "a 2 MB block is more than 25 % too large: fresh space instead"
    # many distinct sizes, freed: once the slab is idle its space is carved from the start again (no monotonic growth)
    sizes = [(3 + k) << 12 for k in range(200)]       # 84 MB: one slab
    bufs = [capi.DeviceBuffer(sz) for sz in sizes]
    first = min(b.ptr for b in bufs)
    for b in bufs + [c, d, anchor]:
        b.free()
    live, cached, slabs = capi.pool_stats()
    assert live == live0
    if live0 == 0:                         # (a block of some earlier test still out would keep its slab from going idle)
        e = capi.DeviceBuffer(7 << 12)
        assert e.ptr <= first, "the idle slab is reused from its start"
        e.free()
    capi.pool_trim()
    assert capi.pool_stats()[1] == 0


def test_pool_large_blocks_bypass_slabs_and_recycle(gpu):
    capi.pool_trim()
    a = capi.DeviceBuffer(80 << 20)        # above a quarter slab: its own allocation
    p = a.ptr
    a.write(_pattern(1024, 3))
    a.free()
    b = capi.DeviceBuffer(80 << 20)
    assert b.ptr == p
    b.free()
    capi.pool_trim()
    assert capi.pool_stats()[1] == 0


def test_bfs_matrix_sent_again_rebuilds_the_device_schedules(gpu):
    """BFS.send_matrix_host_to_device on an object that has already run: the captured hipGraphs and bit-vector buffers of
    the first matrix hold its plans' device pointers and its n -- they must be dropped, not replayed."""
    g1 = datasets.rmat(20000, 300000, seed=21, symmetric=True)
    g2 = datasets.rmat(33000, 500000, seed=22, symmetric=True)
    bfs = app.BFS(16, 0, 0, 0)
    bfs.set_up_runtime()
    for g in (g1, g2, g1):
        bfs.load_and_format_matrix(g, True)
        bfs.send_matrix_host_to_device()
        ref_m = g.copy()
        io.util_round_csr_matrix_dim(ref_m, 128, 128)
        ref_m.adj_data = np.ones(ref_m.nnz, np.float32)
        deg = np.diff(g.adj_indptr.astype(np.int64))
        src = int(np.argmax(deg > 0))
        ref = O.bfs(to_oracle(ref_m), src, 6)
        for _ in range(3):                 # eager, capture, replay
            assert np.array_equal(bfs.pull_push(src, 6, 0.01), ref)
            assert np.array_equal(bfs.pull(src, 6), ref)


def test_d2h_levels_is_exact_whatever_the_buffer_holds(gpu):
    """gl_buf_d2h_levels: packed when every value is a level that fits (nibbles up to 15, bytes up to 255), the floats
    themselves otherwise -- a fraction, a negative, a NaN, a value above the promised maximum, -0.0 (a different word than 0.0): the words that come back are the buffer's."""
    n = 1 << 18
    rng = np.random.default_rng(11)
    cases = {
        "nibbles": (rng.integers(0, 16, size=n).astype(np.float32), 15.0, True),
        "bytes": (rng.integers(0, 256, size=n).astype(np.float32), 255.0, True),
        "bytes promised, nibbles held": (rng.integers(0, 8, size=n).astype(np.float32), 200.0, True),
        "above the promise": (np.where(np.arange(n) == 777, 16.0, 3.0).astype(np.float32), 15.0, False),
        "a fraction": (np.where(np.arange(n) == n - 1, 2.5, 2.0).astype(np.float32), 15.0, False),
        "a negative": (np.where(np.arange(n) == 5, -1.0, 1.0).astype(np.float32), 255.0, False),
        "a NaN": (np.where(np.arange(n) == 12345, np.nan, 1.0).astype(np.float32), 255.0, False),
        "minus zero": (np.where(np.arange(n) == 9, -0.0, 0.0).astype(np.float32), 15.0, False),
        "too many levels": (rng.integers(0, 300, size=n).astype(np.float32), 299.0, False),
    }
    for name, (vals, mx, want_packed) in cases.items():
        buf = capi.DeviceBuffer.from_host(vals)
        out = np.full(n, -7.0, np.float32)
        packed = capi.d2h_levels(out, buf, n, mx)
        assert packed == (want_packed and capi.host_unpack_threads() >= 4), name
        assert np.array_equal(out.view(np.uint32), vals.view(np.uint32)), name      # (bit for bit either way)
    small = capi.DeviceBuffer.from_host(np.arange(1000, dtype=np.float32) % 7)
    out = np.zeros(1000, np.float32)
    assert capi.d2h_levels(out, small, 1000, 15.0) is False and np.array_equal(out, np.arange(1000, dtype=np.float32) % 7)


def test_streamed_level_read_back(gpu):
    """gl_levels_pack_stream / gl_sync_levels_unpack_stream (include/graphlily_hip.h): the pack kernel stores nibbles / bytes
    straight into a page-locked host block, a flag behind every chunk of 2048 words, and the host threads expand the chunks as
    they land -- the floats and the tail words that come back are the buffer's, for sizes below, at and across chunk boundaries,
    run after run on one block (re-armed each time); a block whose pack was never enqueued is reported, not waited for."""
    rng = np.random.default_rng(21)
    tw = 48
    tail_h = rng.integers(0, 1 << 32, size=tw, dtype=np.uint64).astype(np.uint32)
    tail = capi.DeviceBuffer.from_host(tail_h)
    for bits in (4, 8):
        per_word = 32 // bits
        for n in (8, 2048 * per_word - 8, 2048 * per_word, 2048 * per_word + 8, 600064, 3 * (1 << 20) + 24):
            block = capi.pinned_empty(capi.levels_stream_bytes(n, bits, tw), np.uint8)
            for rep in range(3):
                lev = rng.integers(0, 1 << bits, size=n).astype(np.float32)
                buf = capi.DeviceBuffer.from_host(lev)
                out = capi.pinned_empty(n + 4, np.float32)
                out[:] = -3.0
                capi.levels_stream_arm(block, n, bits, tw)
                capi.levels_pack_stream(buf, n, bits, tail, tw, block)
                got_tail = capi.sync_levels_unpack_stream(out, block, n, bits, tw)
                assert np.array_equal(out[:n], lev), (bits, n, rep)
                assert np.all(out[n:] == -3.0) and np.array_equal(got_tail, tail_h), (bits, n, rep)
    # an unaligned destination takes the plain loop
    n = 600064
    lev = rng.integers(0, 16, size=n).astype(np.float32)
    block = capi.pinned_empty(capi.levels_stream_bytes(n, 4, tw), np.uint8)
    out = np.zeros(n + 1, np.float32)[1:]
    capi.levels_stream_arm(block, n, 4, tw)
    capi.levels_pack_stream(capi.DeviceBuffer.from_host(lev), n, 4, tail, tw, block)
    capi.sync_levels_unpack_stream(out, block, n, 4, tw)
    assert np.array_equal(out, lev)
    # armed, nothing enqueued: the stream drains, the flags stay down
    capi.levels_stream_arm(block, n, 4, tw)
    with pytest.raises(capi.GraphLilyError, match="without delivering"):
        capi.sync_levels_unpack_stream(out, block, n, 4, tw)
    # pageable memory is refused by the pack
    with pytest.raises(capi.GraphLilyError, match="page-locked"):
        capi.levels_pack_stream(capi.DeviceBuffer.from_host(lev), n, 4, tail, tw, np.zeros(capi.levels_stream_bytes(n, 4, tw), np.uint8))


def test_gl_init_binds_the_calling_thread_to_the_devices_numa_node(gpu):
    """gl_host_bind_near_device (called by gl_init): the calling thread's affinity becomes the CPUs of the GPU's NUMA node that the
    process was allowed before -- or stays as it was when the platform names no node or the mask already lies inside it;
    GRAPHLILY_BIND_NUMA=0 leaves it alone either way.  (Fresh processes: this one is bound already.)"""
    import os
    import subprocess
    import sys
    prog = r'''
import os, sys
sys.path.insert(0, %r)
before = os.sched_getaffinity(0)
from graphlily_amd import capi
capi.init(0)
after = os.sched_getaffinity(0)
node, cpus = capi.host_bind_near_device()       # (again: same answer, or "nothing to do" once the mask lies inside the node)
again = os.sched_getaffinity(0)
want = None
import torch
pr = torch.cuda.get_device_properties(0)
try:
    d = "/sys/bus/pci/devices/%%04x:%%02x:%%02x.0" %% (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    nd = int(open(d + "/numa_node").read())
    if nd >= 0:
        want = set()
        for part in open("/sys/devices/system/node/node%%d/cpulist" %% nd).read().strip().split(","):
            a, _, b = part.partition("-")
            want.update(range(int(a), int(b or a) + 1))
        want &= before
except OSError:
    pass
print(len(before), len(after), len(again), -1 if want is None else len(want), int(want is not None and after == want), node, cpus)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for knob in ("1", "0"):
        env = dict(os.environ, GRAPHLILY_BIND_NUMA=knob)
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[knob] = [int(v) for v in r.stdout.strip().splitlines()[-1].split()]
    before, after, again, want, equal, node, cpus = outs["1"]
    if want > 0 and want < before:
        assert equal == 1 and after == want and again == want, outs
    else:
        assert after == before, outs
    b0, a0 = outs["0"][0], outs["0"][1]
    assert a0 == b0, outs        # the knob: gl_init does not touch the affinity
