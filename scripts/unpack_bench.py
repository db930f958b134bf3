"""scratch (round 6): the host half of a packed BFS read-back alone -- gl_host_levels_unpack of orkut's 3 M nibbles into two alternating
result arrays (what a caller's loop leaves: the previous result is dropped after the next one exists), per store mode
(GRAPHLILY_HOST_STORES=stream|cached|clzero) and team size (GRAPHLILY_HOST_THREADS).  usage: python scripts/unpack_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi  # noqa: E402

capi.init(0)        # (binds this thread to the GPU's NUMA node)
n = 3072448
src = capi.pinned_empty(n // 2, np.uint8)
src[:] = np.random.default_rng(0).integers(0, 256, size=n // 2, dtype=np.uint8)
bufs = [capi.pinned_empty(n, np.float32) for _ in range(2)]
ts = []
for i in range(202):
    d = bufs[i & 1]
    t0 = time.perf_counter()
    capi.host_levels_unpack(d, src, n, 4)
    ts.append((time.perf_counter() - t0) * 1e6)
ref = np.empty(n, np.float32)
ref[0::2] = src & 15
ref[1::2] = src >> 4
assert np.array_equal(bufs[0], ref) and np.array_equal(bufs[1], ref)
print("stores %-7s threads %3d: median %6.1f us  p10 %6.1f  min %6.1f  (12 MB of floats: %.0f GB/s at the median)" % (
    os.environ.get("GRAPHLILY_HOST_STORES", "default"), capi.host_unpack_threads(), np.median(ts[20:]), np.percentile(ts[20:], 10), min(ts),
    4 * n / np.median(ts[20:]) / 1e3), flush=True)
