import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from graphlily_amd import capi
capi.init(0)
n = 3072512
src = capi.pinned_empty(n // 2, np.uint8); src[:] = 0x21
dst = capi.pinned_empty(n, np.float32)
for pause in (0.0, 0.0005):
    ts = []
    for _ in range(200):
        if pause: time.sleep(pause)
        capi.host_threads_warm()
        t0 = time.perf_counter(); capi.host_levels_unpack(dst, src, n, 4); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("pause %.1f ms: median %.1f  p10 %.1f  p90 %.1f  max %.1f us" % (pause * 1e3, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), ts.max()))
