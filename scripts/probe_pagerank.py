"""Scratch: what a PageRank pull(0.9, 10) is made of -- the ten iterations on the device, and the final read-back into (a) a
recycled page-locked block (the default), (b) a fresh np.empty() per call (round 4's)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "orkut"
dev = torch.device("cuda:0")
capi.init(0)
raw = datasets.paper_graph(name, 1.0, device=dev)
pr = app.PageRank(16, 0, 0)
pr.set_up_runtime()
pr.load_and_format_matrix(raw, 0.9, True)
pr.send_matrix_host_to_device()
def med(fn, runs=9, warm=2):
    for _ in range(warm): fn()
    ts = []
    for _ in range(runs):
        capi.sync(); t0 = time.perf_counter(); out = fn(); capi.sync(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
t_rec = med(lambda: pr.pull(0.9, 10))
new = app.HipBackend.download_result
app.HipBackend.download_result = lambda self, buf, count: buf.read(np.float32, count)
t_fresh = med(lambda: pr.pull(0.9, 10))
app.HipBackend.download_result = lambda self, buf, count: None
t_dev = med(lambda: pr.pull(0.9, 10))
app.HipBackend.download_result = new
print("%s n=%d: pull(0.9, 10) %.3f ms recycled page-locked result, %.3f ms fresh np.empty result, %.3f ms without the read-back  (per iteration %.4f / %.4f / %.4f)"
      % (name, pr.n_, t_rec, t_fresh, t_dev, t_rec / 10, t_fresh / 10, t_dev / 10))
