"""Scratch: what a PageRank pull(0.9, 10) is made of -- the ten iterations on the device with chained runs (gl_spmv_run_chained:
no packing pass in front of a run whose vector is the previous run's results) and with a packing pass per run, each with and
without the final 12 MB read-back."""
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
t_chain = med(lambda: pr.pull(0.9, 10))
r_chain = pr.pull(0.9, 10)
chained = app._GraphApp._run_chained
app._GraphApp._run_chained = lambda self, k: self.SpMV_.run()
t_plain = med(lambda: pr.pull(0.9, 10))
r_plain = pr.pull(0.9, 10)
new = app.HipBackend.download_result
app.HipBackend.download_result = lambda self, buf, count: None
t_plain_dev = med(lambda: pr.pull(0.9, 10))
app._GraphApp._run_chained = chained
t_chain_dev = med(lambda: pr.pull(0.9, 10))
app.HipBackend.download_result = new
print("%s n=%d: pull(0.9, 10) per iteration: %.4f ms chained runs (%.4f without the read-back), %.4f ms with a packing pass per run (%.4f); results bit-identical: %s"
      % (name, pr.n_, t_chain / 10, t_chain_dev / 10, t_plain / 10, t_plain_dev / 10, bool(np.array_equal(r_chain.view(np.uint32), r_plain.view(np.uint32)))))
