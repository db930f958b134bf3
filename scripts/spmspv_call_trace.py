"""Where does a blocking SpMSpV call spend its time?  One graph, one sparsity, 60 blocking runs; run under
rocprofv3 --kernel-trace --stats.  usage: r03_spmspv_call_trace.py graph sparsity [semiring]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io
from graphlily_amd import module as M
g, sparsity = sys.argv[1], float(sys.argv[2])
sem = {"Arithmetic": M.ArithmeticSemiring, "Tropical": M.TropicalSemiring, "Logical": M.LogicalSemiring}[sys.argv[3] if len(sys.argv) > 3 else "Arithmetic"]
capi.init(0)
csr = datasets.paper_graph(g, 1.0, device=torch.device("cuda:0"))
csc = io.csr2csc(csr)
mod = M.SpMSpVModule(512 * 1024)
mod.set_semiring(sem); mod.set_mask_type(M.kNoMask); mod.set_up_runtime("x")
mod.load_and_format_matrix(csc); mod.send_matrix_host_to_device(); mod.enable_own_pull()
mod.send_mask_host_to_device(np.zeros(csc.num_rows, np.float32))
cnt = int(np.floor((1 - sparsity) * csc.num_cols))
idx = (np.arange(cnt, dtype=np.int64) * (csc.num_cols // cnt)).astype(np.uint32)
vals = ((np.random.default_rng(1).integers(0, 99, size=cnt) + 1) / 100.0).astype(np.float32)
mod.send_vector_host_to_device(M.make_sparse_vec(idx, vals))
for _ in range(5):
    mod.run()
ts = []
for _ in range(60):
    t0 = time.perf_counter(); mod.run(); ts.append(time.perf_counter() - t0)
print("blocking run: median %.1f us, min %.1f us; vector %d entries, results %d" % (np.median(ts) * 1e6, np.min(ts) * 1e6, cnt, mod.get_results_nnz()))
# the same without the wait: enqueue 60, wait once
mod.blocking = False
capi.sync(); t0 = time.perf_counter()
for _ in range(60):
    mod.run()
capi.sync(); print("enqueued back to back: %.1f us per run" % ((time.perf_counter() - t0) / 60 * 1e6))
