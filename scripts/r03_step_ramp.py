"""Per-step GPU time of the bench's SpMV step right after start-up (torch events around every step): is the slower start of
a short timed region (bench.py --steps 20) a ramp or a few outliers?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io
dev = torch.device("cuda:0")
capi.init(0); capi.set_stream(torch.cuda.current_stream().cuda_stream)
csr = datasets.paper_graph("orkut", 1.0, device=dev)
csr.adj_data = np.full(csr.nnz, np.float32(1.0 / csr.num_rows), dtype=np.float32)
io.util_round_csr_matrix_dim(csr, 128, 8)
plan = capi.SpMVPlan(csr.num_rows, csr.num_cols, csr.adj_indptr, csr.adj_indices, csr.adj_data, flags=capi.GL_PLAN_KEEP_VALUES)
x = torch.randint(0, 2, (csr.num_cols,), device=dev).float(); y = torch.zeros(max(csr.num_rows, csr.num_cols), device=dev)
bx, by = capi.DeviceBuffer.from_torch(x), capi.DeviceBuffer.from_torch(y)
K = 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(K):
    plan.run(bx, None, by, capi.GL_OP_MULADD, 0.0, capi.GL_NOMASK)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
print("steps 0-9   :", " ".join("%.3f" % t for t in ts[:10]))
print("steps 10-29 : mean %.4f" % np.mean(ts[10:30]))
print("steps 30-59 : mean %.4f" % np.mean(ts[30:60]))
print("steps 60-119: mean %.4f" % np.mean(ts[60:]))
time.sleep(0.5)
ev[0].record()
for i in range(30):
    plan.run(bx, None, by, capi.GL_OP_MULADD, 0.0, capi.GL_NOMASK); ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(30)]
print("after a 0.5 s pause: first 5", " ".join("%.3f" % t for t in ts[:5]), "mean of 30 %.4f" % np.mean(ts))
