"""Scratch: dump GRAPHLILY_SPMV_CLOCKS for one graph / variant and summarise it (scripts/unit_clocks.py)."""
import os, subprocess, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, var = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "base")
path = "/tmp/clocks_%s.txt" % name
os.environ["GRAPHLILY_SPMV_CLOCKS"] = path
from graphlily_amd import capi, datasets, io
dev = torch.device("cuda:0")
capi.init(0)
m = datasets.paper_graph(name, 1.0, device=dev)
m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32)
io.util_round_csr_matrix_dim(m, 128, 8)
if var != "base":
    for kv in var.split(","):
        k, v = kv.split("=")
        os.environ["GRAPHLILY_SPMV_" + k] = v
plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=capi.GL_PLAN_KEEP_VALUES)
x = torch.randint(0, 2, (m.num_cols,), device=dev).float()
y = torch.zeros(m.num_rows, device=dev)
bx, by = (capi.DeviceBuffer.from_torch(t) for t in (x, y))
for _ in range(5):
    plan.run(bx, None, by, 0, 0.0, 0)
capi.sync()
print("==", name, var, plan.info(), flush=True)
subprocess.call([sys.executable, os.path.join(os.path.dirname(__file__), "unit_clocks.py"), path])
if len(sys.argv) > 3:
    import shutil
    shutil.copy(path, sys.argv[3])
