#!/bin/bash
# how long does the 12 MB read-back at the end of a PageRank pull take from Python (np.empty destination) and from C++ (host-pool block)?
cd /root/repo; mkdir -p gpurun_out
GRAPHLILY_TRACE_API=gpurun_out/api_pagerank_py.txt timeout 600 python scripts/probe_pagerank.py orkut 2>&1 | grep -v amdgpu.ids | tail -1
grep "gl_buf_d2h " gpurun_out/api_pagerank_py.txt | tail -12 > gpurun_out/r05_d2h_python_vs_cpp.txt
echo "---- C++ (bench_pagerank, oracle/_ref)" >> gpurun_out/r05_d2h_python_vs_cpp.txt
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp, torch
from graphlily_amd import datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/orkut_csr_float32.npz", A, compressed=False)
PY
for i in 1 2 3; do GRAPHLILY_TRACE_API=gpurun_out/api_pagerank.txt oracle/_ref/bench_pagerank_on_hip 16 1024000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 2>&1 | grep iteration; grep "gl_buf_d2h \|gl_sync" gpurun_out/api_pagerank.txt | tail -3 >> gpurun_out/r05_d2h_python_vs_cpp.txt; done
cat gpurun_out/r05_d2h_python_vs_cpp.txt
