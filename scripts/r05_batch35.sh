#!/bin/bash
# round 5, batch 35: the C ABI call timeline of the reference's bench_pagerank (one timed pull after one warm-up)
cd /root/repo; mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp, torch
from graphlily_amd import datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/orkut_csr_float32.npz", A, compressed=False)
PY
GRAPHLILY_TRACE_API=gpurun_out/api_pagerank.txt oracle/_ref/bench_pagerank_on_hip 16 1024000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 2>&1 | grep -v amdgpu.ids
wc -l gpurun_out/api_pagerank.txt
tail -60 gpurun_out/api_pagerank.txt > gpurun_out/r05_api_pagerank_tail.txt
