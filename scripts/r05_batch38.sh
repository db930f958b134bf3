#!/bin/bash
# round 5, batch 38: gl_buf_d2h waits for the stream before it copies to pageable memory -- the reference's drivers, knob off / on
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
for rep in 1 2 3; do for k in 0 1; do
echo -n "d2h_presync=$k: "; GRAPHLILY_DEBUG=d2h_presync=$k oracle/_ref/bench_pagerank_on_hip 16 1024000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 2>&1 | grep "iteration" | tr '\n' ' '
GRAPHLILY_DEBUG=d2h_presync=$k oracle/_ref/bench_sssp_on_hip 16 1024000 256000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 6 2>&1 | grep "average_time" | tr '\n' ' '
GRAPHLILY_DEBUG=d2h_presync=$k oracle/_ref/bench_bfs_on_hip 16 1024000 256000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 6 2>&1 | grep "average_time" | tr '\n' ' '; echo
done; done | tee gpurun_out/r05_ab_d2h_presync.txt
