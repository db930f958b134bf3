#!/bin/bash
# round 5, batch 33: executable graphs uploaded when they are instantiated (the reference's bench drivers time the FIRST replay) --
# the reference's bench_bfs / bench_sssp on orkut with the knob off / on, three times each, same box
cd /root/repo; mkdir -p gpurun_out
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp, torch
from graphlily_amd import datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/orkut_csr_float32.npz", A, compressed=False)
PY
for rep in 1 2 3; do for k in 0 1; do
echo -n "graph_upload=$k: "; GRAPHLILY_DEBUG=graph_upload=$k oracle/_ref/bench_bfs_on_hip 16 1024000 256000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 6 2>&1 | grep "average_time" | tr '\n' ' '
GRAPHLILY_DEBUG=graph_upload=$k oracle/_ref/bench_sssp_on_hip 16 1024000 256000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 6 2>&1 | grep "average_time" | tr '\n' ' '; echo
done; done | tee gpurun_out/r05_graph_upload.txt
timeout 900 python -m pytest tests/test_gpu_apps.py tests/test_cpp_layer.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
