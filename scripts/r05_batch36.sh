#!/bin/bash
# which SpMV kernel does the reference's bench_pagerank run on the C++ headers?
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
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prpr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prpr -- /root/repo/oracle/_ref/bench_pagerank_on_hip 16 1024000 30720 unused.xclbin /tmp/orkut_csr_float32.npz 2>&1 | grep "iteration"
f=$(find /tmp/prpr -name "*kernel_stats.csv" | head -1); grep "spmv_\|fill\|ewise" "$f" | cut -c1-200 > /root/repo/gpurun_out/r05_pagerank_cpp_kernels.txt
cat /root/repo/gpurun_out/r05_pagerank_cpp_kernels.txt
