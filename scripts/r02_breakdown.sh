cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from graphlily_amd import datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/orkut.npz", A, compressed=False)
PY
build/api_breakdown /tmp/orkut.npz 5 2>&1 | tee gpurun_out/r02_api_breakdown.txt
python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank,sssp --npz /tmp/orkut.npz 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_reference_benches_orkut.txt
