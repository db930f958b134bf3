cd $GRAFT_REPO_ROOT
rm -f gpurun_out/format_seconds.jsonl
python -m pytest tests/test_gpu_format.py -m gpu -x -q 2>&1 | tail -30
python - <<'PY'
import sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from graphlily_amd import datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/orkut.npz", A, compressed=False)
PY
cat /sys/kernel/mm/transparent_hugepage/enabled
build/api_breakdown /tmp/orkut.npz 4 2>&1 | grep -v amdgpu | tee gpurun_out/r02_api_breakdown.txt
python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank --npz /tmp/orkut.npz 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_reference_benches_orkut.txt
