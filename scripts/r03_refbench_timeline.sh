# The reference's own bench_bfs / bench_pagerank (unmodified, oracle/_ref/*_on_hip) on the orkut stand-in with the C ABI's
# call timeline switched on (GRAPHLILY_TRACE_API): what the driver does between its module calls.
cd $GRAFT_REPO_ROOT
G=${1:-orkut}
python - <<PY
import sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from graphlily_amd import datasets
m = datasets.paper_graph("$G", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/g.npz", A, compressed=False)
PY
IT=$(python -c "import sys; sys.path.insert(0,'.'); from graphlily_amd import datasets; print(datasets.PAPER_GRAPHS['$G']['iters'])")
echo "== bench_bfs"
GRAPHLILY_TRACE_API=gpurun_out/api_bfs.txt oracle/_ref/bench_bfs_on_hip 16 1024000 256000 30720 x /tmp/g.npz $IT 2>&1 | grep -v amdgpu.ids
echo "== bench_pagerank"
GRAPHLILY_TRACE_API=gpurun_out/api_pagerank.txt oracle/_ref/bench_pagerank_on_hip 16 1024000 30720 x /tmp/g.npz 2>&1 | grep -v amdgpu.ids
wc -l gpurun_out/api_bfs.txt gpurun_out/api_pagerank.txt
