cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from graphlily_amd import datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
sp.save_npz("/tmp/orkut.npz", A, compressed=False)
PY
export TMPDIR=/tmp
for v in "" "GRAPHLILY_POOL_SPARE=0" "GRAPHLILY_BLOCKING=1"; do
  echo "== env: $v"
  env $v oracle/_ref/bench_bfs_on_hip 16 1024000 256000 30720 x /tmp/orkut.npz 6 2>&1 | grep -v amdgpu.ids
done
cd /tmp && rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_bfs -- $GRAFT_REPO_ROOT/oracle/_ref/bench_bfs_on_hip 16 1024000 256000 30720 x /tmp/orkut.npz 6 2>&1 | grep -v amdgpu.ids | tail -8
cd $GRAFT_REPO_ROOT; find gpurun_out/trace_bfs -name "*.csv" | xargs ls -la
python tests/../scripts/r02_trace_summary.py gpurun_out/trace_bfs || true
