# The reference's own bench_sssp (unmodified, oracle/_ref/bench_sssp_on_hip) on a stand-in with the C ABI's call timeline
# switched on (GRAPHLILY_TRACE_API): where SSSP::pull spends its 2.6 ms on orkut besides the six SpMV launches.
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
GRAPHLILY_TRACE_API=gpurun_out/api_sssp.txt oracle/_ref/bench_sssp_on_hip 16 1024000 256000 30720 x /tmp/g.npz $IT 2>&1 | grep -v amdgpu.ids
wc -l gpurun_out/api_sssp.txt; head -3 gpurun_out/api_sssp.txt
python - <<'PY'
lines = open("gpurun_out/api_sssp.txt").read().splitlines()
# the first timed pull: from the second send_vector-sized gl_buf_h2d after the matrix was sent to the following large d2h
print("\n".join(lines[:4]))
idx = [i for i, l in enumerate(lines) if "gl_spmv_run" in l]
print("... calls around the first pulls:")
lo = max(0, idx[0] - 6)
print("\n".join(lines[lo:lo + 60]))
PY
