"""Scratch perf probe: gl_spmspv_run on a stand-in over the reference's sparsity sweep
(benchmark/bench_spmspv.cpp:261-276), throughput by its definition (8 B x sum nnz(active columns) / t)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io, module as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="googleplus")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--op", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda:0")
capi.init(0)
capi.set_stream(torch.cuda.current_stream().cuda_stream)
m = datasets.paper_graph(args.graph, 1.0, device=dev)
io.util_round_csr_matrix_dim(m, 128, 128)
t0 = time.time()
csc = io.csr2csc(m)
t1 = time.time()
plan = capi.SpMSpVPlan(csc.num_rows, csc.num_cols, csc.adj_indptr, csc.adj_indices, csc.adj_data)
print("graph %s n=%d nnz=%d  csr2csc %.1fs plan %.1fs" % (args.graph, m.num_rows, m.nnz, t1 - t0, time.time() - t1), flush=True)
n = csc.num_cols
coldeg = np.diff(csc.adj_indptr.astype(np.int64))
mask = torch.randint(0, 2, (n,), device=dev).float()
res = torch.zeros(n + 1, dtype=torch.int64, device=dev)
bm, br = capi.DeviceBuffer.from_torch(mask), capi.DeviceBuffer.from_torch(res)
zero = 0.0 if args.op < 2 else 255.0
for sparsity in (0.5, 0.9, 0.99, 0.999, 0.9999):
    cnt = max(1, int(n * (1 - sparsity)))
    inc = n // cnt
    idx = np.arange(cnt, dtype=np.uint32) * inc
    v = M.make_sparse_vec(idx, np.ones(cnt, np.float32))
    tv = torch.from_numpy(v.view(np.int64).copy()).to(dev)
    bv = capi.DeviceBuffer.from_torch(tv)
    active = int(coldeg[idx].sum())
    for mt in (0, 1):
        for _ in range(3):
            plan.run(bv, bm, br, args.op, zero, mt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            plan.run(bv, bm, br, args.op, zero, mt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print("sparsity %.4f%% nnz(v)=%8d active nnz=%10d mask %d: %.4f ms  %.1f GB/s  %.2f GTEPS  out=%d" %
              (sparsity * 100, cnt, active, mt, ms, 8 * active / ms / 1e6, active / ms / 1e6, capi.sparse_nnz(br)), flush=True)
