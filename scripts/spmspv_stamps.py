"""Scratch: phase stamps of the SpMSpV bin / fold kernels in blocking calls (a -DGL_STAMPS build of the library:
bash scripts/build_variant.sh WORK stamps -DGL_STAMPS; GRAPHLILY_HIP_LIB=scripts/_variants/stamps.so python scripts/spmspv_stamps.py).
Per workgroup, 10 ns ticks relative to the earliest stamp of the launch; printed: per-phase min / median / max over workgroups."""
import argparse, ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io, module as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="hollywood")
ap.add_argument("--sparsity", type=float, nargs="*", default=[0.9, 0.99, 0.9995])
ap.add_argument("--op", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
capi.init(0)
capi.set_stream(torch.cuda.current_stream().cuda_stream)
m = datasets.paper_graph(args.graph, 1.0, device=dev)
io.util_round_csr_matrix_dim(m, 128, 128)
csc = io.csr2csc(m)
plan = capi.SpMSpVPlan(csc.num_rows, csc.num_cols, csc.adj_indptr, csc.adj_indices, csc.adj_data)
n = csc.num_cols
mask = torch.zeros(n, device=dev)
res = torch.zeros(n + 1, dtype=torch.int64, device=dev)
bm, br = capi.DeviceBuffer.from_torch(mask), capi.DeviceBuffer.from_torch(res)
L = ctypes.CDLL(capi.LIB_PATH)
BIN = ["start", "sums out", "rendezvous", "-", "staged", "owner tab", "stream in", "histogram", "reserved", "stored", "end",
       "L owner tab", "L stream in", "L histogram", "L reserved", "L stored"]    # (L: the range's last batch when it has several)
FOLD = ["start", "cleared", "accumulated", "count", "fronts seen", "written"]
for sp in args.sparsity:
    cnt = max(1, int(n * (1 - sp)))
    idx = np.arange(cnt, dtype=np.uint32) * (n // cnt)
    v = M.make_sparse_vec(idx, np.ones(cnt, np.float32))
    bv = capi.DeviceBuffer.from_torch(torch.from_numpy(v.view(np.int64).copy()).to(dev))
    ts = []
    for it in range(12):
        time.sleep(0.002)
        t0 = time.perf_counter()
        plan.run(bv, bm, br, args.op, 0.0, 0)
        plan.wait()
        ts.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    st = np.zeros((2, 256, 16), np.uint64)
    assert L.gl_debug_stamps(st.ctypes.data_as(ctypes.c_void_p)) == 0
    st = st.astype(np.int64)
    os.makedirs("gpurun_out", exist_ok=True)
    np.save("gpurun_out/stamps_%s_%g.npy" % (args.graph, sp), st)
    print("== %s sparsity %.4f: %d entries, blocking call median %.1f us" % (args.graph, sp, cnt, float(np.median(ts))), flush=True)
    base = st[0][:, 0][st[0][:, 0] > 0].min()
    for k, names in ((0, BIN), (1, FOLD)):
        for p, name in enumerate(names):
            col = st[k][:, p]
            live = col[col >= base]          # stamps of this launch (older launches left smaller ones)
            if live.size == 0:
                continue
            rel = (live - base) / 100.0
            print("  %s %-12s wgs %3d  min %6.2f  med %6.2f  max %6.2f us" % ("bin " if k == 0 else "fold", name, live.size, rel.min(), float(np.median(rel)), rel.max()))
