"""Shape sweep for ONE emulated rank of the row-sharded BFS (one-launch slots): the shard's boolean plan forced to blocks x
segments, schedule time of pull_push and pull.  usage: r03_shard_shape_sweep.py [graph] [k/N] [BxS,BxS,...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets
from graphlily_amd.dist import EmulatedComm
g = sys.argv[1] if len(sys.argv) > 1 else "orkut"
k, N = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0/8").split("/"))
shapes = (sys.argv[3] if len(sys.argv) > 3 else "0x0,64x4,128x2,256x1").split(",")
m = datasets.paper_graph(g, 1.0, device=torch.device("cuda:0"))
iters = datasets.PAPER_GRAPHS[g]["iters"]
deg = np.diff(m.adj_indptr.astype(np.int64))
src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
whole = app.BFS(16, 0, 0, 0)
whole.set_up_runtime(); whole.load_and_format_matrix(m, True); whole.send_matrix_host_to_device()
for shape in shapes:
    B, S = (int(v) for v in shape.split("x"))
    if B:
        os.environ["GRAPHLILY_SPMV_BLOCKS"], os.environ["GRAPHLILY_SPMV_SEGMENTS"] = str(B), str(S)
    comm = EmulatedComm(k, N)
    b = app.BFS(16, 0, 0, 0, comm=comm)
    b.set_up_runtime(); b.load_and_format_matrix(m, True); b.send_matrix_host_to_device()
    os.environ.pop("GRAPHLILY_SPMV_BLOCKS", None); os.environ.pop("GRAPHLILY_SPMV_SEGMENTS", None)
    b.gather_result_ = False
    info = b.SpMV_.plan_.info()
    out = {"shape": shape, "plan": {kk: info[kk] for kk in ("blocks", "segments", "num_units")}}
    for mode in ("pull_push", "pull"):
        run_whole = (lambda: whole.pull_push(src, iters, 0.001)) if mode == "pull_push" else (lambda: whole.pull(src, iters))
        fn = (lambda: b.pull_push(src, iters, 0.001)) if mode == "pull_push" else (lambda: b.pull(src, iters))
        ref = run_whole().copy()
        comm.set_truth(whole.bits_loop_["vecs"], whole.bits_loop_["words"])
        b.time_schedule_ = False
        for _ in range(3):
            d = fn()
        r0, r1 = b.result_range_
        ok = bool(np.array_equal(d, ref[r0:r1]))
        b.time_schedule_ = True
        ts = []
        for _ in range(15):
            fn()
            ts.append(b.schedule_ms_)
        out[mode] = {"schedule_ms": round(float(np.median(ts)), 4), "ok": ok}
    print(out, flush=True)
    del b
