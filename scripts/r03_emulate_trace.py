"""rocprofv3 evidence for the row-sharded BFS schedule: one emulated rank k of N on the orkut stand-in.
Run under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv`; scripts/r03_emulate_trace_summary.py cuts out
the last replayed call and lists its kernels and copies.  usage: r03_emulate_trace.py [graph] [k/N] [mode]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets
from graphlily_amd.dist import EmulatedComm
g = sys.argv[1] if len(sys.argv) > 1 else "orkut"
k, N = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0/8").split("/"))
mode = sys.argv[3] if len(sys.argv) > 3 else "pull_push"
scale = float(os.environ.get("SCALE", "1.0"))
m = datasets.paper_graph(g, scale, device=torch.device("cuda:0"))
iters = datasets.PAPER_GRAPHS[g]["iters"]
deg = np.diff(m.adj_indptr.astype(np.int64))
src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
whole = app.BFS(16, 0, 0, 0)
whole.set_up_runtime(); whole.load_and_format_matrix(m, True); whole.send_matrix_host_to_device()
run_whole = (lambda: whole.pull_push(src, iters, 0.001)) if mode == "pull_push" else (lambda: whole.pull(src, iters))
ref = run_whole().copy()
comm = EmulatedComm(k, N)
b = app.BFS(16, 0, 0, 0, comm=comm)
b.set_up_runtime(); b.load_and_format_matrix(m, True); b.send_matrix_host_to_device()
b.gather_result_ = False
comm.set_truth(whole.bits_loop_["vecs"], whole.bits_loop_["words"])
fn = (lambda: b.pull_push(src, iters, 0.001)) if mode == "pull_push" else (lambda: b.pull(src, iters))
for i in range(5):
    run_whole()
    capi.sync()
    print("MARK whole-done %d" % i, flush=True)
    t0 = time.time_ns()
    d = fn()
    t1 = time.time_ns()
    r0, r1 = b.result_range_
    print("CALL %d %d %d rank %d/%d rows %d equal %s pushes %d slots %s" % (i, t0, t1, k, N, r1 - r0, bool(np.array_equal(d, ref[r0:r1])),
                                                                          b.push_iterations_, b.bfs_slot_counts_.tolist()), flush=True)
