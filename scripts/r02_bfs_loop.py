"""orkut BFS pull_push: host-driven loop vs the device-resident schedule (eager / graph, with / without the side stream)."""
import sys, time, numpy as np, torch, os
sys.path.insert(0, '.')
from graphlily_amd import app, capi, datasets
g = sys.argv[1] if len(sys.argv) > 1 else "orkut"
m = datasets.paper_graph(g, 1.0, device=torch.device("cuda:0"))
iters = datasets.PAPER_GRAPHS[g]["iters"]
t0 = time.time()
bfs = app.BFS(16, 0, 0, 0)
bfs.set_up_runtime(); bfs.load_and_format_matrix(m, True); bfs.send_matrix_host_to_device()
print("%s BFS setup %.2f s" % (g, time.time() - t0))
for mode, gr, ov in (("0", "0", "0"), ("1", "0", "0"), ("1", "1", "0"), ("1", "0", "1"), ("1", "1", "1")):
    os.environ["GRAPHLILY_BFS_DEVICE_LOOP"] = mode
    os.environ["GRAPHLILY_BFS_GRAPH"] = gr
    os.environ["GRAPHLILY_BFS_OVERLAP"] = ov
    if hasattr(bfs, "dev_loop_"):
        bfs.dev_loop_["graphs"].clear(); bfs.dev_loop_.pop("warm", None)
    ts = []
    for i in range(9):
        capi.sync(); t = time.perf_counter(); d = bfs.pull_push(0, iters, 0.001); ts.append((time.perf_counter() - t) * 1e3)
    print("device_loop=%s graph=%s overlap=%s: median %.3f ms (%s), pushes %d reached %d" % (
        mode, gr, ov, float(np.median(ts[3:])), " ".join("%.3f" % x for x in ts), bfs.push_iterations_, int((d != 0).sum())), "again", getattr(bfs, "push_iterations_again_", None))
