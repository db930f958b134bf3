"""BFS pull_push on a stand-in: host-driven loop vs the list-based device schedule vs the bit-frontier schedule
(eager / hipGraph), wall-clock per call."""
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
deg = np.diff(m.adj_indptr.astype(np.int64))
src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
ref = None
for mode, bits, gr in (("0", "0", "0"), ("1", "0", "1"), ("1", "1", "0"), ("1", "1", "1")):
    os.environ["GRAPHLILY_BFS_DEVICE_LOOP"] = mode
    os.environ["GRAPHLILY_BFS_BITS"] = bits
    os.environ["GRAPHLILY_BFS_GRAPH"] = gr
    ts = []
    for i in range(11):
        capi.sync(); t = time.perf_counter(); d = bfs.pull_push(src, iters, 0.001); ts.append((time.perf_counter() - t) * 1e3)
    if ref is None:
        ref = d.copy()
    print("%s device_loop=%s bits=%s graph=%s: median %.3f ms (%s), pushes %d again %s reached %d same %s" % (
        g, mode, bits, gr, float(np.median(ts[3:])), " ".join("%.3f" % x for x in ts), bfs.push_iterations_,
        getattr(bfs, "push_iterations_again_", None), int((d != 0).sum()), bool(np.array_equal(d, ref))))
for bits, gr in (("0", "1"), ("1", "1"), ("1", "0"), ("0", "1"), ("1", "1"), ("1", "0")):
    os.environ["GRAPHLILY_BFS_BITS"] = bits
    os.environ["GRAPHLILY_BFS_GRAPH"] = gr
    ts = []
    for i in range(9):
        capi.sync(); t = time.perf_counter(); d = bfs.pull(src, iters); ts.append((time.perf_counter() - t) * 1e3)
    print("%s pull bits=%s graph=%s: median %.3f ms same %s" % (g, bits, gr, float(np.median(ts[3:])), bool(np.array_equal(d, ref))))
