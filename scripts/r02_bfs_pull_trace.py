"""rocprofv3 target: BFS.pull on a stand-in, five calls (scripts/r02_timeline.py cuts out the last)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets
g = sys.argv[1] if len(sys.argv) > 1 else "orkut"
m = datasets.paper_graph(g, 1.0, device=torch.device("cuda:0"))
bfs = app.BFS(16, 0, 0, 0)
bfs.set_up_runtime(); bfs.load_and_format_matrix(m, True); bfs.send_matrix_host_to_device()
for i in range(5):
    capi.sync()
    t0 = time.perf_counter()
    d = bfs.pull(0, datasets.PAPER_GRAPHS[g]["iters"])
    print("CALL %d %.3f ms reached %d" % (i, (time.perf_counter() - t0) * 1e3, int((d != 0).sum())), flush=True)
