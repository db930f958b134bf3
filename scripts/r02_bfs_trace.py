"""Evidence for SURVEY 8f-1: a rocprofv3 kernel + memory-copy trace of BFS.pull_push on the orkut stand-in.
Run under `rocprofv3 --kernel-trace --memory-copy-trace`; prints wall-clock markers so the summary can cut out the last
(replayed) call.  scripts/r02_bfs_trace_summary.py turns the CSVs into the per-call copy list."""
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
    t0 = time.time_ns()
    d = bfs.pull_push(0, datasets.PAPER_GRAPHS[g]["iters"], 0.001)
    t1 = time.time_ns()
    print("CALL %d %d %d reached %d pushes %d again %d" % (i, t0, t1, int((d != 0).sum()), bfs.push_iterations_, bfs.push_iterations_again_), flush=True)
