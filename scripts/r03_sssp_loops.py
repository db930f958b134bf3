"""SSSP::pull_push (app/sssp.h:197-243) on the stand-ins: the host-driven loop (reads the count back every push iteration,
like the reference) against the device-resident schedule, same box, same process; and SSSP::pull."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets, module as M
for g in sys.argv[1:] or ["orkut"]:
    m = datasets.paper_graph(g, 1.0, device=torch.device("cuda:0"))
    iters = datasets.PAPER_GRAPHS[g]["iters"]
    deg = np.diff(m.adj_indptr.astype(np.int64))
    src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
    s = app.SSSP(16, 0, 0, 0, semiring=M.TropicalSemiringUfixed)
    s.set_up_runtime(); s.load_and_format_matrix(m, True); s.send_matrix_host_to_device()
    res = {}
    for name, env in (("host_loop", "0"), ("device_schedule", "1")):
        os.environ["GRAPHLILY_SSSP_DEVICE_LOOP"] = env
        for _ in range(3):
            d = s.pull_push(src, iters, 0.001)
        ts = []
        for _ in range(7):
            capi.sync(); t0 = time.perf_counter(); d = s.pull_push(src, iters, 0.001); ts.append(time.perf_counter() - t0)
        res[name] = (np.median(ts) * 1e3, d.copy(), s.push_iterations_)
    ts = []
    for _ in range(5):
        capi.sync(); t0 = time.perf_counter(); dp = s.pull(src, iters); ts.append(time.perf_counter() - t0)
    same = bool(np.array_equal(res["host_loop"][1], res["device_schedule"][1]) and np.array_equal(dp, res["host_loop"][1]))
    print("%s SSSP %d iterations, threshold 0.001: pull_push host-driven loop %.3f ms (%d pushes), device-resident schedule %.3f ms (%d pushes), pull %.3f ms; distances equal: %s"
          % (g, iters, res["host_loop"][0], res["host_loop"][2], res["device_schedule"][0], res["device_schedule"][2], np.median(ts) * 1e3, same), flush=True)
