"""Round 4, same-box A/B of the BFS / SSSP schedules on the six stand-ins (pull_push slower than in
round 2 on several graphs, BFS pull_push on googleplus 2.4 x slower than its own pull).

Every variant runs in ONE process on ONE box, on the same app objects: the switches are read per call.
    python scripts/r04_ab_schedules.py [graphs...]  > profiles/r04_ab_schedules.txt
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets, module as M  # noqa: E402

BFS_VARIANTS = (
    ("default", {}),
    ("two_launch_slots", {"GRAPHLILY_BFS_ONE_LAUNCH": "0"}),
    ("float_readback", {"GRAPHLILY_BFS_U8": "0"}),
    ("two_launch+float (r02)", {"GRAPHLILY_BFS_ONE_LAUNCH": "0", "GRAPHLILY_BFS_U8": "0"}),
    ("no_graph", {"GRAPHLILY_BFS_GRAPH": "0"}),
)
SSSP_VARIANTS = (
    ("host_loop (default since r04)", {}),
    ("device_schedule", {"GRAPHLILY_SSSP_DEVICE_LOOP": "1"}),


)


def med(fn, runs=9, warm=4):
    for _ in range(warm):
        out = fn()
    ts = []
    for _ in range(runs):
        capi.sync()
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3, out


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    graphs = sys.argv[1:] or list(datasets.PAPER_GRAPHS)
    dev = torch.device("cuda:0")
    for g in graphs:
        raw = datasets.paper_graph(g, 1.0, device=dev)
        iters = datasets.PAPER_GRAPHS[g]["iters"]
        deg = np.diff(raw.adj_indptr.astype(np.int64))
        src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
        rec = {"graph": g, "n": raw.num_rows, "nnz": raw.nnz, "iters": iters, "bfs": {}, "sssp": {}}
        bfs = app.BFS(16, 0, 0, 0)
        bfs.set_up_runtime()
        bfs.load_and_format_matrix(raw, True)
        bfs.send_matrix_host_to_device()
        ref = None
        for name, env in BFS_VARIANTS:
            ms, mn, d = with_env(env, lambda: med(lambda: bfs.pull_push(src, iters, 0.001)))
            ref = d.copy() if ref is None else ref
            line = {"pull_push_ms": round(ms, 4), "min_ms": round(mn, 4), "equal": bool(np.array_equal(d, ref)),
                    "pushes": int(bfs.push_iterations_)}
            if hasattr(bfs, "bfs_slot_modes_"):
                line["slot_modes"] = [int(x) for x in bfs.bfs_slot_modes_]
                line["slot_counts"] = [int(x) for x in bfs.bfs_slot_counts_]
            # GPU time of the schedule alone (no read-back)
            bfs.time_schedule_ = True
            with_env(env, lambda: [bfs.pull_push(src, iters, 0.001) for _ in range(3)])
            line["schedule_gpu_ms"] = round(float(getattr(bfs, "schedule_ms_", float("nan"))), 4)
            bfs.time_schedule_ = False
            rec["bfs"][name] = line
        ms, mn, d = med(lambda: bfs.pull(src, iters))
        rec["bfs"]["pull"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "equal": bool(np.array_equal(d, ref))}
        bfs.time_schedule_ = True
        for _ in range(3):
            bfs.pull(src, iters)
        rec["bfs"]["pull"]["schedule_gpu_ms"] = round(float(getattr(bfs, "schedule_ms_", float("nan"))), 4)
        del bfs
        s = app.SSSP(16, 0, 0, 0)
        s.set_up_runtime()
        s.load_and_format_matrix(raw, True)
        s.send_matrix_host_to_device()
        ref = None
        for name, env in SSSP_VARIANTS:
            ms, mn, d = with_env(env, lambda: med(lambda: s.pull_push(src, iters, 0.001), runs=7, warm=4))
            ref = d.copy() if ref is None else ref
            rec["sssp"][name] = {"pull_push_ms": round(ms, 4), "min_ms": round(mn, 4), "equal": bool(np.array_equal(d, ref)),
                                 "pushes": int(s.push_iterations_)}
        ms, mn, d = med(lambda: s.pull(src, iters), runs=5, warm=2)
        rec["sssp"]["pull"] = {"ms": round(ms, 4), "min_ms": round(mn, 4), "equal": bool(np.array_equal(d, ref))}
        del s
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
