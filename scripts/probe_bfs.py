"""Scratch: where does a BFS run spend its time?  Wall time of every driver step with a device sync after it."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets, module as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="orkut")
    ap.add_argument("--scale", type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = datasets.PAPER_GRAPHS[args.graph]
    raw = datasets.paper_graph(args.graph, args.scale, device=dev)
    bfs = app.BFS(16, 0, 0, 0, backend=app.HipBackend(0, use_torch=True))
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(raw, True)
    bfs.send_matrix_host_to_device()
    deg = np.diff(raw.adj_indptr.astype(np.int64))
    source = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
    iters = g["iters"]
    B = bfs.backend

    def timed(label, fn, acc):
        B.sync()
        t0 = time.perf_counter()
        r = fn()
        B.sync()
        acc.append((label, (time.perf_counter() - t0) * 1e3))
        return r

    for rep in range(3):
        acc = []
        n = bfs.n_
        fr, dist, local = timed("start_push", lambda: bfs._start_push(source), acc)
        it = 1
        while True:
            timed("spmspv %d" % it, bfs.SpMSpV_.run, acc)
            timed("sparse_assign %d" % it, lambda: bfs.SparseAssign_.run(float(it + 1)), acc)
            nnz = timed("get_nnz %d" % it, bfs.SpMSpV_.get_results_nnz, acc)
            timed("copy frontier %d (nnz %d)" % (it, nnz), lambda: B.copy(fr, local, 8 * (1 + nnz)), acc)
            it += 1
            if not (it < iters and float(nnz) / n < 0.001):
                break
        vec = timed("alloc vector", lambda: B.alloc(n, np.float32), acc)
        timed("sparse_to_dense", lambda: B.sparse_to_dense(fr, vec, n, M.LogicalSemiring.zero, n), acc)
        timed("bind_pull", lambda: bfs._bind_pull(vec, dist), acc)
        while it <= iters:
            timed("spmv %d" % it, bfs.SpMV_.run, acc)
            timed("ewise %d" % it, lambda: bfs.eWiseAdd_.run(n, 0.0), acc)
            timed("assign %d" % it, lambda: bfs.DenseAssign_.run(n, float(it + 1)), acc)
            it += 1
        timed("download", lambda: B.download_result(dist, n), acc)
    print("total (sum of synced steps) %.3f ms" % sum(t for _, t in acc))
    for label, t in acc:
        print("  %-34s %.3f ms" % (label, t))
    for mode in ("pull", "pull_push"):
        fn = (lambda: bfs.pull(source, iters)) if mode == "pull" else (lambda: bfs.pull_push(source, iters, 0.001))
        fn()
        ts = []
        for _ in range(5):
            B.sync()
            t0 = time.perf_counter()
            fn()
            B.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%s: median %.3f ms  min %.3f" % (mode, float(np.median(ts)), min(ts)))


if __name__ == "__main__":
    main()
