#!/usr/bin/env python
"""Per-graph numbers on the six benchmark graphs of the reference (benchmark/run_*.sh), one GPU.

Mirrors the reference's drivers and metric definitions:
  SpMV      benchmark/bench_spmv.cpp      1 warm-up + 100 runs, GTEPS = nnz / t                 (+ effective GB/s)
  BFS       benchmark/bench_bfs.cpp       pull and pull_push(threshold 0.001), GTEPS = nnz * iters / t
  SSSP      benchmark/bench_sssp.cpp      same protocol
  PageRank  benchmark/bench_pagerank.cpp  damping 0.9, 10 iterations, GTEPS = nnz / t_iter
on the synthetic stand-ins of graphlily_amd/datasets.py (the real files are not available offline).
Each line is also checked: SpMV against float64 on a row sample, BFS pull == pull_push, SSSP pull == pull_push.

    python benchmarks/bench_graphs.py [--graphs googleplus,orkut] [--out profiles/rNN_six_graphs.jsonl]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, runs):
    import torch
    fn()
    ts = []
    for _ in range(runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="googleplus,ogbl_ppa,hollywood,pokec,ogbn_products,orkut")
    ap.add_argument("--out", default=None)
    ap.add_argument("--runs", type=int, default=5)
    args = ap.parse_args()
    import torch
    from graphlily_amd import app, capi, datasets, io, module as M
    dev = torch.device("cuda:0")
    capi.init(0)
    capi.set_stream(torch.cuda.current_stream().cuda_stream)
    lines = []
    for name in args.graphs.split(","):
        g = datasets.PAPER_GRAPHS[name]
        raw = datasets.paper_graph(name, 1.0, device=dev)
        rec = {"graph": name, "n": raw.num_rows, "nnz": raw.nnz, "iters": g["iters"], "data": "synthetic R-MAT stand-in"}
        # ---- SpMV, fp32 (+,x)
        m = raw.copy()
        m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32)
        io.util_round_csr_matrix_dim(m, 128, 8)
        x = torch.randint(0, 2, (m.num_cols,), device=dev).float()
        mask = torch.randint(0, 2, (m.num_rows,), device=dev).float()
        y = torch.zeros(m.num_rows, device=dev)
        bx, bm, by = (capi.DeviceBuffer.from_torch(t) for t in (x, mask, y))
        nbytes = 8 * m.nnz + 4 * (m.num_rows + 1) + 4 * m.num_cols + 4 * m.num_rows
        rs = np.random.default_rng(0).integers(0, m.num_rows, size=1000)
        ip = m.adj_indptr.astype(np.int64)
        xs = x.cpu().numpy().astype(np.float64)
        chk = np.array([np.dot(m.adj_data[ip[r]:ip[r + 1]].astype(np.float64), xs[m.adj_indices[ip[r]:ip[r + 1]]]) for r in rs])

        def spmv_line(flags, op, zero, mask_type):
            plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=flags)
            mk = bm if mask_type else None
            for _ in range(2):
                plan.run(bx, mk, by, op, zero, mask_type)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                plan.run(bx, mk, by, op, zero, mask_type)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 100
            line = {"ms": round(ms, 4), "gteps": round(m.nnz / ms / 1e6, 1), "layout": plan.info()["layout"],
                    "device_bytes": plan.info()["device_bytes"]}
            if op == 0:
                line["ok"] = bool(np.allclose(y.cpu().numpy()[rs], chk, rtol=1e-5, atol=1e-12))
            return line, plan.info()

        # headline: the general layout (values streamed), 8 B/nnz algorithmic bytes against the HBM peak
        line, shape = spmv_line(capi.GL_PLAN_KEEP_VALUES, 0, 0.0, 0)
        line.update({"eff_gbps": round(nbytes / line["ms"] / 1e6, 1), "frac_hbm_peak": round(nbytes / line["ms"] / 1e6 / 8000, 4),
                     "shape": shape})
        rec["spmv"] = line
        # what the default plan does with this constant-valued matrix, and the (||,&&) layout BFS pulls with
        rec["spmv_pattern"] = spmv_line(0, 0, 0.0, 0)[0]
        rec["spmv_boolean_masked"] = spmv_line(capi.GL_PLAN_BOOLEAN, 1, 0.0, 1)[0]
        del bx, bm, by, x, mask, y, m
        # ---- BFS
        deg = np.diff(raw.adj_indptr.astype(np.int64))
        src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
        bfs = app.BFS(16, 0, 0, 0, backend=app.HipBackend(0, use_torch=True))
        bfs.set_up_runtime()
        bfs.load_and_format_matrix(raw, True)
        bfs.send_matrix_host_to_device()
        t_pull, d_pull = timed(lambda: bfs.pull(src, g["iters"]), args.runs)
        t_pp, d_pp = timed(lambda: bfs.pull_push(src, g["iters"], 0.001), args.runs)
        nnz = bfs.get_nnz()
        rec["bfs"] = {"source": src, "pull_ms": round(t_pull * 1e3, 3), "pull_gteps": round(nnz * g["iters"] / t_pull / 1e9, 1),
                      "pull_push_ms": round(t_pp * 1e3, 3), "pull_push_gteps": round(nnz * g["iters"] / t_pp / 1e9, 1),
                      "push_iterations": bfs.push_iterations_, "reached": int((d_pull != 0).sum()),
                      "ok": bool(np.array_equal(d_pull, d_pp))}
        del bfs
        # ---- PageRank
        pr = app.PageRank(16, 0, 0, backend=app.HipBackend(0, use_torch=True))
        pr.set_up_runtime()
        pr.load_and_format_matrix(raw, 0.9, True)
        pr.send_matrix_host_to_device()
        t_pr, r = timed(lambda: pr.pull(0.9, 10), args.runs)
        rec["pagerank"] = {"ms_per_iter": round(t_pr * 1e3 / 10, 4), "gteps": round(pr.get_nnz() / (t_pr / 10) / 1e9, 1),
                           "rank_sum": float(r.astype(np.float64).sum())}
        del pr
        # ---- SSSP
        ss = app.SSSP(16, 0, 0, 0, backend=app.HipBackend(0, use_torch=True))
        ss.set_up_runtime()
        ss.load_and_format_matrix(raw, True)
        ss.send_matrix_host_to_device()
        t_sp, d1 = timed(lambda: ss.pull(src, g["iters"]), args.runs)
        t_spp, d2 = timed(lambda: ss.pull_push(src, g["iters"], 0.001), args.runs)
        nnz = ss.get_nnz()
        rec["sssp"] = {"pull_ms": round(t_sp * 1e3, 3), "pull_gteps": round(nnz * g["iters"] / t_sp / 1e9, 1),
                       "pull_push_ms": round(t_spp * 1e3, 3), "pull_push_gteps": round(nnz * g["iters"] / t_spp / 1e9, 1),
                       "push_iterations": ss.push_iterations_, "ok": bool(np.array_equal(d1, d2))}
        del ss
        print(json.dumps(rec), flush=True)
        lines.append(rec)
    if args.out:
        with open(args.out, "w") as f:
            for rec in lines:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
