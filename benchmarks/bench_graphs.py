#!/usr/bin/env python
"""Per-graph numbers on the six benchmark graphs of the reference (benchmark/run_*.sh), one GPU.

Mirrors the reference's drivers and metric definitions:
  SpMV      benchmark/bench_spmv.cpp      1 warm-up + 100 runs, GTEPS = nnz / t                 (+ effective GB/s)
  BFS       benchmark/bench_bfs.cpp       pull and pull_push(threshold 0.001), GTEPS = nnz * iters / t
  SSSP      benchmark/bench_sssp.cpp      same protocol
  PageRank  benchmark/bench_pagerank.cpp  damping 0.9, 10 iterations, GTEPS = nnz / t_iter
on the synthetic stand-ins of graphlily_amd/datasets.py (the real files are not available offline; `--npz PATH` runs a real
scipy-npz CSR through the same code).  Each line is also checked: SpMV against float64 on a row sample, BFS pull == pull_push,
SSSP pull == pull_push.  `run_graph` is what bench.py's `six_graphs` object calls.

    python benchmarks/bench_graphs.py [--graphs googleplus,orkut] [--npz file.npz --iters 10] [--out profiles/rNN_six_graphs.jsonl]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def timed(fn, runs, warm=3):
    """Median wall time of `runs` calls after `warm` untimed ones (the device-resident schedules record their hipGraph on the
    third call with the same arguments)."""
    from graphlily_amd import capi
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        capi.sync()
        t0 = time.perf_counter()
        out = fn()
        capi.sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def bfs_times(bfs, src, iters, runs, fence=None, threshold=0.001):
    """bench_bfs.cpp:55-89 for both modes, pull_push first: 14 untimed calls (the schedule is recorded on the third call; the
    driver then measures both read-back ways), then the median wall time of `runs` calls, each bracketed by `fence`.  ONE helper
    for bench.py's headline `bfs` object and for every `six_graphs` line, so that the two cannot drift apart (round 4: 33 %).
    Returns {mode: {"s", "d", "push_iterations", "readback", "slot_modes"}}."""
    from graphlily_amd import capi
    fence = fence or capi.sync
    out = {}
    for mode in ("pull_push", "pull"):
        fn = (lambda: bfs.pull_push(src, iters, threshold)) if mode == "pull_push" else (lambda: bfs.pull(src, iters))
        for _ in range(14):
            d = fn()
        ts = []
        for _ in range(runs):
            fence()
            t0 = time.perf_counter()
            d = fn()
            fence()
            ts.append(time.perf_counter() - t0)
        rb = getattr(bfs, "readback_", None)
        modes = getattr(bfs, "bfs_slot_modes_", None)
        out[mode] = {"s": float(np.median(ts)), "d": d, "push_iterations": getattr(bfs, "push_iterations_", None),
                     "readback": dict(rb) if rb is not None else None, "slot_modes": None if modes is None else modes.copy()}
    return out


def edges_traversed(bfs, raw, d, iters, modes=None):
    """beside the nominal GTEPS, the edges a run actually looked at, from how the device evaluated every slot
    (BFS.bfs_slot_modes_): scattered = the non-zeros of the frontier's columns; streamed row-wise = every non-zero of the
    matrix; bottom-up = AT MOST the non-zeros of the rows not reached before the slot (a row stops at its first hit)."""
    d = d.astype(np.int64)
    modes = [int(v) for v in (bfs.bfs_slot_modes_ if modes is None else modes)]
    n = d.shape[0]
    ip = raw.adj_indptr.astype(np.int64)
    row_len = np.zeros(n, np.int64)
    row_len[:raw.num_rows] = np.diff(ip)
    col_len = np.bincount(raw.adj_indices[:ip[-1]], minlength=n).astype(np.int64)
    total, bound = 0, False
    for s in range(1, iters + 1):
        m = modes[s - 1] if s - 1 < len(modes) else 0
        if m == 1:
            total += int(col_len[d == s].sum())
        elif m == 2:
            total += int(ip[-1])
        elif m == 3:
            total += int(row_len[(d == 0) | (d > s)].sum())
            bound = True
    return total, bound, modes[:iters]


def run_graph(name, raw, iters, dev, runs=5, spmv_steps=100, apps=("bfs", "pagerank", "sssp")):
    """One graph: general-layout SpMV (wall ms, effective GB/s against the 8 B/nnz algorithmic bytes, kernel ms by HIP events),
    pattern and boolean layouts, BFS / PageRank / SSSP by the reference's metric definitions, each with its self-check."""
    import torch
    from graphlily_amd import app, capi, io
    rec = {"graph": name, "n": raw.num_rows, "nnz": raw.nnz, "iters": iters}
    # ---- SpMV, fp32 (+,x): values 1 / num_rows, rows padded to x128, columns to x8, x in {0, 1} (bench_spmv.cpp:50-60)
    m = raw.copy()
    m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32)
    io.util_round_csr_matrix_dim(m, 128, 8)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    x = torch.randint(0, 2, (m.num_cols,), generator=gen, device=dev).float()
    mask = torch.randint(0, 2, (m.num_rows,), generator=gen, device=dev).float()
    y = torch.zeros(m.num_rows, device=dev)
    bx, bm, by = (capi.DeviceBuffer.from_torch(t) for t in (x, mask, y))
    torch.cuda.synchronize()          # (torch filled them on its stream; the library runs on its own)
    nbytes = 8 * m.nnz + 4 * (m.num_rows + 1) + 4 * m.num_cols + 4 * m.num_rows
    rs = np.random.default_rng(0).integers(0, m.num_rows, size=1000)
    ip = m.adj_indptr.astype(np.int64)
    xs = x.cpu().numpy().astype(np.float64)
    chk = np.array([np.dot(m.adj_data[ip[r]:ip[r + 1]].astype(np.float64), xs[m.adj_indices[ip[r]:ip[r + 1]]]) for r in rs])

    def spmv_line(flags, op, zero, mask_type, prof=False):
        plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=flags)
        mk = bm if mask_type else None
        # Warm-up, a short idle (the seconds of formatting kernels that plan creation has just run leave the first tens of
        # milliseconds slower than the steady state: bench.py --settle), then a timed region of >= ~60 ms sized from a probe.
        t0 = time.perf_counter()
        for _ in range(10):
            plan.run(bx, mk, by, op, zero, mask_type)
        capi.sync()
        est = (time.perf_counter() - t0) / 10
        time.sleep(0.25)
        for _ in range(3):
            plan.run(bx, mk, by, op, zero, mask_type)
        capi.sync()
        steps = int(min(max(spmv_steps, 0.06 / max(est, 1e-6)), 2000))
        if prof:
            capi.prof_begin(steps, every=4)
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.run(bx, mk, by, op, zero, mask_type)
        capi.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        info = plan.info()
        line = {"ms": round(ms, 4), "gteps": round(m.nnz / ms / 1e6, 1), "layout": info["layout"], "steps": steps}
        if prof:
            total, launches = capi.prof_end()
            line["kernel_ms"] = round(total / max(launches, 1), 4)
        if op == 0:
            line["ok"] = bool(np.allclose(y.cpu().numpy()[rs], chk, rtol=1e-5, atol=1e-12))
        return line, info

    # headline: the general layout (values streamed), 8 B/nnz algorithmic bytes against the HBM peak
    line, info = spmv_line(capi.GL_PLAN_KEEP_VALUES, 0, 0.0, 0, prof=True)
    line.update({"eff_gbps": round(nbytes / line["ms"] / 1e6, 1), "frac_hbm_peak": round(nbytes / line["ms"] / 1e6 / HBM_PEAK_GBPS, 4),
                 "kernel_frac_hbm_peak": round(nbytes / line["kernel_ms"] / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": nbytes,
                 "shape": {k: info[k] for k in ("blocks", "segments", "num_units") if k in info}})
    rec["spmv"] = line
    # what the default plan does with this constant-valued matrix, and the (||,&&) layout BFS pulls with
    rec["spmv_pattern"] = spmv_line(0, 0, 0.0, 0)[0]
    rec["spmv_boolean_masked"] = spmv_line(capi.GL_PLAN_BOOLEAN, 1, 0.0, 1)[0]
    del bx, bm, by, x, mask, y, m
    deg = np.diff(raw.adj_indptr.astype(np.int64))
    src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
    if "bfs" in apps:
        bfs = app.BFS(16, 0, 0, 0)
        bfs.set_up_runtime()
        bfs.load_and_format_matrix(raw, True)
        bfs.send_matrix_host_to_device()
        bt = bfs_times(bfs, src, iters, runs)
        t_pull, d_pull, t_pp, d_pp = bt["pull"]["s"], bt["pull"]["d"], bt["pull_push"]["s"], bt["pull_push"]["d"]
        e_pull = edges_traversed(bfs, raw, d_pull, iters, bt["pull"]["slot_modes"]) if bt["pull"]["slot_modes"] is not None else None
        e_pp = edges_traversed(bfs, raw, d_pp, iters, bt["pull_push"]["slot_modes"]) if bt["pull_push"]["slot_modes"] is not None else None
        nnz = bfs.get_nnz()
        rec["bfs"] = {"source": src, "pull_ms": round(t_pull * 1e3, 3), "pull_gteps": round(nnz * iters / t_pull / 1e9, 1),
                      "pull_push_ms": round(t_pp * 1e3, 3), "pull_push_gteps": round(nnz * iters / t_pp / 1e9, 1),
                      "push_iterations": bt["pull_push"]["push_iterations"], "reached": int((d_pull != 0).sum()),
                      "readback": {k: (bt[k]["readback"] or {}).get("way") for k in ("pull_push", "pull")},
                      "ok": bool(np.array_equal(d_pull, d_pp))}
        if e_pp is not None:
            rec["bfs"].update({"pull_push_gteps_traversed": round(e_pp[0] / t_pp / 1e9, 1), "pull_gteps_traversed": round(e_pull[0] / t_pull / 1e9, 1),
                               "traversed_is_upper_bound": bool(e_pp[1] or e_pull[1]), "pull_push_slot_modes": e_pp[2]})
        del bfs
    if "pagerank" in apps:
        pr = app.PageRank(16, 0, 0)
        pr.set_up_runtime()
        pr.load_and_format_matrix(raw, 0.9, True)
        pr.send_matrix_host_to_device()
        t_pr, r = timed(lambda: pr.pull(0.9, 10), runs, warm=1)
        s = float(r.astype(np.float64).sum())
        rec["pagerank"] = {"ms_per_iter": round(t_pr * 1e3 / 10, 4), "gteps": round(pr.get_nnz() / (t_pr / 10) / 1e9, 1),
                           "rank_sum": s, "ok": bool(np.isfinite(s) and 0.0 < s <= 1.0 + 1e-3)}
        del pr
    if "sssp" in apps:
        ss = app.SSSP(16, 0, 0, 0)
        ss.set_up_runtime()
        ss.load_and_format_matrix(raw, True)
        ss.send_matrix_host_to_device()
        t_sp, d1 = timed(lambda: ss.pull(src, iters), runs, warm=1)
        t_spp, d2 = timed(lambda: ss.pull_push(src, iters, 0.001), runs)
        nnz = ss.get_nnz()
        rec["sssp"] = {"pull_ms": round(t_sp * 1e3, 3), "pull_gteps": round(nnz * iters / t_sp / 1e9, 1),
                       "pull_push_ms": round(t_spp * 1e3, 3), "pull_push_gteps": round(nnz * iters / t_spp / 1e9, 1),
                       "push_iterations": ss.push_iterations_, "ok": bool(np.array_equal(d1, d2))}
        del ss
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="googleplus,ogbl_ppa,hollywood,pokec,ogbn_products,orkut")
    ap.add_argument("--npz", default=None, help="a real graph as a scipy-npz CSR (README.md:44-49 of the reference) instead of the stand-ins")
    ap.add_argument("--iters", type=int, default=10, help="BFS / SSSP iterations for --npz (benchmark/run_bfs.sh:20 lists the paper's)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--runs", type=int, default=5)
    args = ap.parse_args()
    import torch
    from graphlily_amd import capi, datasets, io
    dev = torch.device("cuda:0")
    capi.init(0)
    lines = []
    jobs = [("npz:" + os.path.basename(args.npz), None)] if args.npz else [(g, g) for g in args.graphs.split(",")]
    for label, name in jobs:
        if name is None:
            raw, iters, data = io.load_csr_matrix_from_float_npz(args.npz), args.iters, "file " + args.npz
        else:
            raw, iters, data = datasets.paper_graph(name, 1.0, device=dev), datasets.PAPER_GRAPHS[name]["iters"], "synthetic R-MAT stand-in"
        rec = run_graph(label, raw, iters, dev, runs=args.runs)
        rec["data"] = data
        print(json.dumps(rec), flush=True)
        lines.append(rec)
    if args.out:
        with open(args.out, "w") as f:
            for rec in lines:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
