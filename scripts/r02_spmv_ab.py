"""Scratch A/B (not part of the product): time gl_spmv_run on the stand-ins under plan-level knobs, one process, one box.

    python scripts/r02_spmv_ab.py --graphs googleplus,pokec --variants "base;FUSE=0;FUSE=0,BLOCKS=256,SEGMENTS=1"

A variant is a comma list of GRAPHLILY_SPMV_<KEY>=<value> settings applied while the plan is created (the knobs are
read at plan creation).  Prints ms per run (100 back-to-back runs, events on the library's stream) and the fraction of
8 TB/s the algorithmic bytes amount to; every variant is checked against an f64 row sample.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="googleplus,pokec")
    ap.add_argument("--variants", default="base;FUSE=0")
    ap.add_argument("--flags", type=int, default=capi.GL_PLAN_KEEP_VALUES)
    ap.add_argument("--op", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    capi.init(0)
    capi.set_stream(torch.cuda.current_stream().cuda_stream)
    out = []
    for name in args.graphs.split(","):
        m = datasets.paper_graph(name, 1.0, device=dev)
        m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32)
        io.util_round_csr_matrix_dim(m, 128, 8)
        x = torch.randint(0, 2, (m.num_cols,), device=dev).float()
        y = torch.zeros(m.num_rows, device=dev)
        bx, by = (capi.DeviceBuffer.from_torch(t) for t in (x, y))
        nbytes = 8 * m.nnz + 4 * (m.num_rows + 1) + 4 * m.num_cols + 4 * m.num_rows
        rs = np.random.default_rng(0).integers(0, m.num_rows, size=1000)
        ip = m.adj_indptr.astype(np.int64)
        xs = x.cpu().numpy().astype(np.float64)
        chk = np.array([np.dot(m.adj_data[ip[r]:ip[r + 1]].astype(np.float64), xs[m.adj_indices[ip[r]:ip[r + 1]]]) for r in rs])
        zero = 0.0 if args.op < 2 else 255.0
        plans = []
        for var in args.variants.split(";"):
            keys = []
            if var != "base":
                for kv in var.split(","):
                    k, v = kv.split("=")
                    os.environ["GRAPHLILY_SPMV_" + k] = v
                    keys.append("GRAPHLILY_SPMV_" + k)
            plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=args.flags)
            for k in keys:
                del os.environ[k]
            plans.append((var, plan))
        best = {}
        for rep in range(args.reps):   # interleaved repetitions: drift of the box hits every variant alike
            for var, plan in plans:
                for _ in range(3):
                    plan.run(bx, None, by, args.op, zero, 0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    plan.run(bx, None, by, args.op, zero, 0)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 100
                ok = bool(np.allclose(y.cpu().numpy()[rs], chk, rtol=1e-5, atol=1e-12)) if args.op == 0 else None
                best.setdefault(var, []).append((ms, ok))
        for var, plan in plans:
            mss = [t for t, _ in best[var]]
            info = plan.info()
            rec = {"graph": name, "variant": var, "ms_min": round(min(mss), 4), "ms_all": [round(t, 4) for t in mss],
                   "frac_hbm_peak": round(nbytes / min(mss) / 1e6 / 8000, 4), "ok": all(o for _, o in best[var] if o is not None),
                   "blocks": info["blocks"], "segments": info["segments"], "max_block_rows": info["max_block_rows"],
                   "hot_columns": info["hot_columns"], "hot_share": round(info["hot_nnz"] / max(info["nnz"], 1), 3), "mix": info["mix"],
                   "layout": info["layout"]}
            print(json.dumps(rec), flush=True)
            out.append(rec)
        del plans, bx, by, x, y, m
    if args.out:
        with open(args.out, "w") as f:
            for rec in out:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
