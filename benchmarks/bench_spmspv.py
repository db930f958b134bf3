#!/usr/bin/env python
"""SpMSpV sparsity sweep of the reference (benchmark/bench_spmspv.cpp), one GPU.

Same protocol: matrix values 1/num_rows (:153), every k-th column active with values (rand() % 99 + 1) / 100
(:157-185), Arithmetic semiring, kNoMask (:283, :323), vector sparsity 90 ... 99.99 % (:270-276), one warm-up
run that is VERIFIED (here against an f64 evaluation of the same product on the host, the benchmark does not
import the oracle) and 20 timed blocking runs (:228-236); bytes = 8 x sum nnz(active columns) (:61-76),
GTEPS = GB/s / 8 (:238-239).  By default the module gets a row-wise plan of its own matrix, so vectors whose
columns hold more than 1/32 of the non-zeros are applied row-wise (`direction` in the output); --no-own-pull
measures the scatter alone.  The reference's list (:261-268) ends with two off-repo matrices,
uniform_conflict_free_1M_{64,256}: stand-ins are built here as 2^20 x 2^20 CSC matrices with 64 / 256 entries per
column at rows (c + k (n / d + 1)) mod n -- uniform, and consecutive entries of a column fall into different banks
of the FPGA's 8-way interleaved output buffer ("conflict free").  Tropical is swept too (the reference has the line
commented out, :279): values 1/num_rows + vector value, min over the active columns, zero = FLOAT_INF.

    python benchmarks/bench_spmspv.py [--graphs googleplus,pokec] [--semirings Arithmetic,Logical] [--out profiles/rNN_spmspv_sweep.jsonl]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SPARSITIES = (0.90, 0.95, 0.99, 0.995, 0.999, 0.9995, 0.9999)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="googleplus,ogbl_ppa,hollywood,pokec,ogbn_products,"
                                        "uniform_conflict_free_1M_64,uniform_conflict_free_1M_256")
    ap.add_argument("--semirings", default="Arithmetic,Tropical")
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-own-pull", action="store_true",
                    help="scatter only: do not give the module a row-wise plan of its matrix for heavy vectors")
    args = ap.parse_args()
    import scipy.sparse as sp
    import torch
    from graphlily_amd import capi, datasets, io, module as M
    dev = torch.device("cuda:0")
    capi.init(0)
    sems = {"Arithmetic": M.ArithmeticSemiring, "Logical": M.LogicalSemiring, "Tropical": M.TropicalSemiring}
    lines = []
    for name in args.graphs.split(","):
        if name.startswith("uniform_conflict_free_1M_"):
            n, d = 1 << 20, int(name.rsplit("_", 1)[1])
            rows = ((np.arange(n, dtype=np.int64)[:, None] + np.arange(d, dtype=np.int64)[None, :] * (n // d + 1)) % n)
            rows.sort(axis=1)      # rows ascending inside a column, like csr2csc leaves them (io/data_loader.h:108-144)
            csc = io.CSCMatrix(n, n, np.full(n * d, np.float32(1.0 / n), np.float32), rows.reshape(-1).astype(np.uint32),
                               (np.arange(n + 1, dtype=np.int64) * d).astype(np.uint32))
            del rows
        else:
            csr = datasets.paper_graph(name, 1.0, device=dev)
            csr.adj_data = np.full(csr.nnz, np.float32(1.0 / csr.num_rows), np.float32)
            csc = io.csr2csc(csr)
            del csr
        coldeg = np.diff(csc.adj_indptr.astype(np.int64))
        A = sp.csc_matrix((csc.adj_data.astype(np.float64), csc.adj_indices.astype(np.int64), csc.adj_indptr.astype(np.int64)),
                          shape=(csc.num_rows, csc.num_cols))
        for sname in args.semirings.split(","):
            mod = M.SpMSpVModule(512 * 1024)
            mod.set_semiring(sems[sname])
            mod.set_mask_type(M.kNoMask)
            mod.set_up_runtime("unused.xclbin")
            mod.load_and_format_matrix(csc)
            mod.send_matrix_host_to_device()
            if not args.no_own_pull:
                mod.enable_own_pull()       # direction switch inside the operator (gl_spmspv_plan_attach_pull)
            mod.send_mask_host_to_device(np.zeros(csc.num_rows, np.float32))
            rng = np.random.default_rng(1)
            for sparsity in SPARSITIES:
                cnt = int(np.floor((1 - sparsity) * csc.num_cols))
                if cnt == 0:
                    continue
                idx = (np.arange(cnt, dtype=np.int64) * (csc.num_cols // cnt)).astype(np.uint32)
                vals = np.ones(cnt, np.float32) if sname == "Logical" else ((rng.integers(0, 99, size=cnt) + 1) / 100.0).astype(np.float32)
                mod.send_vector_host_to_device(M.make_sparse_vec(idx, vals))
                mod.run()                                   # warm-up, verified
                res = mod.send_results_device_to_host()
                if sname == "Tropical":
                    # min over the active columns of (a + x) in float32 (one rounding, the device's own); bit-exact
                    got32 = M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, M.FLOAT_INF)
                    ip = csc.adj_indptr.astype(np.int64)
                    lens = (ip[idx.astype(np.int64) + 1] - ip[idx.astype(np.int64)])
                    pos = np.repeat(ip[idx.astype(np.int64)], lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
                    cand = (csc.adj_data[pos] + np.repeat(vals, lens)).astype(np.float32)
                    ref32 = np.full(csc.num_rows, np.float32(M.FLOAT_INF), np.float32)
                    np.minimum.at(ref32, csc.adj_indices[pos].astype(np.int64), cand)
                    ok = bool(np.array_equal(got32, ref32))
                else:
                    got = M.convert_sparse_vec_to_dense_vec(res, csc.num_rows, 0.0).astype(np.float64)
                    x = np.zeros(csc.num_cols)
                    x[idx] = vals.astype(np.float64)
                    ref = A @ x
                    if sname == "Logical":
                        ref = (ref != 0).astype(np.float64)
                    ok = bool(np.allclose(got, ref, rtol=1e-5, atol=1e-12))
                # the host-side verification above left the GPU idle for milliseconds; the reference times right after its
                # warm-up run (:228-236), so one more untimed run brings the device back to that state
                mod.run()
                ts = []
                for _ in range(args.runs):
                    t0 = time.perf_counter()
                    mod.run()                               # blocking, like the reference's run()
                    ts.append(time.perf_counter() - t0)
                ms = float(np.mean(ts)) * 1e3
                active = int(coldeg[idx].sum())
                # the same runs enqueued back to back (no completion record awaited per call), one synchronisation at the end:
                # what an app loop that does not need the count on the host pays per iteration
                mod.blocking = False
                capi.sync()
                t0 = time.perf_counter()
                for _ in range(args.runs):
                    mod.run()
                capi.sync()
                b2b = (time.perf_counter() - t0) / args.runs * 1e3
                mod.blocking = True
                direction = mod.plan_.last_direction()
                # bytes MOVED per product: bin + fold 24 B (stream read, bin write, bin read), the one-launch kernel 8 B, row-wise runs
                # stream the attached SpMV plan (no per-product figure: 0)
                bpp = 0 if direction == "row-wise" else (8 if (cnt <= 1024 and active <= 2048) else 24)
                rec = {"graph": name, "semiring": sname, "vector_sparsity": sparsity, "vector_nnz": cnt,
                       "active_nnz": active, "result_nnz": int(res["index"][0]), "ms": round(ms, 4),
                       "ms_median": round(float(np.median(ts)) * 1e3, 4), "ms_max": round(float(np.max(ts)) * 1e3, 4),
                       "ms_first": [round(t * 1e3, 4) for t in ts[:3]],
                       "gbps": round(8 * active / ms / 1e6, 2), "gteps": round(active / ms / 1e6, 3), "verified": ok,
                       "ms_back_to_back": round(b2b, 4), "gbps_back_to_back": round(8 * active / b2b / 1e6, 2),
                       "bytes_per_product": bpp, "moved_gbps": round(bpp * active / ms / 1e6, 2), "direction": direction}
                print(json.dumps(rec), flush=True)
                lines.append(rec)
            del mod
    if args.out:
        with open(args.out, "w") as f:
            for rec in lines:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
