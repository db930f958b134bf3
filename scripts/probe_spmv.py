"""Scratch perf probe (not part of the product): time gl_spmv_run on a paper-graph stand-in."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="ogbn_products")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--ops", default="0")
    ap.add_argument("--shard", default="0/1", help="k/N: time only the k-th of N nnz-balanced row shards")
    ap.add_argument("--flags", type=int, default=0, help="GL_PLAN_* flags (1 NO_MULADD, 2 BOOLEAN)")
    ap.add_argument("--no-copy", action="store_true")
    ap.add_argument("--density", type=float, default=0.5, help="fraction of non-zeros in x")
    ap.add_argument("--lib", default=None, help="load this build of libgraphlily_hip.so instead (A/B runs on one box)")
    args = ap.parse_args()
    if args.lib:
        capi.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device("cuda:0")
    capi.init(0)
    capi.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.time()
    m = datasets.paper_graph(args.graph, args.scale, device=dev)
    io.util_round_csr_matrix_dim(m, 128, 8)
    m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32)
    t1 = time.time()
    lens = np.diff(m.adj_indptr.astype(np.int64))
    print("graph %s: n=%d nnz=%d gen %.1fs; deg max %d mean %.1f median %d empty %d" %
          (args.graph, m.num_rows, m.nnz, t1 - t0, lens.max(), lens.mean(), np.median(lens), (lens == 0).sum()),
          flush=True)
    from graphlily_amd.dist import partition_rows_by_nnz
    k, N = (int(t) for t in args.shard.split("/"))
    bounds = partition_rows_by_nnz(m.adj_indptr, N)
    plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, bounds[k], bounds[k + 1], args.flags)
    print("plan create %.1fs" % (time.time() - t1), plan.info(), flush=True)
    x = (torch.rand(m.num_cols, device=dev) < args.density).float()
    mask = torch.randint(0, 2, (m.num_rows,), device=dev).float()
    y = torch.zeros(m.num_rows, device=dev)
    bx, bm, by = (capi.DeviceBuffer.from_torch(t) for t in (x, mask, y))
    snnz = plan.info()["nnz"]
    nbytes = 8 * snnz + 8 * (bounds[k + 1] - bounds[k]) + 4 * m.num_cols + 4
    for op in [int(o) for o in args.ops.split(",")]:
        for mt in (0, 1):
            zero = 0.0 if op < 2 else 255.0
            for _ in range(3):
                plan.run(bx, bm, by, op, zero, mt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                plan.run(bx, bm, by, op, zero, mt)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            print("op %d mask %d: %.3f ms  %.1f GTEPS  %.0f GB/s effective (%.1f%% of 8 TB/s)" %
                  (op, mt, ms, snnz / ms / 1e6, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80), flush=True)
    if args.no_copy:
        return
    # streaming ceiling for reference: float4 copy of a 2 GiB buffer
    a = torch.empty(1 << 29, device=dev)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("copy ceiling: %.0f GB/s (read+write)" % (2 * a.numel() * 4 / ms / 1e6))


if __name__ == "__main__":
    main()
