#!/usr/bin/env python
"""Run the REFERENCE's own benchmark drivers (benchmark/bench_{spmv,bfs,pagerank,sssp}.cpp, compiled unmodified
against include/graphlily by `make -C oracle ref_benches`) on the HIP backend and print what they print.

The binaries take the reference's positional arguments (benchmark/run_bfs.sh:3-10):
    num_channels spmv_out_buf_len spmspv_out_buf_len vec_buf_len bitstream dataset num_iterations
The buffer sizes and the bitstream path are ignored by this backend.  The dataset is a stand-in written as
an uncompressed scipy npz (the loader takes stored and deflated members alike).

    python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="googleplus")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--apps", default="spmv,spmv_verify,bfs,pagerank,sssp",
                    help="spmv = bench_spmv.cpp as it is (100 blocking runs, :96-112); spmv_verify = the same file with its own "
                         "verify (:15-33) called on the results (tests/cpp/ref_bench_spmv_verify.cpp)")
    ap.add_argument("--npz", default=None, help="an already written stand-in (skips generation)")
    args = ap.parse_args()
    import scipy.sparse as sp
    import torch
    from graphlily_amd import datasets
    g = datasets.PAPER_GRAPHS[args.graph]
    dev = torch.device("cuda:0") if torch.cuda.is_available() else None
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        if args.npz:
            path = args.npz
            print("# %s stand-in from %s" % (args.graph, path), flush=True)
        else:
            m = datasets.paper_graph(args.graph, args.scale, device=dev)
            A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)),
                              shape=(m.num_rows, m.num_cols), dtype=np.float32)
            path = os.path.join(tmp, "%s_csr_float32.npz" % args.graph)
            t0 = time.time()
            sp.save_npz(path, A, compressed=False)
            print("# %s stand-in: n=%d nnz=%d, npz written in %.1f s" % (args.graph, m.num_rows, m.nnz, time.time() - t0), flush=True)
            del m, A
        rc = 0
        for a in args.apps.split(","):
            # <app>_refapps: the same driver compiled with -DGRAPHLILY_USE_REFERENCE_APPS (the checkout's app headers over the
            # module layer) instead of this repo's include/graphlily/app/ drivers
            refapps = a.endswith("_refapps")
            a = a[:-len("_refapps")] if refapps else a
            exe = os.path.join(ROOT, "oracle", "_ref", "bench_%s_on_hip%s" % (a, "_refapps" if refapps else ""))
            if not os.path.exists(exe):
                print("# %s not built (needs /root/reference: make -C oracle ref_benches)" % exe)
                rc = 2
                continue
            if a == "spmv_verify":
                cmd = [exe, path]
            elif a in ("pagerank", "spmv"):     # bench_pagerank.cpp:67-75, bench_spmv.cpp:116-122: num_channels out_buf_len vec_buf_len bitstream dataset
                cmd = [exe, "16", "1024000", "30720", "unused.xclbin", path]
            else:
                cmd = [exe, "16", "1024000", "256000", "30720", "unused.xclbin", path, str(g["iters"])]
            print("# " + " ".join(cmd[:5] + ["..."]), flush=True)
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True)
            sys.stdout.write(r.stdout)
            sys.stdout.write(r.stderr[-2000:])
            print("# exit code %d, %.1f s wall (includes the reference's single-thread CPU compute_reference_results)"
                  % (r.returncode, time.time() - t0), flush=True)
            rc = rc or r.returncode
    sys.exit(rc)


if __name__ == "__main__":
    main()
