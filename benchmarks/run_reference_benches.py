#!/usr/bin/env python
"""Run the REFERENCE's own benchmark drivers (benchmark/bench_{spmv,spmspv,bfs,pagerank,sssp}.cpp, compiled unmodified
against include/graphlily by `make -C oracle ref_benches`) on the HIP backend and print what they print.

The binaries take the reference's positional arguments (benchmark/run_bfs.sh:3-10):
    num_channels spmv_out_buf_len spmspv_out_buf_len vec_buf_len bitstream dataset num_iterations
The buffer sizes and the bitstream path are ignored by this backend.  The dataset is a stand-in written as
an uncompressed scipy npz (the loader takes stored and deflated members alike).

    python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank

bench_spmspv.cpp takes no dataset argument: it reads seven named matrices from a directory compiled into it (:25, :255-262).
`--apps spmspv` writes stand-ins under those names into that directory (created if absent -- on a scratch GPU box; an existing
directory that this script did not create is left alone), runs `bench_spmspv hw unused.xclbin <log>` -- 7 matrices x 7 vector
sparsities, each verified by the driver against the reference's own compute_reference_results -- and removes the files again.
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


# benchmark/bench_spmspv.cpp:25 and :255-262
REF_SPMSPV_DIR = "/work/shared/common/research/graphblas/data/sparse_matrix_graph/"
REF_SPMSPV_FILES = [("gplus_108K_13M_csr_float32.npz", "googleplus"), ("ogbl_ppa_576K_42M_csr_float32.npz", "ogbl_ppa"),
                    ("hollywood_1M_113M_csr_float32.npz", "hollywood"), ("pokec_1633K_31M_csr_float32.npz", "pokec"),
                    ("ogbn_products_2M_124M_csr_float32.npz", "ogbn_products"),
                    ("uniform_conflict_free_1M_64_csr_float32.npz", "uniform_conflict_free_1M_64"),
                    ("uniform_conflict_free_1M_256_csr_float32.npz", "uniform_conflict_free_1M_256")]
MARKER = ".written_by_graphlily_hip_run_reference_benches"


def run_spmspv(dev, keep):
    """The reference's bench_spmspv.cpp, unmodified, over stand-ins under the names and in the directory it has compiled in."""
    import scipy.sparse as sp
    from graphlily_amd import datasets
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_spmspv_on_hip")
    if not os.path.exists(exe):
        print("# %s not built (needs /root/reference: make -C oracle ref_benches)" % exe)
        return 2
    if os.path.isdir(REF_SPMSPV_DIR) and not os.path.exists(os.path.join(REF_SPMSPV_DIR, MARKER)):
        print("# %s exists and is not this script's: not touching it" % REF_SPMSPV_DIR)
        return 2
    os.makedirs(REF_SPMSPV_DIR, exist_ok=True)
    open(os.path.join(REF_SPMSPV_DIR, MARKER), "w").close()
    t0 = time.time()
    for fname, name in REF_SPMSPV_FILES:
        path = os.path.join(REF_SPMSPV_DIR, fname)
        if os.path.exists(path):
            continue
        if name.startswith("uniform_conflict_free_1M_"):     # column c holds rows (c + k (n / d + 1)) mod n, as in bench_spmspv.py
            n, d = 1 << 20, int(name.rsplit("_", 1)[1])
            rows = (np.arange(n, dtype=np.int64)[:, None] + np.arange(d, dtype=np.int64)[None, :] * (n // d + 1)) % n
            rows.sort(axis=1)
            A = sp.csc_matrix((np.ones(n * d, np.float32), rows.reshape(-1).astype(np.int32), (np.arange(n + 1, dtype=np.int64) * d).astype(np.int32)),
                              shape=(n, n)).tocsr()
            del rows
        else:
            m = datasets.paper_graph(name, 1.0, device=dev)
            A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)
            del m
        sp.save_npz(path, A, compressed=False)
        print("# %s: n=%d nnz=%d" % (fname, A.shape[0], A.nnz), flush=True)
        del A
    print("# stand-ins written in %.1f s" % (time.time() - t0), flush=True)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        log = os.path.join(tmp, "bench_spmspv.log")
        cmd = [exe, "hw", "unused.xclbin", log]
        print("# " + " ".join(cmd), flush=True)
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        out = r.stdout.splitlines()
        cut = [i for i, l in enumerate(out) if "Verification result" in l]
        sys.stdout.write("\n".join(out[cut[0]:] if cut else out[-60:]) + "\n")      # (the table; the per-case chatter is dropped)
        sys.stdout.write(r.stderr[-2000:])
        print("# exit code %d, %.1f s wall (49 cases: npz load + csr2csc + plan + the reference's CPU check per case, 20 timed runs each)"
              % (r.returncode, time.time() - t0), flush=True)
    if not keep:
        for fname, _ in REF_SPMSPV_FILES:
            try:
                os.remove(os.path.join(REF_SPMSPV_DIR, fname))
            except OSError:
                pass
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="googleplus")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--apps", default="spmv,spmv_verify,bfs,pagerank,sssp",
                    help="spmv = bench_spmv.cpp as it is (100 blocking runs, :96-112); spmv_verify = the same file with its own "
                         "verify (:15-33) called on the results (tests/cpp/ref_bench_spmv_verify.cpp)")
    ap.add_argument("--npz", default=None, help="an already written stand-in (skips generation)")
    ap.add_argument("--keep", action="store_true", help="spmspv: leave the stand-in files in the reference's dataset directory")
    args = ap.parse_args()
    import scipy.sparse as sp
    import torch
    from graphlily_amd import datasets
    g = datasets.PAPER_GRAPHS[args.graph]
    dev = torch.device("cuda:0") if torch.cuda.is_available() else None
    if args.apps == "spmspv":
        sys.exit(run_spmspv(dev, args.keep))
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
