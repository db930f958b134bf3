#!/usr/bin/env python
"""Run the REFERENCE's own benchmark drivers (benchmark/bench_{spmv,spmspv,bfs,pagerank,sssp}.cpp, compiled unmodified
against include/graphlily by `make -C oracle ref_benches`) on the HIP backend and print what they print.

The binaries take the reference's positional arguments (benchmark/run_bfs.sh:3-10):
    num_channels spmv_out_buf_len spmspv_out_buf_len vec_buf_len bitstream dataset num_iterations
The buffer sizes and the bitstream path are ignored by this backend.  The dataset is a stand-in written as
an uncompressed scipy npz (the loader takes stored and deflated members alike).

    python benchmarks/run_reference_benches.py --graph orkut --apps bfs,pagerank

bench_spmspv.cpp takes no dataset argument: it reads seven named matrices from an ABSOLUTE directory compiled into it (:25,
:255-262).  `--apps spmspv --write-reference-dataset-dir` (the opt-in is required: the path lies outside any scratch root) writes
stand-ins under those names into that directory -- only if the directory does not exist yet or carries this script's marker --
runs `bench_spmspv hw unused.xclbin <log>` -- 7 matrices x 7 vector sparsities, each verified by the driver against the
reference's own compute_reference_results -- and removes the files, the marker and every directory level it created again
(`--keep` leaves them; files found there from an earlier `--keep` run are always REGENERATED, never trusted).

`--apps tests --write-reference-dataset-dir` does the same for the reference's own ACCEPTANCE SUITES (tests/test_module_apply.cpp,
tests/test_module_spmv_spmspv.cpp, tests/test_app.cpp, compiled unmodified by `make -C oracle ref_tests`), whose sources name
dense_32 / dense_1K / uniform_10K_10 / gplus_108K_13M under /work/shared/common/project_build/...: float and ap_ufixed<32, 8>.
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


# tests/test_module_spmv_spmspv.cpp:144-145,167-168,251-266 and tests/test_app.cpp:57-58,91-92,111-112
REF_TESTS_DIR = "/work/shared/common/project_build/graphblas/data/sparse_matrix_graph/"
REF_TESTS_FILES = [("dense_32_csr_float32.npz", "dense_32"), ("dense_1K_csr_float32.npz", "dense_1K"),
                   ("uniform_10K_10_csr_float32.npz", "uniform_10K_10"), ("gplus_108K_13M_csr_float32.npz", "googleplus")]


def _stand_in(name, dev):
    """The stand-in of one of the reference's named matrices as a scipy CSR (the files live on the authors' NFS)."""
    import scipy.sparse as sp
    from graphlily_amd import datasets
    if name.startswith("uniform_conflict_free_1M_"):     # column c holds rows (c + k (n / d + 1)) mod n, as in bench_spmspv.py
        n, d = 1 << 20, int(name.rsplit("_", 1)[1])
        rows = (np.arange(n, dtype=np.int64)[:, None] + np.arange(d, dtype=np.int64)[None, :] * (n // d + 1)) % n
        rows.sort(axis=1)
        return sp.csc_matrix((np.ones(n * d, np.float32), rows.reshape(-1).astype(np.int32), (np.arange(n + 1, dtype=np.int64) * d).astype(np.int32)),
                             shape=(n, n)).tocsr()
    if name == "dense_32":
        m = datasets.dense(32)
    elif name == "dense_1K":
        m = datasets.dense(1024)
    elif name == "uniform_10K_10":
        m = datasets.uniform(10000, 10, seed=7)
    else:
        m = datasets.paper_graph(name, 1.0, device=dev)
    return sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols), dtype=np.float32)


class ReferenceDatasetDir:
    """The ABSOLUTE dataset directory one of the reference's drivers has compiled in, filled with stand-ins for the duration of a
    `with` block.  Refuses without the explicit opt-in (the path lies outside any scratch root) and when the directory exists
    without this script's marker; files left by an earlier --keep run are regenerated, never trusted; on exit the files, the
    marker and every directory level THIS run created are removed again (unless keep)."""

    def __init__(self, path, files, dev, allow_abs, keep):
        self.path, self.files, self.dev, self.allow_abs, self.keep, self.made = path, files, dev, allow_abs, keep, []

    def __enter__(self):
        import scipy.sparse as sp
        if not self.allow_abs:
            raise PermissionError("the reference's driver reads %s (compiled in): pass --write-reference-dataset-dir to let this script "
                                  "create it, fill it with stand-ins and remove it again" % self.path)
        if os.path.isdir(self.path) and not os.path.exists(os.path.join(self.path, MARKER)):
            raise PermissionError("%s exists and is not this script's: not touching it" % self.path)
        d = self.path.rstrip("/")
        while d and not os.path.isdir(d):     # the directory levels this run creates (deepest first here)
            self.made.append(d)
            d = os.path.dirname(d)
        os.makedirs(self.path, exist_ok=True)
        open(os.path.join(self.path, MARKER), "w").close()
        t0 = time.time()
        for fname, name in self.files:
            path = os.path.join(self.path, fname)
            if os.path.exists(path):          # left by an earlier --keep run: the generator or its arguments may have changed since
                os.remove(path)
            A = _stand_in(name, self.dev)
            sp.save_npz(path, A, compressed=False)
            print("# %s: n=%d nnz=%d" % (fname, A.shape[0], A.nnz), flush=True)
            del A
        print("# stand-ins written in %.1f s" % (time.time() - t0), flush=True)
        return self

    def __exit__(self, *exc):
        if self.keep:
            return False
        for fname in [f for f, _ in self.files] + [MARKER]:
            try:
                os.remove(os.path.join(self.path, fname))
            except OSError:
                pass
        for d in self.made:                   # deepest first; anything non-empty stays
            try:
                os.rmdir(d)
            except OSError:
                break
        return False


def run_spmspv(dev, keep, allow_abs):
    """The reference's bench_spmspv.cpp, unmodified, over stand-ins under the names and in the directory it has compiled in."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_spmspv_on_hip")
    if not os.path.exists(exe):
        print("# %s not built (needs /root/reference: make -C oracle ref_benches)" % exe)
        return 2
    try:
        with ReferenceDatasetDir(REF_SPMSPV_DIR, REF_SPMSPV_FILES, dev, allow_abs, keep), tempfile.TemporaryDirectory(dir="/tmp") as tmp:
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
            return r.returncode
    except PermissionError as e:
        print("# " + str(e))
        return 2


REF_TEST_SUITES = ["test_module_apply", "test_module_spmv_spmspv", "test_app"]


def run_tests(dev, keep, allow_abs, variants=("", "_ufixed")):
    """The reference's own acceptance suites -- tests/test_module_apply.cpp, tests/test_module_spmv_spmspv.cpp, tests/test_app.cpp,
    compiled UNMODIFIED against include/ (`make -C oracle ref_tests`: include/ap_fixed.h, tests/cpp/gtest/gtest.h) -- for
    val_t = float and for the reference's shipped ap_ufixed<32, 8>, over stand-ins under the names and in the directory the sources
    have compiled in.  Prints every suite's RUN / OK / FAILED lines; returns the number of failing binaries."""
    exes = [(s + v, os.path.join(ROOT, "oracle", "_ref", "ref_" + s + v)) for v in variants for s in REF_TEST_SUITES]
    missing = [e for _, e in exes if not os.path.exists(e)]
    if missing:
        print("# not built (needs /root/reference: make -C oracle ref_tests): %s" % ", ".join(missing))
        return 2
    bad = 0
    try:
        with ReferenceDatasetDir(REF_TESTS_DIR, REF_TESTS_FILES, dev, allow_abs, keep), tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for name, exe in exes:
                t0 = time.time()
                r = subprocess.run([exe], capture_output=True, text=True, cwd=tmp)     # (the Clean fixtures rm -rf ./proj: inside tmp)
                lines = [l for l in r.stdout.splitlines() if l.startswith("[") or "Failure" in l or "Which is" in l or "Actual" in l]
                print("## %s (exit code %d, %.1f s)" % (name, r.returncode, time.time() - t0))
                sys.stdout.write("\n".join(lines) + "\n" + r.stderr[-1500:])
                sys.stdout.flush()
                bad += r.returncode != 0
    except PermissionError as e:
        print("# " + str(e))
        return 2
    print("# %d of %d suite binaries passed" % (len(exes) - bad, len(exes)))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="googleplus")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--apps", default="spmv,spmv_verify,bfs,pagerank,sssp",
                    help="spmv = bench_spmv.cpp as it is (100 blocking runs, :96-112); spmv_verify = the same file with its own "
                         "verify (:15-33) called on the results (tests/cpp/ref_bench_spmv_verify.cpp)")
    ap.add_argument("--npz", default=None, help="an already written stand-in (skips generation)")
    ap.add_argument("--keep", action="store_true", help="spmspv: leave the stand-in files in the reference's dataset directory")
    ap.add_argument("--write-reference-dataset-dir", action="store_true",
                    help="spmspv: allow creating / filling the absolute dataset directory bench_spmspv.cpp has compiled in")
    args = ap.parse_args()
    import scipy.sparse as sp
    import torch
    from graphlily_amd import datasets
    g = datasets.PAPER_GRAPHS[args.graph]
    dev = torch.device("cuda:0") if torch.cuda.is_available() else None
    if args.apps == "spmspv":
        sys.exit(run_spmspv(dev, args.keep, args.write_reference_dataset_dir))
    if args.apps == "tests":
        sys.exit(run_tests(dev, args.keep, args.write_reference_dataset_dir))
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
