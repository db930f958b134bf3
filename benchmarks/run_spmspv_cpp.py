#!/usr/bin/env python
"""The SpMSpV sparsity sweep (benchmark/bench_spmspv.cpp's protocol) through the C++ module layer: builds
benchmarks/bench_spmspv_cpp.cpp against include/graphlily + the HIP library, writes every stand-in of the sweep as an
uncompressed scipy npz and runs the binary on it.  What a C++ caller of the drop-in headers measures per blocking run() --
benchmarks/bench_spmspv.py is the same sweep through the Python mirror (three ctypes calls per run).

    python benchmarks/run_spmspv_cpp.py [--graphs googleplus,pokec] [--out profiles/rNN_spmspv_sweep_cpp.jsonl]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    exe = os.path.join(ROOT, "build", "bench_spmspv_cpp")
    lib = os.path.join(ROOT, "graphlily_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "benchmarks", "bench_spmspv_cpp.cpp"), "-o", exe, "-L", lib, "-lgraphlily_hip",
                           "-Wl,-rpath," + lib])
    return exe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="googleplus,ogbl_ppa,hollywood,pokec,ogbn_products,"
                                        "uniform_conflict_free_1M_64,uniform_conflict_free_1M_256")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import scipy.sparse as sp
    import torch
    from graphlily_amd import datasets
    exe = build()
    dev = torch.device("cuda:0") if torch.cuda.is_available() else None
    lines, rc = [], 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for name in args.graphs.split(","):
            if name.startswith("uniform_conflict_free_1M_"):     # as in bench_spmspv.py: column c holds rows (c + k (n / d + 1)) mod n
                n, d = 1 << 20, int(name.rsplit("_", 1)[1])
                rows = ((np.arange(n, dtype=np.int64)[:, None] + np.arange(d, dtype=np.int64)[None, :] * (n // d + 1)) % n)
                A = sp.csc_matrix((np.ones(n * d, np.float32), rows.reshape(-1).astype(np.int32), (np.arange(n + 1, dtype=np.int64) * d).astype(np.int32)),
                                  shape=(n, n)).tocsr()
                A.sort_indices()
                del rows
            else:
                m = datasets.paper_graph(name, 1.0, device=dev)
                A = sp.csr_matrix((m.adj_data, m.adj_indices.view(np.int32), m.adj_indptr.view(np.int32)), shape=(m.num_rows, m.num_cols),
                                  dtype=np.float32)
                del m
            path = os.path.join(tmp, "%s_csr_float32.npz" % name)
            sp.save_npz(path, A, compressed=False)
            del A
            r = subprocess.run([exe, name, path], capture_output=True, text=True)
            rc = rc or r.returncode
            for ln in r.stdout.splitlines():
                if ln.startswith("{"):
                    print(ln, flush=True)
                    lines.append(json.loads(ln))
            if r.returncode:
                sys.stderr.write(r.stdout[-1000:] + r.stderr[-2000:])
            os.remove(path)
    if lines:
        small = sorted(x["ms"] for x in lines if x["vector_sparsity"] >= 0.999)
        print("# %d cases, %d verified; >= 99.9 %%: median %.4f ms, min %.4f, max %.4f" %
              (len(lines), sum(x["verified"] for x in lines), small[len(small) // 2], small[0], small[-1]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            for x in lines:
                f.write(json.dumps(x) + "\n")
    sys.exit(rc)


if __name__ == "__main__":
    main()
