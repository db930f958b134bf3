"""Same-box A/B (round 4): how the per-run helper launch (packed gather vector + hot table) is paid for on the
short streams.  Variants are planner overrides (GRAPHLILY_DEBUG, read at plan creation), all in one process on one box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io  # noqa: E402

VARIANTS = [("default", ""), ("self-hot, table <= 4096", "spmv_helper=2,spmv_hot=4096"), ("gather helper", "spmv_helper=0"),
            ("spread helper", "spmv_helper=1"), ("no packed vector", "spmv_compact=0")]


def main():
    global VARIANTS
    graphs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["googleplus", "ogbl_ppa", "pokec"]
    if len(sys.argv) > 2:      # "label:knobs;label:knobs"
        VARIANTS = [tuple(v.split(":")) for v in sys.argv[2].split(";")]
    iters = 200
    dev = torch.device("cuda:0")
    capi.init(0)
    capi.set_stream(torch.cuda.current_stream().cuda_stream)
    for name in graphs:
        m = datasets.paper_graph(name, 1.0, device=dev)
        io.util_round_csr_matrix_dim(m, 128, 8)
        pattern = len(sys.argv) > 3 and sys.argv[3] == "pattern"     # equal values: the 4-byte pattern layout
        m.adj_data = (np.full(m.nnz, np.float32(1.0 / m.num_rows), dtype=np.float32) if pattern
                      else np.random.default_rng(1).random(m.nnz, dtype=np.float32))
        alg = (4 if pattern else 8) * m.nnz + 4 * m.num_cols + 4 * m.num_rows
        x = torch.rand(m.num_cols, device=dev)
        y = torch.zeros(m.num_rows, device=dev)
        bx, by = capi.DeviceBuffer.from_torch(x), capi.DeviceBuffer.from_torch(y)
        ref = None
        for label, knobs in VARIANTS:
            os.environ["GRAPHLILY_DEBUG"] = knobs
            plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, 0, m.num_rows, 0)
            info = plan.info()
            best = []
            for rep in range(3):
                for _ in range(10):
                    plan.run(bx, None, by, 0, 0.0, 0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    plan.run(bx, None, by, 0, 0.0, 0)
                e1.record()
                torch.cuda.synchronize()
                best.append(e0.elapsed_time(e1) / iters)
            ms = sorted(best)[1]
            out = y.clone()
            if ref is None:
                ref = out
            same = bool(torch.allclose(out, ref, rtol=1e-5, atol=1e-6))
            print("%-11s %-26s %.4f ms  %5.1f %% of 8 TB/s  %s helper=%s hot=%d (%.0f %% of nnz) packed=%d units=%d  same=%s" %
                  (name, label, ms, alg / ms / 1e6 / 8000 * 100, info["layout"], info["helper"], info["hot_columns"],
                   100.0 * info["hot_nnz"] / max(info["nnz"], 1), info["packed_columns"], info["num_units"], same), flush=True)
            del plan
    os.environ["GRAPHLILY_DEBUG"] = ""


if __name__ == "__main__":
    main()
