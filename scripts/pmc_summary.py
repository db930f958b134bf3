"""Summarise the rocprofv3 passes of bench.py into profiles/ (tracked).

Inputs (written by the gpurun command documented in profiles/README.md):
  gpurun_out/prof/*kernel_stats.csv      rocprofv3 --kernel-trace --stats
  gpurun_out/prof/pmc_fetch_rbcs.csv     rocprofv3 --pmc FETCH_SIZE   (rows of spmv_rbcs_kernel only)
  gpurun_out/prof/pmc_write_rbcs.csv     rocprofv3 --pmc WRITE_SIZE
HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE reports half of the bytes of a coalesced streaming read, so
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
The x2 factor is re-checked against a kernel with a known byte count when a calibration pass is present.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
GRAPH = sys.argv[2] if len(sys.argv) > 2 else "orkut"      # a second graph's passes: profile_bench.sh with GRAPH=<name>
SRC = os.path.join(ROOT, "gpurun_out", "prof" if GRAPH == "orkut" else "prof_" + GRAPH)
if GRAPH != "orkut":
    TAG = TAG + "_" + GRAPH


def mean_counter(path, kernel_substr):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    vals = vals[1:] if len(vals) > 2 else vals   # drop the first (cold) launch
    return sum(vals) / len(vals), len(vals)


def main():
    os.makedirs(DST, exist_ok=True)
    out = {"tag": TAG, "graph": GRAPH, "n_gpus": 1}
    stats = glob.glob(os.path.join(SRC, "*kernel_stats.csv")) + glob.glob(os.path.join(SRC, ".*kernel_stats.csv"))
    if stats:
        dst = os.path.join(DST, "%s_bench_kernel_stats.csv" % TAG)
        with open(stats[0]) as f, open(dst, "w") as g:   # this library's kernels only (+ header)
            for i, line in enumerate(f):
                if i == 0 or "gl::" in line:
                    g.write(line)
        with open(stats[0]) as f:
            for row in csv.DictReader(f):
                if "spmv_rbcs_kernel<0, 0, 1," in row["Name"]:
                    out["spmv_rbcs_kernel_avg_ns"] = float(row["AverageNs"])
                    out["spmv_rbcs_kernel_calls"] = int(row["Calls"])
    for name in ("pmc_fetch_rbcs.csv", "pmc_write_rbcs.csv"):
        p = os.path.join(SRC, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, "%s_%s" % (TAG, name)))
    fetch, nf = mean_counter(os.path.join(SRC, "pmc_fetch_rbcs.csv"), "spmv_rbcs_kernel<0, 0, 1,")
    write, nw = mean_counter(os.path.join(SRC, "pmc_write_rbcs.csv"), "spmv_rbcs_kernel<0, 0, 1,")
    out.update({"FETCH_SIZE_KiB_mean": fetch, "WRITE_SIZE_KiB_mean": write, "launches_averaged": [nf, nw],
                "fetch_correction": 2.0,
                "spmv_rbcs_kernel_bytes_per_launch": int((2.0 * fetch + write) * 1024)})
    # the other SpMV kernels of the same run (pattern leg, BFS leg), same correction
    fs, ws = os.path.join(SRC, "pmc_fetch_spmv.csv"), os.path.join(SRC, "pmc_write_spmv.csv")
    if os.path.exists(fs) and os.path.exists(ws):
        shutil.copy(fs, os.path.join(DST, "%s_pmc_fetch_spmv.csv" % TAG))
        shutil.copy(ws, os.path.join(DST, "%s_pmc_write_spmv.csv" % TAG))
        other = {}
        for k in ("spmv_rbcs_kernel<0, 0, 3,", "spmv_bool_kernel<0, 4, 1>", "spmv_bool_kernel<0, 4, 0>", "spmv_prescale_kernel", "spmv_bool_pack_kernel"):
            try:
                f_, n1 = mean_counter(fs, k)
                w_, n2 = mean_counter(ws, k)
                other[k] = {"FETCH_SIZE_KiB_mean": f_, "WRITE_SIZE_KiB_mean": w_, "launches_averaged": [n1, n2],
                            "bytes_per_launch": int((2.0 * f_ + w_) * 1024)}
            except ZeroDivisionError:
                pass
        out["other_kernels"] = other
    if stats:
        with open(stats[0]) as f:
            rows = {}
            for row in csv.DictReader(f):
                for k in ("spmv_rbcs_kernel<0, 0, 3,", "spmv_bool_kernel<0, 4, 1>", "spmv_bool_kernel<0, 4, 0>", "spmv_prescale_kernel", "spmv_bool_pack_kernel",
                          "spmv_hot_gather_kernel", "spmspv_bin_kernel<0>", "spmspv_fold_kernel<0>", "spmspv_work_kernel", "bfs_shard_step_kernel"):
                    if k in row["Name"]:
                        rows[k] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
            out["kernel_avg_ns"] = rows
    calib = os.path.join(SRC, "pmc_fetch_calib.csv")
    if os.path.exists(calib):
        c, n = mean_counter(calib, "k_stream<0, 4>")
        out["calibration"] = {"kernel": "ubench k_stream<0,4> (8 B/lane nt loads of exactly 1 GiB)",
                              "FETCH_SIZE_KiB_mean": c, "expected_KiB": 1 << 20,
                              "measured_over_expected": c / float(1 << 20)}
        shutil.copy(calib, os.path.join(DST, "%s_pmc_fetch_calib.csv" % TAG))
    # (bench.py reads pmc_traffic.json for the bench graph; another graph's summary gets its own file)
    with open(os.path.join(DST, "pmc_traffic.json" if GRAPH == "orkut" else "%s_pmc_traffic.json" % TAG), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
