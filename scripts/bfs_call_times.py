"""scratch (round 6): the distribution of blocking BFS calls -- bench.py's bfs_pull_push_ms read 0.39 ms in one process and 0.46 in
the next on one box.  Prints every call's wall time for `--calls` consecutive calls after the warm-up, per mode.
usage: python scripts/bfs_call_times.py [--graph orkut] [--calls 60]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _cpu():
    import ctypes
    return ctypes.CDLL(None).sched_getcpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="orkut")
    ap.add_argument("--calls", type=int, default=60)
    ap.add_argument("--near", action="store_true", help="bind the process to the CPUs of the GPU's NUMA node first")
    ap.add_argument("--modes", default="pull_push,pull")
    ap.add_argument("--no-timed", action="store_true", help="skip the calls that time the schedule alone (their read-back is a launch of its own)")
    args = ap.parse_args()
    import torch
    from graphlily_amd import app, capi, datasets
    dev = torch.device("cuda:0")
    capi.init(0)
    raw = datasets.paper_graph(args.graph, 1.0, device=dev)
    iters = datasets.PAPER_GRAPHS[args.graph]["iters"]
    bfs = app.BFS(16, 0, 0, 0)
    bfs.set_up_runtime()
    bfs.load_and_format_matrix(raw, True)
    bfs.send_matrix_host_to_device()
    src = int(np.argmax(np.diff(raw.adj_indptr.astype(np.int64)) > 0))
    # where this process runs relative to the GPU: its CPU now, the CPUs it may use, the GPU's NUMA node and that node's CPUs
    pr = torch.cuda.get_device_properties(0)
    try:
        dev_dir = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(dev_dir + "/numa_node").read())
        cpus = open("/sys/devices/system/node/node%d/cpulist" % max(node, 0)).read().strip()
    except Exception as e:      # noqa
        node, cpus = "?", repr(e)
    if args.near and isinstance(node, int) and node >= 0:
        want = set()
        for part in cpus.split(","):
            a, _, b = part.partition("-")
            want.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, want & os.sched_getaffinity(0))
    print("cpu now %d of %d allowed; GPU numa node %s (cpus %s); nodes: %s" % (
        _cpu(), len(os.sched_getaffinity(0)), node, cpus,
        " ".join(sorted(d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")))), flush=True)
    for mode in args.modes.split(","):
        fn = (lambda: bfs.pull_push(src, iters, 0.001)) if mode == "pull_push" else (lambda: bfs.pull(src, iters))
        for _ in range(14):
            fn()
        ts = []
        for _ in range(args.calls):
            capi.sync()
            t0 = time.perf_counter()
            fn()
            capi.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        bfs.time_schedule_ = not args.no_timed
        sched = [0.0]
        for _ in range(0 if args.no_timed else 8):
            fn()
            sched.append(bfs.schedule_ms_)
        bfs.time_schedule_ = False
        print("%s %s: schedule GPU time (8 calls) median %.3f ms; cpu now %d" % (args.graph, mode, float(np.median(sched)), _cpu()))
        a = np.array(ts)
        print("%s %s: median %.3f min %.3f p90 %.3f max %.3f ms | %s" % (args.graph, mode, np.median(a), a.min(), np.percentile(a, 90), a.max(),
              " ".join("%.3f" % v for v in a)), flush=True)


if __name__ == "__main__":
    main()
