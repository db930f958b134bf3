"""Where the BFS wall goes after the schedule (orkut, one GPU): host-side timestamps around the phases of the packed
read-back.  usage: r03_bfs_readback_phases.py [bits]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import app, capi, datasets
m = datasets.paper_graph("orkut", 1.0, device=torch.device("cuda:0"))
deg = np.diff(m.adj_indptr.astype(np.int64)); src = 0 if deg[0] > 0 else int(np.argmax(deg > 0))
capi.init(0)
b = app.BFS(16, 0, 0, 0); b.set_up_runtime(); b.load_and_format_matrix(m, True); b.send_matrix_host_to_device()
for _ in range(5): b.pull_push(src, 6, 0.001)
st = b.bits_loop_; n = b.n_; cw = st["ctl_words"]
g = [v for k, v in st["graphs"].items() if v][0]
for bits in (8, 4):
    pw = capi.levels_packed_words(n, bits)
    dev = capi.DeviceBuffer(4 * (pw + cw)); h8 = capi.pinned_empty(4 * (pw + cw), np.uint8); res = capi.pinned_empty(n, np.float32)
    rows = []
    for rep in range(30):
        capi.sync(); t = [time.perf_counter()]
        g.launch(); t.append(time.perf_counter())
        capi.levels_pack(st["distance"], n, bits, st["ctl"], cw, dev); dev.read_async(h8); t.append(time.perf_counter())
        capi.host_threads_warm(); t.append(time.perf_counter())
        capi.sync(); t.append(time.perf_counter())
        capi.host_levels_unpack(res, h8, n, bits); t.append(time.perf_counter())
        rows.append(np.diff(t) * 1e6)
    r = np.median(np.array(rows), axis=0)
    print("bits %d: launch %.1f | pack+copy enqueue %.1f | warm %.1f | wait %.1f | unpack %.1f | total %.1f us" % (bits, *r, r.sum()))
# unpack alone, threads cold (after a pause) and warm
for pause in (0.0, 0.002):
    ts = []
    for rep in range(20):
        time.sleep(pause); t0 = time.perf_counter(); capi.host_levels_unpack(res, h8, n, 4); ts.append(time.perf_counter() - t0)
    print("unpack alone after %.0f ms pause: median %.1f us" % (pause * 1e3, np.median(ts) * 1e6))
import cProfile, pstats
ts = []
for rep in range(40):
    capi.sync(); t0 = time.perf_counter(); b.pull_push(src, 6, 0.001); ts.append(time.perf_counter() - t0)
print("BFS.pull_push wall: median %.1f us" % (np.median(ts) * 1e6))
pr = cProfile.Profile(); pr.enable()
for rep in range(200): b.pull_push(src, 6, 0.001)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
