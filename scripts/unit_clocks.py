"""Scratch: what every unit (row block) of the main SpMV launch takes, run after run -- how much of the launch's tail
(max / mean of the units' durations) repeats, i.e. what re-cutting the row blocks from MEASURED per-unit times could take back.
Needs a -DGL_UNIT_CLOCKS build:  bash scripts/build_variant.sh WORK clocks -DGL_UNIT_CLOCKS
    GRAPHLILY_HIP_LIB=scripts/_variants/clocks.so python scripts/unit_clocks.py orkut [flags]      (flags 4 = general, 0 = pattern)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlily_amd import capi, datasets, io  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "orkut"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else capi.GL_PLAN_KEEP_VALUES
dev = torch.device("cuda:0")
capi.init(0)
m = datasets.paper_graph(name, 1.0, device=dev)
m.adj_data = np.full(m.nnz, np.float32(1.0 / m.num_rows), np.float32)
io.util_round_csr_matrix_dim(m, 128, 8)
plan = capi.SpMVPlan(m.num_rows, m.num_cols, m.adj_indptr, m.adj_indices, m.adj_data, flags=flags)
info = plan.info()
U = info["num_units"]
x = torch.randint(0, 2, (m.num_cols,), device=dev).float()
y = torch.zeros(m.num_rows, device=dev)
bx, by = capi.DeviceBuffer.from_torch(x), capi.DeviceBuffer.from_torch(y)
L = ctypes.CDLL(capi.LIB_PATH)
buf = np.zeros(2 * 4096, np.uint64)
for _ in range(30):
    plan.run(bx, None, by, 0, 0.0, 0)
capi.sync()
runs = []
for _ in range(24):
    for _ in range(3):
        plan.run(bx, None, by, 0, 0.0, 0)
    capi.sync()
    assert L.gl_debug_unit_clocks(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    c = buf.astype(np.int64).reshape(-1, 2)[:U]
    runs.append(np.stack([(c[:, 0] - c[:, 0].min()) / 100.0, (c[:, 1] - c[:, 0]) / 100.0], axis=1))
r = np.stack(runs)                      # runs x units x {start, duration} in us
dur = r[:, :, 1]
span = (r[:, :, 0] + r[:, :, 1]).max(axis=1)
print("%s %s layout, %d units: launch span %.1f us (median of %d runs), unit duration mean %.1f, per-run max / mean %.4f"
      % (name, info["layout"], U, np.median(span), len(runs), dur.mean(), float(np.median(dur.max(axis=1) / dur.mean(axis=1)))))
med = np.median(dur, axis=0)            # what repeats: a unit's median over the runs
noise = dur - med[None, :]
print("  units' MEDIAN durations: max / mean %.4f (the part of the tail that repeats), std %.2f us; run-to-run std of a unit around "
      "its median %.2f us" % (med.max() / med.mean(), med.std(), noise.std()))
# what a perfect re-cut of the repeatable part would leave: every unit at the mean of the medians + its run-to-run noise
ideal = (med.mean() + noise).max(axis=1)
print("  launch ends (units only) at %.1f us now; with the repeatable part cut away and the same noise: %.1f us (%.1f %% less)"
      % (np.median(dur.max(axis=1)), np.median(ideal), 100.0 * (1.0 - np.median(ideal) / np.median(dur.max(axis=1)))))
half = len(runs) // 2
a, b = np.median(dur[:half], axis=0), np.median(dur[half:], axis=0)
print("  correlation of the units' medians between the first and the second half of the runs: %.2f" % float(np.corrcoef(a, b)[0, 1]))
order = np.argsort(-med)[:6]
print("  slowest units (median us):", ", ".join("%d: %.1f" % (int(u), med[u]) for u in order), " fastest:", ", ".join("%d: %.1f" % (int(u), med[u]) for u in np.argsort(med)[:4]))

# who are the slow units?  the longest rows of the matrix and the units that hold them
try:
    uw = plan.export("units").reshape(-1, 12)[:U]            # 3 x uint4 per unit: {g0, ncold, first row, rows | flags}, {hub off, hubs, hot groups, seg}, ...
    first, nrows, nhub = uw[:, 2].astype(np.int64), (uw[:, 3] & 0xffff).astype(np.int64), uw[:, 5].astype(np.int64)
    deg = np.diff(m.adj_indptr.astype(np.int64))
    ent = np.array([int(m.adj_indptr[min(f + r, m.num_rows)]) - int(m.adj_indptr[f]) for f, r in zip(first, nrows)])
    top = np.argsort(-deg)[:5]
    for rr in top:
        u = int(np.searchsorted(first, rr, side="right") - 1)
        print("  row %d: %d entries = %.1f %% of unit %d's %d (unit median %.1f us = %+.1f %% of the mean, %d hub rows)"
              % (rr, deg[rr], 100.0 * deg[rr] / max(ent[u], 1), u, ent[u], med[u], 100.0 * (med[u] / med.mean() - 1.0), nhub[u]))
    for u in order[:3]:
        rows = np.arange(first[u], min(first[u] + nrows[u], m.num_rows))
        big = rows[np.argmax(deg[rows])]
        print("  slow unit %d: %d rows, %d entries, longest row %d entries (%.1f %%), %d hub rows" % (u, nrows[u], ent[u], deg[big], 100.0 * deg[big] / max(ent[u], 1), nhub[u]))
except Exception as e:     # (the export's layout is an implementation detail: this part may rot)
    print("  (units export not understood: %r)" % (e,))

# round 6: what a unit's duration follows -- its cold groups and its hot elements (a least-squares fit over the units): is a cut by
# COST (cold and hot entries priced differently) worth more than the cut by entries?
try:
    ncold, nhot = uw[:, 1].astype(np.float64), uw[:, 6].astype(np.float64)
    A = np.stack([ncold, nhot, np.ones_like(ncold)], axis=1)
    coef, *_ = np.linalg.lstsq(A, med, rcond=None)
    fit = A @ coef
    print("  fit: unit us = %.5f x cold groups + %.5f x hot %s + %.2f; residual std %.2f us; cold groups per unit min / mean / max %d / %d / %d, "
          "hot %d / %d / %d" % (coef[0], coef[1], "elements" if info["layout"] == "pattern" else "groups", coef[2], float((med - fit).std()),
                                ncold.min(), ncold.mean(), ncold.max(), nhot.min(), nhot.mean(), nhot.max()))
    print("  the fit's own max / mean over the units: %.4f (what a cut by this cost model could remove of the tail)" % float(fit.max() / fit.mean()))
except Exception as e:
    print("  (fit failed: %r)" % (e,))
