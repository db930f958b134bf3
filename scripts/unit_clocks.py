"""scratch: summarise a GRAPHLILY_SPMV_CLOCKS dump (wall_clock64 ticks at 100 MHz).
Columns: unit, 5 stamps (entry, prologue done, wave 0's loop done, all waves done, end),
hardware id (HW_ID | XCC_ID << 32), #cold groups, #hot groups, #rows, #hub rows."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
st = a[:, 1:6].astype(np.float64) / 100.0   # us
cold, hot, rows, hubs = (a[:, k].astype(float) for k in (7, 8, 9, 10))
hw = a[:, 6]
xcc, cu, sh, se = (hw >> 32) & 0xf, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
t_first = st[:, 0].min()
print("units %d  kernel span %.1f us  first entry spread %.1f us  last unit entry at %.1f us" %
      (len(a), st[:, 4].max() - t_first, np.sort(st[:, 0])[min(255, len(a) - 1)] - t_first, st[:, 0].max() - t_first))
names = ["prologue", "loop (wave 0)", "wait for all waves", "epilogue"]
for k, nm in enumerate(names):
    d = st[:, k + 1] - st[:, k]
    print("  %-26s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (nm, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
dur = st[:, 4] - st[:, 0]
print("  %-26s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % ("whole unit", dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
A = np.stack([cold, hot, np.ones_like(cold)], axis=1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print("fit unit = %.4f us * cold_groups + %.4f us * hot_groups + %.1f us ; residual rms %.2f us" %
      (coef[0], coef[1], coef[2], float(np.sqrt(np.mean((dur - A @ coef) ** 2)))))
print("ideal at 45 clk / 512 B stream: %.1f us per unit (mean groups %.0f)" % ((cold + hot).mean() * 45 / 2400.0, (cold + hot).mean()))
loop = st[:, 3] - st[:, 1]   # prologue done -> all waves done
per = loop / np.maximum(cold + hot, 1) * 2400.0   # clocks per group at 2.4 GHz
print("loop clocks per group: mean %.1f  min %.1f  max %.1f" % (per.mean(), per.min(), per.max()))
for nm, key in (("xcc", xcc), ("se", se), ("cu", cu)):
    vals = sorted(set(key.tolist()))
    print("  by %s: " % nm + "  ".join("%d: %.1f (n=%d)" % (v, per[key == v].mean(), (key == v).sum()) for v in vals))
print("corr(clocks per group, cold share) %.2f   corr(.., rows) %.2f   corr(.., unit index) %.2f" %
      (np.corrcoef(per, cold / np.maximum(cold + hot, 1))[0, 1], np.corrcoef(per, rows)[0, 1], np.corrcoef(per, a[:, 0])[0, 1]))
