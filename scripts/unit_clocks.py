"""scratch: summarise a GRAPHLILY_SPMV_CLOCKS dump (wall_clock64 ticks at 100 MHz)"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
t0, t1 = a[:, 1], a[:, 2]
dur = (t1 - t0) / 100.0   # us
span = (t1.max() - t0.min()) / 100.0
print("units %d  span %.1f us  dur mean %.1f  min %.1f  p50 %.1f  p90 %.1f  max %.1f us  start spread %.1f us" %
      (len(a), span, dur.mean(), dur.min(), np.median(dur), np.percentile(dur, 90), dur.max(), (t0.max() - t0.min()) / 100.0))
order = np.argsort(dur)[::-1][:5]
print("slowest units:", [(int(a[i, 0]), round(float(dur[i]), 1)) for i in order])

if a.shape[1] >= 7:
    cold, hot, rows, hubs = (a[:, k].astype(float) for k in (3, 4, 5, 6))
    A = np.stack([cold, hot, np.ones_like(cold)], axis=1)
    coef, res, *_ = np.linalg.lstsq(A, dur, rcond=None)
    pred = A @ coef
    print("fit dur = %.4f us * cold_groups + %.4f us * hot_groups + %.1f us ; residual rms %.2f us (dur std %.2f)" %
          (coef[0], coef[1], coef[2], float(np.sqrt(np.mean((dur - pred) ** 2))), float(dur.std())))
    print("corr(dur, cold) %.2f  corr(dur, hot) %.2f  corr(dur, rows) %.2f  corr(dur, hubs) %.2f" %
          tuple(float(np.corrcoef(dur, v)[0, 1]) for v in (cold, hot, rows, hubs)))
    for i in order:
        print("  unit %d: %.1f us cold %d hot %d rows %d hubs %d" % (int(a[i, 0]), dur[i], cold[i], hot[i], rows[i], hubs[i]))
