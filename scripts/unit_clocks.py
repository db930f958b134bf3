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
