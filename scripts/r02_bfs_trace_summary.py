"""Summarise scripts/r02_bfs_trace.py's rocprofv3 output: for the LAST pull_push call, every memory copy and the number
of kernels between the first kernel and the last copy."""
import csv, glob, re, sys
d, log = sys.argv[1], sys.argv[2]
calls = [tuple(int(x) for x in re.findall(r"CALL \d+ (\d+) (\d+)", l)[0]) for l in open(log) if l.startswith("CALL")]
kern, cop = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), r.get("Size", "")))
kern.sort(); cop.sort()
# rocprofv3 timestamps and time.time_ns() share CLOCK_REALTIME only approximately: locate the last call by its kernels --
# the last bfs_begin_kernel starts it
begins = [k for k in kern if "bfs_begin_kernel" in k[2] or "bfs_bits_begin_kernel" in k[2]]
t0 = begins[-1][0]
ks = [k for k in kern if k[0] >= t0]
cs = [c for c in cop if c[0] >= t0 - 200000]
print("last pull_push call: %d kernels in %.3f ms (first kernel start -> last kernel end)" % (len(ks), (ks[-1][1] - ks[0][0]) / 1e6))
for c in cs:
    where = "BEFORE the first kernel" if c[1] <= ks[0][0] else ("AFTER the last kernel" if c[0] >= ks[-1][1] else "BETWEEN kernels")
    print("  copy %-14s %10s bytes  %+9.3f ms from the first kernel, %.3f ms long  -- %s" % (c[2], c[3], (c[0] - ks[0][0]) / 1e6, (c[1] - c[0]) / 1e6, where))
between = [c for c in cs if not (c[1] <= ks[0][0] or c[0] >= ks[-1][1])]
print("copies between the first and the last kernel of the call: %d" % len(between))
from collections import Counter
for name, cnt in Counter(k[2].split("(")[0] for k in ks).most_common():
    print("  %3d x %s" % (cnt, name))
