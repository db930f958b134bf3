"""Per-kernel timeline of the LAST BFS.pull_push call in a rocprofv3 kernel trace (scripts/r02_bfs_trace.py):
start offset, duration and the gap to the previous kernel's end, in microseconds."""
import csv, glob, sys
d = sys.argv[1]
kern = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("gl::", "")[:64]))
kern.sort()
begins = [k for k in kern if "bfs_begin_kernel" in k[2] or "bfs_bits_begin_kernel" in k[2]]
t0 = begins[-1][0]
ks = [k for k in kern if k[0] >= t0]
prev = ks[0][0]
busy = 0
for s, e, n in ks:
    print("%9.1f  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    busy += e - s
    prev = e
print("total %.1f us, busy %.1f us, gaps %.1f us, %d kernels" % ((ks[-1][1] - t0) / 1e3, busy / 1e3, (ks[-1][1] - t0 - busy) / 1e3, len(ks)))
