"""Timeline of the LAST call traced by scripts/r03_emulate_trace.py: every kernel and copy from the last
bfs_bits_begin_kernel on, with start offset, duration and the gap to the previous activity (microseconds)."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", ""), r.get("Size", ""))))
ev.sort()
begins = [e for e in ev if "bfs_bits_begin_kernel" in e[2]]
t0 = begins[-1][0]
ks = [e for e in ev if e[0] >= t0]
prev = ks[0][0]
busy = 0.0
for e in ks:
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3, e[2]))
    prev = e[1]
    busy += (e[1] - e[0]) / 1e3
kern = [e for e in ks if not e[2].startswith("COPY")]
print("total %.1f us first start -> last end; %d kernels, %.1f us in kernels (rocprofv3 durations), %d copies" % (
    (ks[-1][1] - t0) / 1e3, len(kern), sum((e[1] - e[0]) / 1e3 for e in kern), len(ks) - len(kern)))
