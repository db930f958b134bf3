"""Print the tail of a rocprofv3 hip-runtime + kernel + memory-copy trace as one merged timeline (ms)."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api  " + r["Function"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERN " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", "")))
ev.sort()
# the last ~30 ms before the final pull-push phase is too much; print the window around the LAST 'spmv_bool' heavy burst: take last 400 events
t_end = ev[-1][1]
sel = [e for e in ev if e[0] > t_end - 40_000_000]
t0 = sel[0][0]
for s, e, n in sel:
    if (e - s) > 20000 or n.startswith("KERN") or n.startswith("COPY"):
        print("%9.3f +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
