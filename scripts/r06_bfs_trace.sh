# scratch (round 6): which kernels a BFS schedule on orkut spends its 0.245 ms in -- rocprofv3 --kernel-trace over a few blocking calls
# (the graph replays are traced kernel by kernel), summarised per kernel name and per position in one replay
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bfstrace; mkdir -p /tmp/bfstrace
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfstrace -- python $R/scripts/bfs_call_times.py --graph ${GRAPH:-orkut} --calls 6 --no-timed --modes ${MODES:-pull_push} > /tmp/bfstrace/out.txt 2>&1
f=$(find /tmp/bfstrace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay: walk back from the end to the previous bfs_bits_begin kernel
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "bfs_bits_begin" in n]
lo = starts[-1] if starts else max(0, len(rows) - 40)
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
print("last replay: %d kernels" % (len(rows) - lo))
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +gap %5.1f  dur %7.1f us  grid %8s wg %5s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"][:110]))
    prev_end = e
print("replay span %.1f us" % ((prev_end - t0) / 1e3))
PY
