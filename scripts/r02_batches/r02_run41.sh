cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_pull_trace.py orkut > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log || tail -20 /tmp/bfs_trace.log
python scripts/r02_timeline.py /tmp/bfs_trace
export GRAPHLILY_BFS_BITS=0
cd /tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_pull_trace.py orkut > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log || tail -20 /tmp/bfs_trace.log
python - <<'PY'
import csv, glob
kern=[]
for f in glob.glob("/tmp/bfs_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
kern.sort()
for s,e,n in kern[-12:]:
    print("%10.1f dur %8.1f %s" % ((s-kern[-12][0])/1e3, (e-s)/1e3, n))
PY
