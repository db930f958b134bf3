# round 4: per-kernel durations of blocking SpMSpV calls (rocprofv3 kernel trace of scripts/spmspv_call_trace.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in "$@"; do set -- $c
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sp_trace && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_trace -- python $GRAFT_REPO_ROOT/scripts/spmspv_call_trace.py $1 $2 > /tmp/sp_trace.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== $1 $2"; grep "blocking\|enqueued" /tmp/sp_trace.log
  python - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/sp_trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-240:]                    # the back-to-back tail: steady state
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].replace("void ", "").replace("gl::", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print("   %-60s n=%3d median %7.1f us  min %7.1f" % (k, len(v), v[len(v) // 2], v[0]))
PY
done
