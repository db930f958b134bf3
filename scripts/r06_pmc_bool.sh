# round 6: measured HBM traffic of the boolean layout's kernel (masked (||,&&) SpMV, orkut): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes (no tracing), then --kernel-trace --stats for its duration; bool_compress=0 gives the 4-byte stream for comparison
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for k in 1 0; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pb; mkdir -p /tmp/pb
    GRAPHLILY_DEBUG="bool_compress=$k" timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/pb -- python $R/scripts/probe_spmv.py --graph ${GRAPH:-orkut} --flags 2 --ops 1 --no-copy --iters 10 > /tmp/pb/out.txt 2>&1
    f=$(find /tmp/pb -name "*counter_collection.csv" | head -1)
    python - "$f" "$C" "$k" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spmv_bool_kernel" in r["Kernel_Name"]]
vals = [float(r["Counter_Value"]) for r in rows]
print("bool_compress=%s %s: %d launches of spmv_bool_kernel, mean %.1f KiB (units: 64 B for FETCH per the guide's gfx950 note -> see pmc_summary calibration)" % (sys.argv[3], sys.argv[2], len(vals), sum(vals) / max(len(vals), 1)))
PY
  done
  rm -rf /tmp/pb; mkdir -p /tmp/pb
  GRAPHLILY_DEBUG="bool_compress=$k" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python $R/scripts/probe_spmv.py --graph ${GRAPH:-orkut} --flags 2 --ops 1 --no-copy --iters 50 > /tmp/pb/out.txt 2>&1
  f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); grep "spmv_bool_kernel" "$f" | cut -c1-200 | sed "s/^/bool_compress=$k /"
  grep "plan create" /tmp/pb/out.txt | cut -c1-300
done
