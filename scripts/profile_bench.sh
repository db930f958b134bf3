#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; leaves small CSVs under gpurun_out/prof (see profiles/README.md).
# Separate runs: --kernel-trace --stats, then one --pmc counter per run (never combined with tracing).
# GRAPH=<stand-in> profiles another graph's SpMV legs only (into gpurun_out/prof_<graph>; scripts/pmc_summary.py <tag> <graph>).
R=${GRAFT_REPO_ROOT:-/root/repo}
G=${GRAPH:-orkut}
if [ "$G" = "orkut" ]; then OUT=$R/gpurun_out/prof; XTRA=""; else OUT=$R/gpurun_out/prof_$G; XTRA="--graph $G --no-bfs --no-spmspv"; fi
XTRA="$XTRA --no-six-graphs"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/trace -- python $R/bench.py --steps 50 --no-cpu-baseline $XTRA > $OUT/trace_bench.json 2> /tmp/prof/trace.err
f=$(find /tmp/prof/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/prof/pmc_$C -- python $R/bench.py --steps 10 --no-cpu-baseline --bfs-runs 2 $XTRA > /tmp/prof/pmc_$C.json 2> /tmp/prof/pmc_$C.err
  f=$(find /tmp/prof/pmc_$C -name "*counter_collection.csv" | head -1)
  n=$(echo $C | tr A-Z a-z | sed 's/_size//')
  if [ -n "$f" ]; then head -1 "$f" > $OUT/pmc_${n}_spmv.csv; grep "spmv_" "$f" >> $OUT/pmc_${n}_spmv.csv; grep "spmv_rbcs_kernel<0, 0, 1," $OUT/pmc_${n}_spmv.csv > /dev/null && { head -1 "$f" > $OUT/pmc_${n}_rbcs.csv; grep "spmv_rbcs_kernel<0, 0, 1," "$f" >> $OUT/pmc_${n}_rbcs.csv; }; fi
done
if [ -x $R/build/ubench_gather ]; then
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/calib -- $R/build/ubench_gather > /tmp/prof/calib.log 2>&1
  f=$(find /tmp/prof/calib -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { head -1 "$f" > $OUT/pmc_fetch_calib.csv; grep "k_stream<0, 4>" "$f" >> $OUT/pmc_fetch_calib.csv; }
fi
ls -la $OUT
