# after the final batch: the bench line, its three-launch A/B and the BFS timelines again (the packed read-back moved into the
# recorded schedule after scripts/r03_final.sh had run)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --emulate-rank 0/8,3/8,7/8,1/4,0/2 > gpurun_out/r03_bench_orkut_n1.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$?"
GRAPHLILY_BFS_SHARD_STEP=0 GRAPHLILY_BFS_ONE_LAUNCH=0 GRAPHLILY_BFS_U8=0 timeout 900 python bench.py --emulate-rank 0/8,1/4,0/2 --no-cpu-baseline --no-pattern > gpurun_out/r03_bench_orkut_n1_three_launch_slots.json 2>> gpurun_out/r03_bench_final.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/r02_timeline.py /tmp/bfs_trace > gpurun_out/r03_bfs_bits_timeline.txt; tail -4 gpurun_out/r03_bfs_bits_timeline.txt
for spec in "0/8 pull_push" "3/8 pull_push" "0/8 pull" "1/4 pull_push" "0/2 pull_push"; do set -- $spec
  cd /tmp && rm -rf /tmp/emu_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/emu_trace -- python $GRAFT_REPO_ROOT/scripts/r03_emulate_trace.py orkut $1 $2 > /tmp/emu_trace.log 2>&1
  cd $GRAFT_REPO_ROOT; n=$(echo $1 | tr / of); { grep "^CALL 4" /tmp/emu_trace.log; python scripts/r03_emulate_trace_summary.py /tmp/emu_trace; } > gpurun_out/r03_emulated_rank${n}_$2_timeline.txt; tail -1 gpurun_out/r03_emulated_rank${n}_$2_timeline.txt
done
uptime
