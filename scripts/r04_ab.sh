# round 4, first batch: same-box A/B of the BFS / SSSP schedules on the six stand-ins + kernel timelines of googleplus
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scripts/r04_ab_schedules.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_ab_schedules.txt
echo "ab rc=$?"; cut -c1-900 gpurun_out/r04_ab_schedules.txt
for g in googleplus; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py $g > /tmp/bfs_trace.log 2>&1
  cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log | tail -2
  python scripts/r02_timeline.py /tmp/bfs_trace > gpurun_out/r04_bfs_timeline_$g.txt; tail -25 gpurun_out/r04_bfs_timeline_$g.txt
done
