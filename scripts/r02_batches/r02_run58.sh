cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py orkut > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/r02_timeline.py /tmp/bfs_trace | tee gpurun_out/r02_bfs_bits_timeline.txt
cd /tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py hollywood > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/r02_timeline.py /tmp/bfs_trace
