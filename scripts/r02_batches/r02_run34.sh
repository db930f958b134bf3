cd $GRAFT_REPO_ROOT
export GRAPHLILY_BFS_DEBUG=1
python scripts/r02_bfs_loop.py pokec 2>&1 | grep -v "^pokec device_loop=0\|bits=0" | tail -8
python scripts/r02_bfs_loop.py googleplus 2>&1 | grep -v "^pokec device_loop=0\|bits=0" | tail -5
python scripts/r02_bfs_loop.py orkut 2>&1 | grep -v "^pokec device_loop=0\|bits=0" | tail -5
unset GRAPHLILY_BFS_DEBUG
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log || tail -20 /tmp/bfs_trace.log
python scripts/r02_timeline.py /tmp/bfs_trace | tee gpurun_out/r02_bfs_bits_timeline.txt
python scripts/r02_bfs_trace_summary.py /tmp/bfs_trace /tmp/bfs_trace.log | head -12
