cd $GRAFT_REPO_ROOT
env | grep -i "sdma\|HSA_\|HIP_\|ROC" | head
export GRAPHLILY_BFS_DEBUG=1
for K in 1 2 4; do
echo "== side copy kernel x$K"
GRAPHLILY_SIDE_COPY_KERNEL=$K python scripts/r02_bfs_loop.py orkut 2>&1 | grep "bits=1 graph=1\|BFS_DEBUG" | tail -3
done
echo "== early copy off"
GRAPHLILY_BFS_EARLY_COPY=0 python scripts/r02_bfs_loop.py orkut 2>&1 | grep "bits=1 graph=1" | tail -3
python scripts/r02_bfs_loop.py pokec 2>&1 | grep "bits=1 graph=1\|BFS_DEBUG" | tail -3
GRAPHLILY_BFS_EARLY_COPY=0 python scripts/r02_bfs_loop.py pokec 2>&1 | grep "bits=1 graph=1" | tail -3
unset GRAPHLILY_BFS_DEBUG
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^CALL" /tmp/bfs_trace.log || tail -20 /tmp/bfs_trace.log
python scripts/r02_timeline.py /tmp/bfs_trace | tee gpurun_out/r02_bfs_bits_timeline.txt
