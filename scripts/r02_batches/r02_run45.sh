cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -5
for g in orkut pokec googleplus hollywood ogbl_ppa ogbn_products; do
echo -n "bu off: "; GRAPHLILY_BFS_BU_DIV=0 python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-70
python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1\|pull bits=1 graph=1" | cut -c1-70 | head -2
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py orkut > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/r02_timeline.py /tmp/bfs_trace | tee gpurun_out/r02_bfs_bits_timeline.txt
