cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -5
for g in orkut pokec googleplus hollywood ogbl_ppa ogbn_products; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "graph=1\|pull:" | tail -3; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bfs_trace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bfs_trace -- python $GRAFT_REPO_ROOT/scripts/r02_bfs_trace.py pokec > /tmp/bfs_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/r02_timeline.py /tmp/bfs_trace | head -12
