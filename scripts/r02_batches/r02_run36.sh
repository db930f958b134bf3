cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -15
for g in orkut pokec googleplus hollywood; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "graph=1\|pull:" | tail -3; done
