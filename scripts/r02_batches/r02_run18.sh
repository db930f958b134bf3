cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_apps.py tests/test_gpu_spmspv.py -m gpu -x -q 2>&1 | tail -5
for g in orkut googleplus pokec; do timeout 300 python scripts/r02_bfs_loop.py $g 2>&1 | grep -v amdgpu; done
echo "== GRAPHLILY_BFS_BACK=0"
for g in orkut googleplus; do GRAPHLILY_BFS_BACK=0 timeout 300 python scripts/r02_bfs_loop.py $g 2>&1 | grep -v amdgpu | grep "graph=1 overlap=0"; done
