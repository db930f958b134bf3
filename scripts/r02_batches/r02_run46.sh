cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -3
for div in 3 2 1; do
echo "== BU_DIV=$div"
for g in orkut pokec googleplus hollywood ogbl_ppa ogbn_products; do
GRAPHLILY_BFS_BU_DIV=$div python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1\|pull bits=1 graph=1" | cut -c1-70 | head -2
done; done
