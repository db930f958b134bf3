cd $GRAFT_REPO_ROOT
for B in default 0 0.015625 0.0005; do
echo "== GRAPHLILY_BFS_BACK=$B"
if [ $B = default ]; then unset GRAPHLILY_BFS_BACK; else export GRAPHLILY_BFS_BACK=$B; fi
timeout 300 python scripts/r02_bfs_loop.py orkut 2>&1 | grep -v amdgpu | grep "device_loop=0\|graph=1 overlap=0"
done
unset GRAPHLILY_BFS_BACK
GRAPHLILY_BFS_BACK=0.015625 timeout 300 python scripts/r02_bfs_loop.py ogbn_products 2>&1 | grep -v amdgpu | grep "device_loop=0\|graph=1 overlap=0"
GRAPHLILY_BFS_BACK=0 timeout 300 python scripts/r02_bfs_loop.py ogbn_products 2>&1 | grep -v amdgpu | grep "graph=1 overlap=0"
