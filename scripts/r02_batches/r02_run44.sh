cd $GRAFT_REPO_ROOT
export GRAPHLILY_BFS_BACK=1.0
for rep in 1 2; do
for g in orkut pokec hollywood; do
echo -n "prev: "; GRAPHLILY_HIP_LIB=scripts/_variants/prev.so GRAPHLILY_SPMSPV_PULL_DIV=128 python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-70
echo -n "cur:  "; python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-70
done; done
