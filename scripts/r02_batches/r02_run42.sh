cd $GRAFT_REPO_ROOT
for g in orkut pokec ogbn_products; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "pull bits" ; done
