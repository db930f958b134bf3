cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -3
for g in orkut pokec ogbn_products hollywood; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "pull bits\|bits=1 graph=1" | cut -c1-80 ; done
