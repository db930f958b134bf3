cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
for g in ogbn_products orkut hollywood; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-75 | head -2; done
