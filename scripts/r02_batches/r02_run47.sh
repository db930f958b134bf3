cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for g in orkut ogbn_products; do
python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1\|pull bits=1 graph=1" | cut -c1-70 | head -2
done
