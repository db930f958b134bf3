cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py tests/test_gpu_configs.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for g in orkut pokec googleplus hollywood ogbl_ppa ogbn_products; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1\|pull:" | cut -c1-75,200-; done
