cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
for g in orkut googleplus; do timeout 300 python scripts/r02_bfs_loop.py $g 2>&1 | grep -v amdgpu | grep "device_loop=0\|graph=1 overlap=0"; done
