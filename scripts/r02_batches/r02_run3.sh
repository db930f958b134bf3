cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_apps.py tests/test_gpu_format.py -m gpu -x -q 2>&1 | tail -8
python scripts/r02_bfs_loop.py orkut 2>&1 | grep -v amdgpu
python scripts/r02_bfs_loop.py googleplus 2>&1 | grep -v amdgpu
