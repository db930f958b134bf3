cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5
