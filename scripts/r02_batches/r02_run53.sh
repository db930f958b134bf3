cd $GRAFT_REPO_ROOT
echo "== apps alone"; timeout 900 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-200
echo "== both, KEEP 0"; GRAPHLILY_SPMV_KEEP_MB=0 timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-200
echo "== apps: only device loop tests"; timeout 900 python -m pytest tests/test_gpu_apps.py -m gpu -x -q -k "device_loop or returns_to_push" 2>&1 | tail -2 | cut -c1-200
