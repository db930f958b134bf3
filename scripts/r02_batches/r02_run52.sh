cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -25 | cut -c1-220; done
