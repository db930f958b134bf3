cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_apps.py tests/test_gpu_spmspv.py -m gpu -x -q 2>&1 | tail -2
