cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest "tests/test_gpu_apps.py::test_bfs_pull_push_device_loop_equals_host_loop" -m gpu -x -q 2>&1 | grep -v "^$" | tail -12 | cut -c1-200; done
echo "== KEEP 0"
for i in 1 2 3; do GRAPHLILY_SPMV_KEEP_MB=0 timeout 900 python -m pytest "tests/test_gpu_apps.py::test_bfs_pull_push_device_loop_equals_host_loop" -m gpu -x -q 2>&1 | tail -2 | cut -c1-200; done
