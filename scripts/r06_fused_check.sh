# scratch (round 6): the streamed read-back after a change of its fences -- the test that catches stale chunks (repeated), the BFS
# tests, the schedule's kernel trace and the blocking calls with the read-back streamed (1) and the round-5 way (0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_runtime.py -k streamed 2>&1 | tail -1; done
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_runtime.py tests/test_gpu_apps.py tests/test_gpu_fullsize.py tests/test_cpp_layer.py 2>&1 | tail -3
bash scripts/r06_bfs_trace.sh 2>&1 | tail -12
for i in 1 2 3; do for sm in 0 1; do echo "== BFS_STREAM=$sm"; GRAPHLILY_BFS_STREAM=$sm timeout 200 python scripts/bfs_call_times.py --calls 30 --no-timed 2>&1 | grep "pull" | cut -c1-110; done; done
GRAPHLILY_DEBUG=levels_stream_stamps=1 timeout 200 python scripts/bfs_call_times.py --calls 5 --no-timed --modes pull_push 2>&1 | grep "levels stream" | tail -2
