# scratch (round 6): the host half of a blocking BFS call -- store mode of the expansion, team size, and the call's timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
grep -m1 "model name" /proc/cpuinfo
for t in 8 16 32; do for m in stream cached; do GRAPHLILY_HOST_THREADS=$t GRAPHLILY_HOST_STORES=$m timeout 100 python scripts/unpack_bench.py 2>&1 | grep stores; done; done
for i in 1 2 3; do for m in stream cached; do for t in 16 32; do echo "== HOST_STORES=$m HOST_THREADS=$t"; GRAPHLILY_HOST_THREADS=$t GRAPHLILY_HOST_STORES=$m timeout 200 python scripts/bfs_call_times.py --calls 30 --modes pull_push,pull 2>&1 | grep "pull" | cut -c1-110; done; done; done
GRAPHLILY_DEBUG=levels_stream_stamps=1 timeout 200 python scripts/bfs_call_times.py --calls 4 --modes pull_push 2>&1 | grep "levels stream" | head -18 | tail -4
