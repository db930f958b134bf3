# scratch (round 6): blocking BFS calls process by process -- the streamed read-back on / off (GRAPHLILY_BFS_STREAM), the calling thread bound
# to the GPU's NUMA node by gl_init or not (GRAPHLILY_BIND_NUMA)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4; do for bind in 0 1; do for sm in 0 1; do echo "== BIND_NUMA=$bind BFS_STREAM=$sm"; GRAPHLILY_BIND_NUMA=$bind GRAPHLILY_BFS_STREAM=$sm timeout 200 python scripts/bfs_call_times.py --calls 30 --modes ${MODES:-pull_push} 2>&1 | grep "pull\|cpu now" | cut -c1-150; done; done; done
