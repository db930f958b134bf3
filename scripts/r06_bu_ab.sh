# scratch (round 6): rows per lane and step of the bottom-up BFS scan (GL_BFS_BU_ROWS: in-tree 2, variants bu3 / bu4): schedule GPU time + calls
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products hollywood pokec}; do for v in cur bu3 bu4; do
  lib=""; [ "$v" != cur ] && lib=scripts/_variants/$v.so
  echo -n "$g [$v] "; GRAPHLILY_HIP_LIB=$lib timeout 200 python scripts/bfs_call_times.py --graph $g --calls 20 2>&1 | grep "pull" | sed -e 's/ | .*//' | tr '\n' ' '; echo
done; done; done
