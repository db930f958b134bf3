# scratch (round 6): when a non-scattering slot goes bottom-up -- unreached rows hold fewer than nnz / bfs_bu_div non-zeros (default 3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for g in ${GRAPHS:-orkut ogbn_products hollywood pokec ogbl_ppa googleplus}; do for d in 3 2 1; do
  echo -n "$g [bfs_bu_div=$d] "; GRAPHLILY_DEBUG="bfs_bu_div=$d" timeout 200 python scripts/bfs_call_times.py --graph $g --calls 20 --modes pull_push 2>&1 | grep "pull" | sed -e 's/ | .*//' -e 's/; cpu now [0-9]*//' | tr '\n' ' '; echo
done; done; done
