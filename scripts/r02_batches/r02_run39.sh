cd $GRAFT_REPO_ROOT
for cfg in "32 default" "64 default" "128 default" "64 1.0" "128 1.0" "256 1.0"; do
set -- $cfg
export GRAPHLILY_SPMSPV_PULL_DIV=$1
if [ "$2" = "default" ]; then unset GRAPHLILY_BFS_BACK; else export GRAPHLILY_BFS_BACK=$2; fi
echo "== PULL_DIV=$1 BACK=$2"
for g in orkut pokec googleplus hollywood ogbl_ppa ogbn_products; do python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-70,200-; done
done
