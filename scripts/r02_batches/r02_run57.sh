cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for g in orkut pokec hollywood ogbn_products googleplus; do
echo -n "4096: "; GRAPHLILY_HIP_LIB=scripts/_variants/chunk4096.so python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-70
echo -n "1024: "; python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1" | cut -c1-70
done; done
