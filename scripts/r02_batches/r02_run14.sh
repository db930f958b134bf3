cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "one_rank or two_ranks" 2>&1 | tail -15
timeout 300 python bench.py --gpus 1 --force-dist --steps 50 --no-cpu-baseline --bfs-runs 3 2>&1 | grep "^{" | cut -c1-1500
for g in orkut googleplus; do timeout 300 python scripts/r02_bfs_loop.py $g 2>&1 | grep -v amdgpu; done
