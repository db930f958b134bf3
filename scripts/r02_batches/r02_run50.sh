cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -3
for keep in 0 224; do
echo "== KEEP_MB=$keep"
export GRAPHLILY_SPMV_KEEP_MB=$keep
for g in googleplus pokec ogbl_ppa hollywood; do
for f in 4 0 2; do echo -n "$g flags=$f: "; python scripts/probe_spmv.py --graph $g --flags $f --no-copy --iters 100 2>&1 | grep "^op 0 mask 0\|^op 1 mask 1" | head -1 | cut -c1-90; done
python scripts/r02_bfs_loop.py $g 2>&1 | grep "bits=1 graph=1\|pull bits=1 graph=1" | cut -c1-70 | head -2
done; done
