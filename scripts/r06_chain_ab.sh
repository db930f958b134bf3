# scratch (round 6): chained SpMV runs (gl_spmv_plan_chain) -- parity tests, then PageRank / SSSP per graph with the chain off and on
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_apps.py tests/test_cpp_layer.py tests/test_gpu_spmv.py tests/test_gpu_configs.py 2>&1 | tail -4
for rep in 1 2; do for c in 0 1; do
  GRAPHLILY_SPMV_CHAIN=$c timeout 900 python benchmarks/bench_graphs.py --graphs ${GRAPHS:-orkut,ogbn_products,hollywood,pokec} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('chain=$c', r['graph'], 'pagerank ms/iter', r['pagerank']['ms_per_iter'], r['pagerank']['ok'], 'sssp pull', r['sssp']['pull_ms'], 'pull_push', r['sssp']['pull_push_ms'], r['sssp']['ok'])"
done; done
for c in 0 1; do echo "== reference drivers, GRAPHLILY_SPMV_CHAIN=$c"; GRAPHLILY_SPMV_CHAIN=$c timeout 900 python benchmarks/run_reference_benches.py --graph orkut --apps pagerank,sssp 2>&1 | grep "average_time\|iteration:"; done
