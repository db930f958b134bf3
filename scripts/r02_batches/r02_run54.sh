cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_spmspv.py tests/test_gpu_apps.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 | cut -c1-220
timeout 900 python benchmarks/bench_spmspv.py --semirings Logical --graphs googleplus,pokec --out gpurun_out/tiny_sweep.jsonl 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['graph'], d['vector_sparsity'], d['vector_nnz'], d['active_nnz'], d['ms_median'], d['ms_max'], d['verified'])
"
