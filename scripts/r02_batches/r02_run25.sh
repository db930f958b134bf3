cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_spmv.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in scripts/_variants/head.so ""; do
echo "== lib=${lib:-cur} general"
GRAPHLILY_HIP_LIB=$lib timeout 600 python scripts/r02_spmv_ab.py --graphs googleplus,pokec,ogbn_products,orkut --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-110
echo "== lib=${lib:-cur} pattern"
GRAPHLILY_HIP_LIB=$lib timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,orkut --flags 0 --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-110
done; done
