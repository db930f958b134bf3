cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GRAPHLILY_SPMV_PREFETCH=1 timeout 900 python -m pytest tests/test_gpu_spmv.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for PF in 0 1; do
echo "== prefetch=$PF general"
GRAPHLILY_SPMV_PREFETCH=$PF timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,ogbn_products,orkut --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== prefetch=$PF pattern"
GRAPHLILY_SPMV_PREFETCH=$PF timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,ogbn_products,orkut --flags 0 --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-120
done; done
