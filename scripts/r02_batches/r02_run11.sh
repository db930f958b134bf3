cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "" scripts/_variants/f32.so; do
echo "== lib=${lib:-cur}"
GRAPHLILY_HIP_LIB=$lib timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,ogbn_products,orkut --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-330
done; done
echo "== pattern"
for lib in "" scripts/_variants/f32.so; do
echo "== lib=${lib:-cur}"
GRAPHLILY_HIP_LIB=$lib timeout 600 python scripts/r02_spmv_ab.py --graphs orkut --flags 0 --variants "base" --reps 2 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
timeout 600 python benchmarks/bench_spmspv.py --graphs googleplus,pokec --semirings Arithmetic --out gpurun_out/r02_spmspv_sweep_b.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep "0.9999\|0.999,"
