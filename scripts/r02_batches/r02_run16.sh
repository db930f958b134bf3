cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_format.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python scripts/r02_spmv_ab.py --graphs googleplus,pokec,ogbl_ppa,hollywood,ogbn_products,orkut --variants "base;BALANCE=0" --out gpurun_out/r02_ab_balance.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-150
timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,ogbn_products,orkut --flags 0 --variants "base;BALANCE=0" --out gpurun_out/r02_ab_balance_pattern.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-150
timeout 300 python scripts/r02_clocks.py orkut base gpurun_out/clocks_orkut_c.txt 2>&1 | grep -v amdgpu.ids | grep "kernel span\|whole unit\|loop (wave"
