cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_apps.py tests/test_reference_cases.py tests/test_gpu_format.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python scripts/r02_spmv_ab.py --graphs googleplus,pokec,ogbl_ppa,hollywood,ogbn_products,orkut --variants "base;HELPER=0;HELPER=1" --out gpurun_out/r02_ab_helper.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,hollywood,orkut --flags 0 --variants "base;HELPER=0" --out gpurun_out/r02_ab_helper_pattern.jsonl 2>&1 | grep -v amdgpu.ids | cut -c1-330
