cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_apps.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python scripts/r02_spmv_ab.py --graphs googleplus,pokec,ogbl_ppa --variants "base;FUSE=0;BLOCKS=256,SEGMENTS=1;BLOCKS=256,SEGMENTS=1,FUSE=0;HOT_FLOOR=1;HOT_FLOOR=1,BLOCKS=256,SEGMENTS=1" --out gpurun_out/r02_ab_small.jsonl 2>&1 | grep -v amdgpu.ids
timeout 900 python scripts/r02_spmv_ab.py --graphs hollywood,ogbn_products,orkut --variants "base;FUSE=0;HOT_FLOOR=1" --out gpurun_out/r02_ab_large.jsonl 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/r02_spmv_ab.py --graphs pokec,orkut --flags 0 --variants "base;FUSE=0" --out gpurun_out/r02_ab_pattern.jsonl 2>&1 | grep -v amdgpu.ids
