# round 5, batch 14: pipelined general layout (ticket drawn ahead of the gathers), unpipelined pattern layout: parity + A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_format.py tests/test_gpu_spmv.py tests/test_gpu_typed.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -5
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
GRAPHS="orkut ogbn_products pokec hollywood ogbl_ppa googleplus orkut_community_shuffled" bash scripts/ab_variants.sh r05b cur 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ab_pipelined_general_only.txt
