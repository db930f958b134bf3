# round 5, batch 13: software-pipelined slots (next slot's stream loads in flight under this slot's accumulates) -- parity tests,
# then same-box A/B against the unpipelined delta-coded build (r05b) and round 4's
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_format.py tests/test_gpu_spmv.py tests/test_gpu_typed.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
GRAPHS="orkut ogbn_products pokec hollywood ogbl_ppa googleplus orkut_community_shuffled" bash scripts/ab_variants.sh r04 r05b cur 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ab_pipelined.txt
