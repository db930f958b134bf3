# round 5, batch 8: delta-coded cold stream -- format identity + SpMV parity tests, smoke; same-box A/B: round 4's build, the
# run-coded-hot build (r05a) and the current one, general (flags=4) and pattern (flags=0) layouts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_format.py tests/test_gpu_spmv.py tests/test_gpu_typed.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
GRAPHS="orkut ogbn_products pokec hollywood ogbl_ppa googleplus orkut_community orkut_community_shuffled" bash scripts/ab_variants.sh r04 r05a cur 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ab_delta_cold.txt
