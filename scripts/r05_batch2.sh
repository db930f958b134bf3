# round 5, batch 2: the run-coded hot stream -- format tests, SpMV parity tests, smoke; then same-box A/B against round 4's
# build (scripts/_variants/r04.so) on the general (flags=4) and pattern (flags=0) layouts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_format.py tests/test_gpu_spmv.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
GRAPHS="orkut ogbn_products pokec hollywood" bash scripts/ab_variants.sh r04 cur 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_ab_runcoded.txt
