cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_spmv.py -m gpu -x -q -k helper_modes 2>&1 | tail -3
timeout 300 python scripts/r02_clocks.py orkut base gpurun_out/clocks_orkut_a.txt 2>&1 | grep -v amdgpu.ids | head -12
timeout 300 python scripts/r02_clocks.py orkut base gpurun_out/clocks_orkut_b.txt 2>&1 | grep -v amdgpu.ids | grep "kernel span\|whole unit"
timeout 300 python scripts/r02_clocks.py ogbn_products base gpurun_out/clocks_products_a.txt 2>&1 | grep -v amdgpu.ids | grep "kernel span\|whole unit"
