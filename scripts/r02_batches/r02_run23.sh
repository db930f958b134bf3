cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_apps.py tests/test_gpu_spmspv.py tests/test_gpu_apply.py tests/test_gpu_configs.py tests/test_gpu_typed.py tests/test_reference_cases.py -m gpu -x -q 2>&1 | tail -4
for g in orkut googleplus pokec; do timeout 300 python scripts/r02_bfs_loop.py $g 2>&1 | grep -v amdgpu | grep "device_loop=0\|graph=1 overlap=0"; done
echo "== with the scan launch"
for g in orkut googleplus; do GRAPHLILY_COMPACT_SCAN=1 timeout 300 python scripts/r02_bfs_loop.py $g 2>&1 | grep -v amdgpu | grep "device_loop=0\|graph=1 overlap=0"; done
timeout 600 python benchmarks/bench_spmspv.py --graphs googleplus,pokec --semirings Arithmetic 2>&1 | grep -v amdgpu.ids | cut -c1-260 | grep "0.9999\|0.99,\|\"vector_sparsity\": 0.9,"
